"""CPU: the plain-C restatement (oracle/dwt_ref.c) against the reference's outputs in
tests/golden/per_op.npz and against the torch oracle.  dwt_init / iwt_init and the two shuffles
are bit-exact (same expression order as the reference); CustomDWT/IDWT/HaarDWT within 1e-6."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

import cases
from cases import golden, rnd
from oracle import rawformer_ref as R


@pytest.fixture(scope="module")
def lib():
    odir = os.path.join(cases.REPO, "oracle")
    subprocess.run(["make", "-s", "-C", odir], check=True)
    return C.CDLL(os.path.join(odir, "_build", "libdwt_ref.so"))


def run(fn, x, out_shape, *ints, k=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(out_shape, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    args = [x.ctypes.data_as(fp), out.ctypes.data_as(fp)]
    if k is not None:
        kk = np.ascontiguousarray(k, dtype=np.float32)
        args.append(kk.ctypes.data_as(fp))
    fn(*args, *[C.c_int(i) for i in ints])
    return out


def test_shuffles_and_haar_init_bit_exact(lib):
    g = golden("per_op")
    x = rnd("x.shuffle", (2, 3, 8, 12)).numpy()
    assert np.array_equal(run(lib.ref_pixel_unshuffle2, x, (2, 12, 4, 6), 2, 3, 4, 6), g["downshuffle"])
    x = rnd("x.pixelshuffle", (2, 12, 6, 10)).numpy()
    assert np.array_equal(run(lib.ref_pixel_shuffle2, x, (2, 3, 12, 20), 2, 3, 6, 10), g["pixelshuffle"])
    x = rnd("x.dwt", (2, 5, 12, 20)).numpy()
    assert np.array_equal(run(lib.ref_dwt_init, x, (8, 5, 6, 10), 2, 5, 6, 10), g["dwt_init"])
    x = rnd("x.iwt", (8, 5, 6, 10)).numpy()
    assert np.array_equal(run(lib.ref_iwt_init, x, (2, 5, 12, 20), 2, 5, 6, 10), g["iwt_init"])


def test_custom_and_orthonormal_haar(lib):
    g = golden("per_op")
    x = rnd("x.dwt", (2, 5, 12, 20)).numpy()
    k_def = np.asarray(R.DEFAULT_CUSTOM_KERNEL, dtype=np.float32) / 2
    y = run(lib.ref_custom_dwt, x, (2, 20, 6, 10), 2, 5, 6, 10, k=k_def)
    assert np.abs(y - g["custom_dwt_default"]).max() <= 1e-6
    z = run(lib.ref_custom_idwt, y, (2, 5, 12, 20), 2, 5, 6, 10, k=k_def)
    assert np.abs(z - g["custom_idwt_default"]).max() <= 1e-6
    k2 = g["custom_kernel_rand"]
    y2 = run(lib.ref_custom_dwt, x, (2, 20, 6, 10), 2, 5, 6, 10, k=k2)
    assert np.abs(y2 - g["custom_dwt_rand_nonorm"]).max() <= 1e-6
    for tag, shp in (("even", (2, 3, 12, 20)), ("odd", (2, 3, 17, 19))):
        xh = rnd("x.haar." + tag, shp).numpy()
        h2, w2 = (shp[2] + 1) // 2, (shp[3] + 1) // 2
        o = run(lib.ref_haar_dwt, xh, (4, 2, 3, h2, w2), 2, 3, shp[2], shp[3])
        assert np.abs(o - g[f"haar_{tag}"]).max() <= 1e-6


def test_c_matches_torch_oracle_on_other_shapes(lib):
    x = rnd("x.c.extra", (1, 2, 10, 14))
    assert np.array_equal(run(lib.ref_dwt_init, x.numpy(), (4, 2, 5, 7), 1, 2, 5, 7), R.dwt_init(x).numpy())
    assert np.array_equal(run(lib.ref_pixel_unshuffle2, x.numpy(), (1, 8, 5, 7), 1, 2, 5, 7), R.pixel_unshuffle2(x).numpy())
