"""f2: FEB / FFAB (RawFomer_WFB_FFAB/blocks.py:11-92), the hand-written rfft2 / irfft2 they stand on, and the Mamba-free
wavelet branch of WMB (model.py:215-243).  Fixtures: tests/golden/ffab.npz, outputs of the reference's own modules
(oracle/make_golden.py --only-ffab).

Tolerances.  Transforms: 2e-6 on |F| (values of O(1)); the phase is compared as a complex unit vector weighted by the
magnitude (|F| e^{i angle}), because angle() is ill-conditioned where |F| -> 0 and 2 pi-periodic at the +-pi cut, where the
reference itself flips with rounding; the four bins that are real by symmetry must reproduce the reference's +pi / 0
exactly.  FEB <= 2e-5, FFAB <= 2e-4 max-abs on outputs of magnitude up to ~30 (the reference's own float32-vs-float64
difference on these cases is 1e-6 / 1e-5: tools in oracle/make_golden.py log it)."""
import numpy as np
import pytest
import torch

import cases
from cases import golden, rnd
from oracle import rawformer_ref as R

FFT_CASES = (("p2", (2, 3, 16, 32)), ("mixed", (1, 2, 10, 14)), ("sq", (1, 2, 64, 64)), ("tall", (1, 1, 24, 8)))
FEB_CASES = (("feb16", 16, (2, 16, 16, 16)), ("feb8", 8, (1, 8, 12, 20)))
FFAB_CASES = (("ffab16", 16, (2, 16, 16, 16)), ("ffab8", 8, (1, 8, 8, 12)))


def feb_spec(nc):
    s = {"fpre.weight": (nc, nc, 1, 1), "fpre.bias": (nc,)}
    for p in ("process1", "process2"):
        for i in (0, 2):
            s[f"{p}.{i}.weight"], s[f"{p}.{i}.bias"] = (nc, nc, 1, 1), (nc,)
    return s


def pb_spec(nc):
    s = {"frequency_process." + k: v for k, v in feb_spec(nc).items()}
    s.update({"cat.weight": (nc, nc, 1, 1), "cat.bias": (nc,)})
    return s


def ffab_spec(nc):
    s = {"conv0.0.weight": (nc, nc, 1, 1), "conv0.0.bias": (nc,)}
    s.update({"conv0.1." + k: v for k, v in pb_spec(nc).items()})
    for i in (1, 2, 3):
        s.update({f"conv{i}." + k: v for k, v in pb_spec(nc).items()})
    for n in ("conv4", "conv5", "convout"):
        s.update({f"{n}.0." + k: v for k, v in pb_spec(2 * nc).items()})
        s[f"{n}.1.weight"], s[f"{n}.1.bias"] = (nc, 2 * nc, 1, 1), (nc,)
    return s


def params(spec, seed):
    return cases.params(spec, seed=seed)


def maxabs(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max())


# ------------------------------------------------------------------------------------------ CPU: oracle vs reference fixtures
def test_oracle_feb_ffab_match_the_reference():
    g = golden("ffab")
    for tag, nc, shape in FEB_CASES:
        assert maxabs(R.feb(rnd(f"ffab.{tag}.x", shape, -1.5, 1.5, seed=61), params(feb_spec(nc), 1000 + nc), ""), g[f"{tag}.out"]) <= 1e-6
    for tag, nc, shape in FFAB_CASES:
        assert maxabs(R.ffab(rnd(f"ffab.{tag}.x", shape, seed=61), params(ffab_spec(nc), 2000 + nc), ""), g[f"{tag}.out"]) <= 1e-6


def wmb_params(nc):
    p = {"norm1.body." + k: v for k, v in params({"weight": (nc,), "bias": (nc,)}, 3000 + nc).items()}
    p.update({"illu." + k: v for k, v in params(cases.wfb_ie_spec(nc) | {"conv1.weight": (nc, nc + 1, 1, 1), "conv2.weight": (nc, nc, 1, 1),
                                                                       "conv2.bias": (nc,)}, 3100 + nc).items()})
    p.update({"ffab." + k: v for k, v in params(ffab_spec(nc), 3200 + nc).items()})
    return p


def test_oracle_wmb_wavelet_branch_matches_the_reference_modules():
    g = golden("ffab")
    assert maxabs(R.wmb_ll_branch(rnd("wmb.wmb16.x", (2, 16, 16, 24), seed=62), wmb_params(16), ""), g["wmb16.out"]) <= 2e-5


# ------------------------------------------------------------------------------------------ GPU
def _dev(d, device):
    return {k: v.to(device) for k, v in d.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("tag,shape", FFT_CASES)
def test_rfft2_polar_and_inverse(device, tag, shape):
    from bayer_low_light_image_enhancement_amd import ops
    g = golden("ffab")
    h, w = shape[-2:]
    mag, pha = ops.rfft2_polar(rnd(f"fft.{tag}.x", shape, seed=71).to(device))
    mag, pha = mag.cpu().double(), pha.cpu().double()
    rm, rp = torch.from_numpy(g[f"fft.{tag}.mag"]).double(), torch.from_numpy(g[f"fft.{tag}.pha"]).double()
    assert float((mag - rm).abs().max()) <= 2e-6
    assert float(((mag * torch.cos(pha) - rm * torch.cos(rp)) ** 2 + (mag * torch.sin(pha) - rm * torch.sin(rp)) ** 2).sqrt().max()) <= 3e-6
    for y in {0, h // 2}:                       # bins that are real by symmetry: exactly 0 or +pi
        for x in {0, w // 2}:
            assert bool(((pha[..., y, x] == 0) | (pha[..., y, x].float() == np.float32(np.pi))).all()), (y, x)
            if h & (h - 1) == 0 and w & (w - 1) == 0:     # ... which is what the reference gives for power-of-two sizes (for other
                assert torch.equal(pha[..., y, x].float(), rp[..., y, x].float()), (y, x)   # sizes its imaginary part there is rounding noise)
    inv = ops.polar_irfft2(rnd(f"fft.{tag}.mag", tuple(rm.shape), 0.0, 2.0, seed=72).to(device),
                           rnd(f"fft.{tag}.pha", tuple(rm.shape), -3.0, 3.0, seed=73).to(device), w)
    assert maxabs(inv, g[f"fft.{tag}.inv"]) <= 3e-6
    # size-independent property: irfft2(rfft2(x)) = x
    x = rnd("fft.rt", (2, 4, 128, 256), seed=74).to(device)
    m2, p2 = ops.rfft2_polar(x)
    assert maxabs(ops.polar_irfft2(m2 - 1e-6, p2, 256), x) <= 5e-6


@pytest.mark.gpu
def test_feb_and_ffab_match_the_reference(device):
    from bayer_low_light_image_enhancement_amd import ops
    g = golden("ffab")
    for tag, nc, shape in FEB_CASES:
        out = ops.feb(rnd(f"ffab.{tag}.x", shape, -1.5, 1.5, seed=61).to(device), _dev(params(feb_spec(nc), 1000 + nc), device))
        assert maxabs(out, g[f"{tag}.out"]) <= 2e-5, tag
    for tag, nc, shape in FFAB_CASES:
        out = ops.ffab(rnd(f"ffab.{tag}.x", shape, seed=61).to(device), _dev(params(ffab_spec(nc), 2000 + nc), device))
        assert maxabs(out, g[f"{tag}.out"]) <= 2e-4, tag


@pytest.mark.gpu
def test_feb_larger_against_oracle(device):
    """A 128 x 128 power-of-two plane (the LL band of a 256 x 256 level) against the oracle as the reference computes it, and a
    non-power-of-two plane (36 x 52, direct-DFT path) against the oracle with exact self-conjugate bins: there the reference's
    own phase at a negative real bin is +-pi by rounding noise (see oracle feb()), so parity with IT is unpinned by nature."""
    from bayer_low_light_image_enhancement_amd import ops
    for nc, shape, exact in ((32, (2, 32, 128, 128), False), (16, (1, 16, 36, 52), True)):
        p = params(feb_spec(nc), 1100 + nc)
        x = rnd("feb.big.x", shape, -2.0, 2.0, seed=63)
        assert maxabs(ops.feb(x.to(device), _dev(p, device)), R.feb(x, p, "", exact_symmetric_bins=exact)) <= 5e-5, shape


@pytest.mark.gpu
def test_wmb_wavelet_branch(device):
    """DWT -> Illumination_Estimator -> FFAB -> IWT composed as WMB.forward does (model.py:215-243, Mamba left out): the
    wavelet kernels inside a block of the reference."""
    from bayer_low_light_image_enhancement_amd import ops
    g = golden("ffab")
    out = ops.wmb_ll_branch(rnd("wmb.wmb16.x", (2, 16, 16, 24), seed=62).to(device), _dev(wmb_params(16), device))
    assert maxabs(out, g["wmb16.out"]) <= 2e-4
