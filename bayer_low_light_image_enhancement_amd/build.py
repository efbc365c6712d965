"""In-tree build of ``csrc/librawformer_hip.so`` (hipcc, gfx950 only).

``python -m bayer_low_light_image_enhancement_amd.build`` or ``build_library()``.
hipcc cross-compiles without a GPU; the built ``.so`` is git-ignored and travels to the GPU
box with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "librawformer_hip.so")
# Test-only twin of the library: the two host schedules recompiled with -DRF_DIAG, which adds the environment switches
# RF_NO_FUSE / RF_NO_UPCAT / RF_NO_B3 (force the op-by-op schedule / the f32 MFMA GEMMs) so tests can compare the fused kernels with the un-fused chain.
# The shipped library has no such switch.  Selected with RF_LIB_PATH (see _lib.py).
DIAG_LIB = os.path.join(CSRC, "librawformer_hip_diag.so")
DIAG_SOURCES = ["rf_block.hip", "rf_model.hip", "rf_gemm1x1.hip", "rf_fused.hip"]
SOURCES = ["rf_api.hip", "rf_model.hip", "rf_pack.hip", "rf_pointwise.hip", "rf_gemm1x1.hip",
           "rf_conv3x3.hip", "rf_attn.hip", "rf_flca.hip", "rf_fused.hip", "rf_block.hip", "rf_harness.hip", "rf_tokattn.hip", "rf_wfb.hip", "rf_upcat.hip", "rf_fft.hip", "rf_ffab.hip", "rf_truecolor.hip", "rf_train.hip", "rf_trainstep.hip"]
# every header a source may include: ONE list for the product and the diagnostic objects (a stale *_diag.o linked with fresh
# product objects would disagree on struct rf_handle)
HEADERS = [os.path.join(CSRC, "rf_common.h"), os.path.join(CSRC, "rf_handle.h"), os.path.join(HERE, "..", "include", "rawformer_hip.h"),
           os.path.join(HERE, "isa_check.py")]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the RawFormer HIP library cannot be built here")
    return exe


def _run(cmd, cwd=None) -> str:
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=cwd)
    if r.returncode != 0:
        raise RuntimeError("build step failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return r.stderr


def compile_source(src: str, obj: str, extra_flags=(), verbose: bool = False) -> int:
    """One translation unit -> host object with the gfx950 code object embedded, THROUGH the assembly rewrite of isa_check.py:

        hipcc --cuda-device-only -S          gfx950 assembly
        isa_check.commute_vulnerable         packed-f32 op_sel:[0,1,..] -> the exact op_sel:[1,0,..] form (operands exchanged)
        clang (assembler) / lld / clang-offload-bundler      code object -> fat binary, as hipcc itself does
        hipcc --cuda-host-only -fcuda-include-gpubinary      host object embedding it

    Returns the number of instructions rewritten."""
    from . import isa_check

    hipcc = _hipcc()
    llvm = isa_check.LLVM_BIN
    flags = [*FLAGS, *extra_flags]
    tmp = tempfile.mkdtemp(prefix="rf_build_")
    try:
        dev_s, dev_o, dev_co, fb = (os.path.join(tmp, n) for n in ("dev.s", "dev.o", "dev.co", "dev.hipfb"))
        warn = _run([hipcc, *flags, "--cuda-device-only", "-S", src, "-o", dev_s])
        text, nfix = isa_check.commute_vulnerable(open(dev_s).read())
        if nfix:
            open(dev_s, "w").write(text)
        _run([os.path.join(llvm, "clang"), "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", dev_s, "-o", dev_o])
        _run([os.path.join(llvm, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", dev_co, dev_o])
        _run([os.path.join(llvm, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
              "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", f"-input={dev_co}", f"-output={fb}"])
        _run([hipcc, *flags, "--cuda-host-only", "-c", src, "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-o", obj])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if verbose:
        print(f"{os.path.basename(src)}: {nfix} packed-f32 operand-select instruction(s) commuted" + (("\n" + warn) if warn.strip() else ""), flush=True)
    return nfix


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    headers = HEADERS
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(lambda so: compile_source(so[0], so[1], verbose=verbose), jobs))
    if force or jobs or _stale(LIB, objs):
        _run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", LIB])
    return LIB


def build_diag_library(force: bool = False, verbose: bool = False) -> str:
    """``librawformer_hip_diag.so`` = the product objects with rf_block / rf_model rebuilt under -DRF_DIAG."""
    build_library(force=force, verbose=verbose)
    hipcc = _hipcc()
    headers = HEADERS
    objs, rebuilt = [], False
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if src in DIAG_SOURCES:
            o = os.path.join(CSRC, src.replace(".hip", "_diag.o"))
            if force or _stale(o, [s] + headers):
                compile_source(s, o, extra_flags=["-DRF_DIAG"], verbose=verbose)
                rebuilt = True
        else:
            o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
    if force or rebuilt or _stale(DIAG_LIB, objs):
        r = subprocess.run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", DIAG_LIB], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    return DIAG_LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    if "--diag" in sys.argv:
        print(build_diag_library(verbose=True))
