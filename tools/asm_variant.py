#!/usr/bin/env python3
"""Diagnostic builds from EDITED device assembly: compile one source with -save-temps, apply a named edit to the gfx950
assembly of one kernel, re-run hipcc's own remaining steps (assembler, lld, offload bundler, host compile) and link the
object with the regular ones into tools/_diag/lib_<name>.so (select it with RF_LIB_PATH).

    asm_variant.py <name> <source.hip> <kernel-substring> <edit> [--patch FILE] [-DFOO ...]

(--patch as in tools/variant.py: the instrumented / prefetch variants live in tools/repro/rf_fused_repro.patch.)  The build is
the RAW hipcc flow -- no operand-select rewrite -- because these edits were made to rule causes OUT on the failing builds.

edits:  none            re-assemble unchanged (checks the flow)
        trans_nop<N>    s_nop N-1 ... i.e. N wait states after every transcendental VALU instruction (v_rcp/v_sqrt/v_rsq/v_exp/v_log)
        vmcnt0          every s_waitcnt vmcnt(N) becomes vmcnt(0)
        nop_all<N>      N wait states after EVERY VALU instruction (slow; shows whether any issue-distance hazard is involved)
        lgkm_serial     s_waitcnt lgkmcnt(0) after every DS instruction;  vm_serial: vmcnt(0) after every VMEM instruction;
        mfma_drain      32 wait states after every run of MFMAs (before the first non-MFMA instruction that follows)
        both_serial     both;  vmload_serial / vmstore_serial: only after loads / only after stores
"""
import os
import re
import shlex
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bayer_low_light_image_enhancement_amd import build as B  # noqa: E402

TRANS = re.compile(r"^\s*v_(rcp|sqrt|rsq|exp|log|sin|cos)_")


def edit_kernel(lines, edit):
    out = []
    prev_mfma = False
    for ln in lines:
        t = ln.strip()
        is_inst = bool(t) and not t.startswith((";", ".", "s_nop")) and not t.endswith(":")
        if edit == "mfma_drain" and is_inst and prev_mfma and not t.startswith("v_mfma"):
            out += ["\ts_nop 15", "\ts_nop 15"]       # 32 wait states between the last MFMA of a run and whatever follows it
        if edit == "dpp_nop" and "_dpp" in t:
            out.append("\ts_nop 7")                     # 8 wait states in front of every DPP instruction
        if is_inst:
            prev_mfma = t.startswith("v_mfma")
        if edit == "vmcnt0":
            ln = re.sub(r"vmcnt\(\d+\)", "vmcnt(0)", ln)
        out.append(ln)
        m = re.match(r"trans_nop(\d+)$", edit)
        if m and TRANS.match(ln):
            out.append(f"\ts_nop {int(m.group(1)) - 1}")
        if edit in ("lgkm_serial", "both_serial") and t.startswith("ds_"):
            out.append("\ts_waitcnt lgkmcnt(0)")
        if edit in ("vm_serial", "both_serial") and (t.startswith("global_") or t.startswith("buffer_") or t.startswith("scratch_")):
            out.append("\ts_waitcnt vmcnt(0)")
        if edit == "vmload_serial" and (t.startswith("global_load") or t.startswith("buffer_load")):
            out.append("\ts_waitcnt vmcnt(0)")
        if edit == "vmstore_serial" and t.startswith("global_store"):
            out.append("\ts_waitcnt vmcnt(0)")
        if edit == "exec_nop" and re.match(r"s_(or|and|andn2|xor|mov|and_saveexec|or_saveexec)_b64\s+(exec|s\[\d+:\d+\], )", t) and "exec" in t:
            out.append("\ts_nop 7")                     # 8 wait states after every SALU instruction that writes or saves EXEC
        m = re.match(r"nop_all(\d+)$", edit)
        if m and t.startswith("v_") and not t.startswith("v_mfma"):
            out.append(f"\ts_nop {int(m.group(1)) - 1}")
    return out


def main():
    name, src, kernel, edit, flags = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:]
    patch = None
    if "--patch" in flags:
        i = flags.index("--patch")
        patch = os.path.abspath(flags[i + 1])
        del flags[i:i + 2]
    B.build_library()
    outdir = os.path.join(REPO, "tools", "_diag")
    os.makedirs(outdir, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="rf_asm_")
    obj = os.path.join(outdir, f"{name}_{src.replace('.hip', '.o')}")
    path = os.path.join(B.CSRC, src)
    if patch:
        path = os.path.join(B.CSRC, f"_variant_{name}_{src}")
        shutil.copy(os.path.join(B.CSRC, src), path)
        subprocess.check_call(["patch", "-s", "-p0", path, patch])
    cmd = [B._hipcc(), *B.FLAGS, *flags, "-c", path, "-o", obj, "-save-temps", "-v"]
    r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
    if patch:
        for junk in (path, path + ".orig", path + ".rej"):
            if os.path.exists(junk):
                os.remove(junk)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    steps = [shlex.split(ln) for ln in r.stderr.splitlines() if ln.startswith(' "')]
    dev_s = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
    text = open(os.path.join(tmp, dev_s)).read().splitlines()
    start = [i for i, ln in enumerate(text) if ln.startswith("_Z") and kernel in ln and re.match(r"^_Z\w+:", ln)]
    assert len(start) == 1, f"{len(start)} kernels match {kernel!r}"
    end = next(i for i in range(start[0], len(text)) if text[i].startswith(".Lfunc_end"))
    new = text[:start[0]] + edit_kernel(text[start[0]:end], edit) + text[end:]
    open(os.path.join(tmp, dev_s), "w").write("\n".join(new) + "\n")
    print(f"{dev_s}: kernel at lines {start[0]}..{end}, {len(new) - len(text)} lines added by edit {edit!r}")
    # re-run everything after the device compile: device assembler (cc1as on the .s), lld, bundler, then the host steps
    first = next(i for i, s in enumerate(steps) if "-cc1as" in s and dev_s in s)
    for s in steps[first:]:
        rr = subprocess.run(s, cwd=tmp, capture_output=True, text=True)
        if rr.returncode:
            sys.exit(" ".join(s)[:300] + "\n" + rr.stderr[-3000:])
    objs = [obj if s == src else os.path.join(B.CSRC, s.replace(".hip", ".o")) for s in B.SOURCES]
    lib = os.path.join(outdir, f"lib_{name}.so")
    subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", lib])
    shutil.copy(os.path.join(tmp, dev_s), os.path.join(outdir, f"{name}.s"))
    shutil.rmtree(tmp)
    print(lib)


if __name__ == "__main__":
    main()
