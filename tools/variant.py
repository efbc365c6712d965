#!/usr/bin/env python3
"""Diagnostic builds: recompile ONE source with extra -D flags and link it with the regular objects into
tools/_diag/lib_<name>.so (select it with RF_LIB_PATH).

    variant.py <name> <source.hip> [--patch FILE] [--raw] [-DFOO ...]

--patch FILE   apply a unified diff to a temporary copy of the source first (tools/repro/rf_fused_repro.patch holds the
               instrumented / cross-tile-prefetch variants of the level-0 attention kernel used to find round 2's miscompare)
--raw          compile with plain hipcc, WITHOUT the packed-f32 operand-select rewrite of the product build (isa_check.py):
               this is how the failing builds are reproduced; without it the variant goes through build.compile_source
"""
import os, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bayer_low_light_image_enhancement_amd import build as B

args = sys.argv[1:]
name, src = args[0], args[1]
rest = args[2:]
patch, raw, flags = None, False, []
i = 0
while i < len(rest):
    if rest[i] == "--patch":
        patch = os.path.abspath(rest[i + 1]); i += 2
    elif rest[i] == "--raw":
        raw = True; i += 1
    else:
        flags.append(rest[i]); i += 1
B.build_library()
out = os.path.join(REPO, "tools", "_diag")
os.makedirs(out, exist_ok=True)
path = os.path.join(B.CSRC, src)
tmp_src = None
if patch:
    tmp_src = os.path.join(B.CSRC, f"_variant_{name}_{src}")       # next to the headers it includes; removed below
    shutil.copy(path, tmp_src)
    subprocess.check_call(["patch", "-s", "-p0", tmp_src, patch])
    path = tmp_src
obj = os.path.join(out, f"{name}_{src.replace('.hip', '.o')}")
try:
    if raw:
        subprocess.check_call([B._hipcc(), *B.FLAGS, *flags, "-c", path, "-o", obj])
    else:
        n = B.compile_source(path, obj, extra_flags=flags)
        print(f"{n} packed-f32 operand-select instruction(s) commuted")
finally:
    if tmp_src and os.path.exists(tmp_src):
        os.remove(tmp_src)
    for junk in (tmp_src + ".orig", tmp_src + ".rej") if tmp_src else ():
        if os.path.exists(junk):
            os.remove(junk)
objs = [obj if s == src else os.path.join(B.CSRC, s.replace(".hip", ".o")) for s in B.SOURCES]
lib = os.path.join(out, f"lib_{name}.so")
subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", lib])
print(lib)
