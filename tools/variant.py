#!/usr/bin/env python3
"""Diagnostic builds: recompile ONE source with extra -D flags and link it with the regular objects
into tools/_diag/lib_<name>.so (select it with RF_LIB_PATH).  usage: variant.py <name> <source.hip> [-DFOO ...]"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bayer_low_light_image_enhancement_amd import build as B
name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build_library()
out = os.path.join(REPO, "tools", "_diag")
os.makedirs(out, exist_ok=True)
obj = os.path.join(out, f"{name}_{src.replace('.hip', '.o')}")
subprocess.check_call([B._hipcc(), *B.FLAGS, *flags, "-c", os.path.join(B.CSRC, src), "-o", obj])
objs = [obj if s == src else os.path.join(B.CSRC, s.replace(".hip", ".o")) for s in B.SOURCES]
lib = os.path.join(out, f"lib_{name}.so")
subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", lib])
print(lib)
