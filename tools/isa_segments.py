#!/usr/bin/env python3
"""Instruction-class counts per basic block of one kernel in a hipcc -S listing.
usage: isa_segments.py <file.s> <mangled-name-substring> [min_mfma]"""
import re
import sys
from collections import Counter
src, name = sys.argv[1], sys.argv[2]
minm = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and name in l.split(':')[0] and re.match(r'^_Z\w+:', l))
seg, out = Counter(), []
for i in range(start + 1, len(lines)):
    l = lines[i].strip()
    if l.startswith("s_endpgm"):
        break
    if not l or l.startswith(";") or l.startswith("."):
        if l.startswith(".LBB") and sum(seg.values()):
            out.append((i, dict(seg))); seg = Counter()
        continue
    op = l.split()[0]
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op == "s_barrier":
        seg["barrier" if op == "s_barrier" else "branch"] += 1
        out.append((i, dict(seg))); seg = Counter(); continue
    k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "wait" if op.startswith("s_waitcnt") else
         "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer_", "global_", "flat_", "scratch_")) else op)
    seg[k] += 1
for i, s in out:
    if s.get("mfma", 0) >= minm:
        print(i - start, s)
