#!/usr/bin/env python3
"""Diagnostic: build a stand-alone HIP program from EDITED device assembly.
    asm_exe.py <source.hip> <output exe> <python expression mapping the list of asm lines `L` to a new list>
Compiles with -save-temps -v, applies the edit to the gfx950 .s, re-runs hipcc's remaining steps."""
import os, shlex, subprocess, sys, tempfile, shutil, re
src, out, expr = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2]), sys.argv[3]
tmp = tempfile.mkdtemp(prefix="rf_asmexe_")
r = subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", src, "-o", out, "-save-temps", "-v"], cwd=tmp, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
steps = [shlex.split(ln) for ln in r.stderr.splitlines() if ln.startswith(' "')]
dev_s = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
L = open(os.path.join(tmp, dev_s)).read().splitlines()
N = eval(expr, {"L": L, "re": re})
open(os.path.join(tmp, dev_s), "w").write("\n".join(N) + "\n")
print(f"{len(N) - len(L)} lines added")
first = next(i for i, s in enumerate(steps) if "-cc1as" in s and dev_s in s)
for s in steps[first:]:
    rr = subprocess.run(s, cwd=tmp, capture_output=True, text=True)
    if rr.returncode:
        sys.exit(" ".join(s)[:300] + "\n" + rr.stderr[-3000:])
shutil.copy(os.path.join(tmp, dev_s), out + ".s")
shutil.rmtree(tmp)
print(out)
