#!/usr/bin/env python3
"""Skeleton of a kernel's instruction stream -- global loads, s_waitcnt vmcnt, MFMA runs, branches, barriers -- from the gfx950
assembly hipcc emits for one source file (CPU container, no GPU needed).  It is how the software pipelines are checked: a
prefetch is intact when the loads of step k + 1 sit BEFORE the MFMA block of step k and the wait in front of that block leaves
them outstanding (vmcnt(N) with N >= their count).  Round 3 found gram2's prefetch compiled away three different ways
(DESIGN.md section 4); this listing is what showed each of them.

usage: tools/loop_waits.py <file.hip> <kernel-name-substring> [--full]
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "bayer_low_light_image_enhancement_amd", "csrc")


def main():
    src, pat = sys.argv[1], sys.argv[2]
    full = "--full" in sys.argv
    if not os.path.exists(src):
        src = os.path.join(CSRC, src)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "--cuda-device-only", "-S", "-o", out, src,
                        "-I", CSRC, "-I", os.path.join(REPO, "include")], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    found = False
    for m in re.finditer(r"^(_Z\w+):", text, re.M):
        if pat not in m.group(1):
            continue
        found = True
        body = text[m.start(): text.index(".Lfunc_end", m.start())].split("\n")
        print(m.group(1))
        n_mfma = n_load = n_other = 0

        def flush():
            nonlocal n_mfma, n_load, n_other
            if n_load:
                print(f"        [{n_load} global loads]")
            if n_mfma:
                print(f"        [{n_mfma} MFMA]")
            if n_other and full:
                print(f"        [{n_other} other]")
            n_mfma = n_load = n_other = 0

        for ln in body[1:]:
            s = ln.split(";")[0].strip()
            if not s or s.startswith("."):
                if re.match(r"^\.LBB\d+_\d+:", ln):
                    flush()
                    print("   ", ln.split(";")[0].strip(), ("; " + ln.split(";", 1)[1].strip()) if ";" in ln else "")
                continue
            if "v_mfma" in s:
                if n_load:
                    flush()
                n_mfma += 1
            elif s.startswith("global_load") or s.startswith("buffer_load"):
                if n_mfma:
                    flush()
                n_load += 1
            elif s.startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_barrier", "global_store", "buffer_store")):
                flush()
                print("       ", s[:72])
            else:
                n_other += 1
        flush()
    if not found:
        sys.exit(f"no kernel matching '{pat}' in {src}")


if __name__ == "__main__":
    main()
