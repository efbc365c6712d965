"""gfx950 ISA checks and the one ISA rewrite of the build (host-side tooling, no GPU needed).

**The hazard** (``tools/ubench/pk_opsel_probe.hip``, ``profiles/r03_pk_opsel_probe.txt``): on MI355X a packed-f32
VALU instruction -- ``v_pk_fma_f32`` / ``v_pk_mul_f32`` / ``v_pk_add_f32`` -- whose LOW lane reads src0 from the low half and src1
from the HIGH half of their register pairs (``op_sel:[0,1,...]``) loses the src0 x src1 term in lanes 48-63 when another wave on
the same SIMD starts a burst of ``v_mfma_f32_16x16x32_bf16`` at that moment (3e6 wrong results in 8e10; every other operand
select, incl. ``op_sel:[1,0,...]``, is exact).  hipcc 7.2 emits the form from ordinary C++ and knows no hazard for it; it is what
made round 2's "prefetch" builds of the level-0 fused kernels return wrong tiles.

**The rewrite**: multiplication and addition commute, so ``op_sel:[0,1,c]`` with src0 and src1 exchanged is the exact
``op_sel:[1,0,c]`` form (every per-operand modifier list is permuted with the operands).  ``build.py`` compiles every source
to gfx950 assembly, applies :func:`commute_vulnerable`, and assembles the result; :func:`scan_library` (run by
``tests/test_isa_guard.py`` on the shipped ``.so``) fails the suite if such an instruction -- or scratch in a dispatched kernel
-- is left.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
_PK_F32 = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\s+(.*)$")
_MOD = re.compile(r"\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]")


def _split_operands(rest: str):
    """'v[0:1], v[2:3], s[4:5], v[6:7] op_sel:[0,1,1] ...' -> (['v[0:1]', ...], ' op_sel:[0,1,1] ...')"""
    m = _MOD.search(rest)
    ops, mods = (rest[:m.start()], rest[m.start():]) if m else (rest, "")
    comment = ""
    for mark in ("//", ";"):
        if mark in ops:
            ops, comment = ops.split(mark, 1)[0], mark + ops.split(mark, 1)[1]
    return [o.strip() for o in ops.strip().rstrip(",").split(",") if o.strip()], mods, comment


def is_vulnerable(line: str) -> bool:
    """True for a packed-f32 instruction whose low lane takes src0.lo and src1.HI (``op_sel:[0,1,...]``)."""
    m = _PK_F32.match(line)
    if not m:
        return False
    sel = re.search(r"\bop_sel:\[([01]),([01])", m.group(2))
    return bool(sel) and sel.group(1) == "0" and sel.group(2) == "1"


def commute_line(line: str) -> str:
    """The same instruction with src0 and src1 (and the first two entries of every modifier list) exchanged."""
    m = _PK_F32.match(line)
    assert m, line
    ops, mods, comment = _split_operands(m.group(2))
    nsrc = 3 if m.group(1) == "v_pk_fma_f32" else 2
    assert len(ops) == nsrc + 1, line
    ops[1], ops[2] = ops[2], ops[1]

    def swap(mm):
        bits = mm.group(2).split(",")
        assert len(bits) == nsrc, line
        bits[0], bits[1] = bits[1], bits[0]
        return f"{mm.group(1)}:[{','.join(bits)}]"

    mods = _MOD.sub(swap, mods)
    indent = line[: len(line) - len(line.lstrip())]
    return f"{indent}{m.group(1)} {', '.join(ops)} {mods.strip()}".rstrip() + (f" {comment}" if comment else "")


def commute_vulnerable(asm_text: str):
    """Rewrite every vulnerable instruction of a gfx950 assembly listing; returns (text, number rewritten)."""
    out, n = [], 0
    for ln in asm_text.split("\n"):
        if is_vulnerable(ln):
            new = commute_line(ln)
            assert not is_vulnerable(new), (ln, new)
            out.append(new)
            n += 1
        else:
            out.append(ln)
    return "\n".join(out), n


# ------------------------------------------------------------------------------------------------- code-object inspection
def _tool(name: str) -> str:
    exe = os.path.join(LLVM_BIN, name)
    if not os.path.exists(exe):
        found = shutil.which(name)
        if not found:
            raise RuntimeError(f"{name} not found (needed to inspect the gfx950 code objects)")
        exe = found
    return exe


def extract_code_objects(path: str, outdir: str):
    """The gfx950 code objects embedded in a host object / shared library (one per translation unit)."""
    local = os.path.join(outdir, os.path.basename(path))
    shutil.copy(path, local)
    subprocess.run([_tool("llvm-objdump"), "--offloading", os.path.basename(local)], cwd=outdir, check=True, capture_output=True)
    return sorted(os.path.join(outdir, f) for f in os.listdir(outdir) if "amdgcn-amd-amdhsa--gfx950" in f)


def scan_library(path: str):
    """{kernel: {'scratch': bytes/lane, 'vgpr_spill': n, 'sgpr_spill': n, 'vgprs': n, 'vulnerable': [instruction text, ...]}}"""
    report = {}
    with tempfile.TemporaryDirectory(prefix="rf_isa_") as tmp:
        for co in extract_code_objects(path, tmp):
            notes = subprocess.run([_tool("llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
            cur = None
            for ln in notes.splitlines():
                t = ln.strip()
                if t.startswith(".name:"):
                    cur = t.split(":", 1)[1].strip()
                    report.setdefault(cur, {"scratch": 0, "vgpr_spill": 0, "sgpr_spill": 0, "vgprs": 0, "vulnerable": []})
                elif cur and t.startswith(".private_segment_fixed_size:"):
                    report[cur]["scratch"] = int(t.split(":")[1])
                elif cur and t.startswith(".vgpr_spill_count:"):
                    report[cur]["vgpr_spill"] = int(t.split(":")[1])
                elif cur and t.startswith(".sgpr_spill_count:"):
                    report[cur]["sgpr_spill"] = int(t.split(":")[1])
                elif cur and t.startswith(".vgpr_count:"):
                    report[cur]["vgprs"] = int(t.split(":")[1])
            dis = subprocess.run([_tool("llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout
            cur = None
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
                if m:
                    cur = m.group(1)
                    continue
                if cur and is_vulnerable(ln):
                    report.setdefault(cur, {"scratch": 0, "vgpr_spill": 0, "sgpr_spill": 0, "vgprs": 0, "vulnerable": []})
                    report[cur]["vulnerable"].append(ln.split("//")[0].strip())
    return report
