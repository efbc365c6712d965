#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from one rocprofv3 PMC pass (SQ counters only, --kernel-trace):

  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 \
            SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE ...  -- python3 bench.py --no-cpu-baseline --no-profile --steps 2 --warmup 1

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): the share of SIMD-cycles of the dispatch in
which the matrix pipe was executing (SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over
the 8 XCDs -- MI355X_MICROARCH.md).  MOPS counters are in units of 512 flops.

usage: mfma_util.py <dir-or-counter_collection.csv> <out.json> [workload]
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def main():
    path, out_path = sys.argv[1], sys.argv[2]
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*_counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in files:
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void rf::", "").replace("void ", "").replace("rf::", ""))
            if k.startswith(("__amd", "at::", "void at::")):
                continue
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    out = {}
    for k in sorted(acc):
        c = {n: v[0] / v[1] for n, v in acc[k].items()}
        gui = c.get("GRBM_GUI_ACTIVE")
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
        rec = {"dispatches": next(iter(acc[k].values()))[1]}
        if gui and busy is not None:
            rec["mfma_busy_frac"] = round(busy / (gui / 8.0 * 1024.0), 4)
            rec["cycles_per_dispatch"] = round(gui / 8.0)
        for n, key in (("SQ_INSTS_MFMA", "mfma_insts"), ("SQ_INSTS_VALU", "valu_insts_incl_mfma")):
            if n in c:
                rec[key] = round(c[n])
        f32, b16 = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32"), c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16")
        if f32 is not None:
            rec["mfma_gflop_f32"] = round(f32 * 512 / 1e9, 3)
        if b16 is not None:
            rec["mfma_gflop_bf16"] = round(b16 * 512 / 1e9, 3)
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            for n, key in (("SQ_WAIT_ANY", "wave_frac_waiting"), ("SQ_WAIT_INST_ANY", "wave_frac_issue_stalled"), ("SQ_ACTIVE_INST_VALU", "wave_frac_valu_issue")):
                if n in c:
                    rec[key] = round(c[n] / c["SQ_WAVE_CYCLES"], 4)
        out[k] = rec
    wl = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
    json.dump({"source": f"rocprofv3 --kernel-trace --pmc (SQ / GRBM counters, one pass) over bench.py --workload {wl} --steps 2 --warmup 1", "workload": wl,
               "units": "per dispatch (mean over the dispatches of that kernel name)", "kernels": out}, open(out_path, "w"), indent=1)
    for k, v in out.items():
        if "mfma_busy_frac" in v:
            print(f"{k:46s} mfma busy {v['mfma_busy_frac']:.3f}")


if __name__ == "__main__":
    main()
