#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel name: mean of each counter per dispatch.

usage: python tools/pmc_summary.py <dir-or-csv> [substring filter]
"""
import collections
import csv
import glob
import os
import re
import sys


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*_counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    meta = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"].replace("void rf::", ""))
            if filt and filt not in k:
                continue
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
            meta[k] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Scratch_Size"])
    for k in sorted(acc):
        print(f"{k}  vgpr={meta[k][0]} agpr={meta[k][1]} sgpr={meta[k][2]} lds={meta[k][3]} scratch={meta[k][4]}")
        c = {n: v[0] / v[1] for n, v in acc[k].items()}
        n = next(iter(acc[k].values()))[1]
        wc = c.get("SQ_WAVE_CYCLES")
        for name in sorted(c):
            extra = f"  ({c[name] / wc:.3f} of WAVE_CYCLES)" if wc and name.startswith("SQ_") and name != "SQ_WAVE_CYCLES" else ""
            print(f"    {name:32s} {c[name]:16.1f}{extra}")
        print(f"    dispatches {n}")


if __name__ == "__main__":
    main()
