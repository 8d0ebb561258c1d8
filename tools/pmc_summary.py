#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (one directory per counter pass) into
per-kernel averages per dispatch. usage: pmc_summary.py gpurun_out/pmc1 [out.txt]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    agg = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True)):
        per = defaultdict(lambda: defaultdict(float))     # (dispatch, kernel) -> counter -> sum over dims
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            per[(r["Dispatch_Id"], k)][r["Counter_Name"]] += float(r["Counter_Value"])
        for (d, k), cs in per.items():
            for c, v in cs.items():
                agg[k][c].append(v)
    lines = []
    for k, cs in agg.items():
        short = k.split("(")[0][-60:]
        lines.append(f"== {short}")
        for c, vs in sorted(cs.items()):
            lines.append(f"   {c:32s} avg/dispatch {sum(vs)/len(vs):18.1f}   (n={len(vs)})")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
