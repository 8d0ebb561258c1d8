#!/usr/bin/env python3
"""usage: pmc_traffic_configs.py <dir of tools/pmc_traffic_configs.sh> <config>  -> JSON on stdout (profiles/pmc_traffic_c<N>.json).
Per kernel: FETCH_SIZE / WRITE_SIZE (KB, summed over the counter's instances) of every dispatch of the run (1 warm-up + 1 timed
step = 2 steps), per step and per frame of the batch. gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE tallies 128-byte
requests at 64 B, so it is doubled; that factor was calibrated on wide coalesced reads only (what the frame kernels do;
narrower reads are uncalibrated), WRITE_SIZE is taken as reported."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def passes(root, counter):
    per = defaultdict(float)
    n = defaultdict(set)
    for f in sorted(glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smilehip::", "")
            per[k] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    return per, {k: len(v) for k, v in n.items()}


def main():
    root, c = sys.argv[1], sys.argv[2]
    line = json.loads(open(os.path.join(root, f"c{c}_pf.json")).read().strip().splitlines()[-1])
    frames = line["config"]["frames_rank0"]
    steps = line["steps"] + line["warmup"]
    fetch, nf = passes(os.path.join(root, f"c{c}", "pf"), "FETCH_SIZE")
    write, _ = passes(os.path.join(root, f"c{c}", "pw"), "WRITE_SIZE")
    by = {}
    tot = 0.0
    for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
        fb = 2.0 * fetch.get(k, 0.0) * 1024.0 / steps / frames
        wb = write.get(k, 0.0) * 1024.0 / steps / frames
        if fb + wb < 0.5:
            continue
        by[k] = {"fetch_bytes_per_frame": round(fb, 1), "write_bytes_per_frame": round(wb, 1), "dispatches_per_step": nf.get(k, 0) // steps}
        tot += fb + wb
    print(json.dumps({
        "config": int(c), "workload": line["config"]["workload"], "utterances": line["config"]["utterances_per_gpu"], "frames": frames,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/pmc_traffic_configs.sh), every kernel of the step; "
                  "gfx950: FETCH_SIZE x2 (calibrated on wide coalesced reads, MI355X_MICROARCH.md), WRITE_SIZE as reported",
        "hbm_bytes_per_frame": round(tot, 1), "by_kernel": by}, indent=1))


if __name__ == "__main__":
    main()
