#!/usr/bin/env python3
"""usage: profile_summary.py gpurun_out/<tag> profiles/<prefix>
Copies what the judge reads from a tools/profile_run.sh run into profiles/: the bench lines, rocprofv3's
kernel stats of the same command, the per-launch durations of the timed steps taken from the kernel
trace (rocprof's own average also contains the warm-up launches, which run at lower clocks), the
counter summary and the HBM traffic derived from FETCH_SIZE / WRITE_SIZE."""
import csv
import glob
import json
import os
import shutil
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    shutil.copy(os.path.join(src, "bench.json"), dst + "_bench.json")
    shutil.copy(os.path.join(src, "stats_bench.json"), dst + "_bench_under_rocprof.json")
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, dst + "_kernel_stats.csv")
    shutil.copy(os.path.join(src, "pmc_summary.txt"), dst + "_pmc.txt")
    trace = glob.glob(os.path.join(src, "stats", "**", "*kernel_trace.csv"), recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(trace)):
        per.setdefault(r["Kernel_Name"].split("(")[0], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    bench = json.loads(open(os.path.join(src, "stats_bench.json")).read().strip().split("\n")[-1])
    steps = bench["steps"]
    lines = ["# per-launch kernel durations (ms) from rocprofv3 --kernel-trace of `python bench.py --no-cpu-baseline`",
             f"# bench line of that run: kernel_ms {bench['roofline']['kernel_ms']:.4f} (HIP events, the {steps} timed steps), "
             f"delta_kernel_ms {bench['roofline']['delta_kernel_ms']:.4f}"]
    for k, d in per.items():
        if "smilehip" not in k:
            continue
        timed = d[-steps:]
        lines.append(f"{k}: {len(d)} launches, mean of all {sum(d) / len(d):.4f}, mean of the last {len(timed)} (timed steps) "
                     f"{sum(timed) / len(timed):.4f}, min {min(d):.4f}, max {max(d):.4f}")
    open(dst + "_kernel_durations.txt", "w").write("\n".join(lines) + "\n")
    # HBM traffic of the fused kernel (MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request on gfx950 -> x2; KB units)
    fetch = write = None
    cur = None
    for ln in open(os.path.join(src, "pmc_summary.txt")):
        if ln.startswith("=="):
            cur = ln
        elif cur and "lld_mfcc512" in cur:
            if "FETCH_SIZE" in ln:
                fetch = float(ln.split()[2])
            if "WRITE_SIZE" in ln:
                write = float(ln.split()[2])
    frames = bench["config"].get("frames_per_gpu", bench["config"].get("frames_rank0"))
    t = {"kernel": "lld_mfcc512<13,7,true,true,true,false>",
         "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_run.sh)",
         "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
         "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE x1",
         "hbm_bytes_per_launch": (2 * fetch + write) * 1024.0,
         "algorithmic_bytes_per_launch": 372 * frames}
    json.dump(t, open(os.path.join(os.path.dirname(dst), "pmc_traffic.json"), "w"), indent=1)
    print(open(dst + "_kernel_durations.txt").read())
    print(json.dumps(t))


if __name__ == "__main__":
    main()
