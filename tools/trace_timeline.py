"""usage: trace_timeline.py <kernel_trace.csv> [end-kernel-substring]: start / end (ms) of every kernel of the last iteration
(the dispatches between the last two kernels matching the substring; default lld_gemaps_tail), in start order."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "lld_gemaps_tail"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
start, end = idx[-2] + 1, idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
agg = {}
for r in rows[start:end + 1]:
    n = r["Kernel_Name"].split("(")[0].replace("smilehip::", "").replace("void ", "")
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if n in agg:
        agg[n][1] = max(agg[n][1], e); agg[n][2] += 1; agg[n][3] += e - s
    else:
        agg[n] = [s, e, 1, e - s]
for n, (s, e, c, busy) in agg.items():
    print(f"{n:36s} {s:8.2f} .. {e:8.2f} ms  x{c:<3d} busy {busy:7.2f} ms")
