#!/bin/bash
# usage (on the GPU box): tools/ubench/run_variants_stats.sh <config> <utts> name1 name2 ...  -> per-kernel average durations (rocprofv3
# --kernel-trace --stats, SMILEHIP_SERIAL=1: each kernel alone on the device) of bench.py --config N --utts U for each private build
C=$1; U=$2; shift; shift
R=$PWD
if [ $C = 2 ]; then X="--no-configs --no-h2d"; else X="--config $C"; fi
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = default ]; then unset SMILEHIP_LIB; else export SMILEHIP_LIB=$R/tools/ubench/build/libsmilehip_$v.so; fi
  rm -rf /tmp/vs_$v
  SMILEHIP_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vs_$v -- python $R/bench.py $X --utts $U --steps 2 --warmup 1 --no-cpu-baseline > /tmp/vs_$v.json 2>/dev/null
  echo "== $v  $(python -c "import json; r=json.loads(open('/tmp/vs_$v.json').read().strip().splitlines()[-1]); print('step_ms', round(r['ms_per_step'],3), 'accuracy', r['accuracy'].get('pass'))")"
  python - $(find /tmp/vs_$v -name '*kernel_stats.csv' | head -1) <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("   %-52s calls %5s avg_us %10.1f total_ms %9.2f" % (r["Name"].replace("smilehip::", "").replace("void ", "")[:52], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
