#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/r30; mkdir -p $O
for mode in serial default; do
  if [ $mode = serial ]; then export SMILEHIP_SERIAL=1; else unset SMILEHIP_SERIAL; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python /root/repo/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5_$mode.json 2> $O/stats.log
  f=$(find $O/stats -name '*kernel_stats.csv' | head -1); cp $f $O/c5_${mode}_kernel_stats.csv; rm -rf $O/stats
  python3 -c "import json; d=json.loads(open('$O/bench_c5_$mode.json').read().strip().split(chr(10))[-1]); print('$mode', round(d['ms_per_step'],1))"
done
unset SMILEHIP_SERIAL
for i in 1 2; do python /root/repo/bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('plain', round(d['ms_per_step'],1))"; done
