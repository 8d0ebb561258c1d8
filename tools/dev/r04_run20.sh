#!/bin/bash
cd /root/repo; O=/root/repo/gpurun_out/r20; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in fused nofuse; do
  if [ $v = nofuse ]; then export SMILEHIP_NO_FUSED_DELTA=1; else unset SMILEHIP_NO_FUSED_DELTA; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$v -- python /root/repo/bench.py --steps 100 --warmup 30 --no-cpu-baseline > $O/bench_$v.json 2> $O/stats_$v.log
  f=$(find $O/stats_$v -name '*kernel_stats.csv' | head -1); cp $f $O/${v}_kernel_stats.csv; rm -rf $O/stats_$v
  cut -c1-160 $O/${v}_kernel_stats.csv | head -4
done
