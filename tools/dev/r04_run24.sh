#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/r24; mkdir -p $O
export SMILEHIP_SERIAL=1
for v in cand4; do
  export SMILEHIP_LIB=/root/repo/tools/ubench/build/libsmilehip_$v.so
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python /root/repo/bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$v.json 2> $O/stats.log
  f=$(find $O/stats -name '*kernel_stats.csv' | head -1); cp $f $O/${v}_kernel_stats.csv; rm -rf $O/stats
  grep "f0_cand\|f0_spec\|f0_sweep" $O/${v}_kernel_stats.csv | cut -c1-140
done
