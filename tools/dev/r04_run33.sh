#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/r33; mkdir -p $O
export SMILEHIP_SERIAL=1
run() {
  if [ $1 = default ]; then unset SMILEHIP_LIB; else export SMILEHIP_LIB=/root/repo/tools/ubench/build/libsmilehip_$1.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python /root/repo/bench.py --config $2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$1_$2.json 2> $O/stats.log
  f=$(find $O/stats -name '*kernel_stats.csv' | head -1); cp $f $O/$1_c$2_kernel_stats.csv; rm -rf $O/stats
}
run default 5
run fsnoslp 5
run default 4
run knoslp 4
