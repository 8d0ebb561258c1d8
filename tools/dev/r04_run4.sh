#!/bin/bash
# round-4 GPU call 4: why lld_jitter_runs is not faster -- run statistics, both forms alone on the device (SMILEHIP_SERIAL)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_run4
mkdir -p $O
cd $R
python tools/dev/jitter_probe.py 64 > $O/probe.txt 2>&1
cat $O/probe.txt | cut -c1-600
cd /tmp && export TMPDIR=/tmp
export SMILEHIP_SERIAL=1
for mode in runs utt; do
  if [ $mode = utt ]; then export SMILEHIP_JITTER_BY_UTT=1; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$mode -- python $R/bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4_$mode.json 2> $O/stats_$mode.log
  cut -c1-200 $O/bench_c4_$mode.json
  f=$(find $O/stats_$mode -name '*kernel_stats.csv' | head -1)
  cp $f $O/c4_serial_${mode}_kernel_stats.csv
  cut -c1-150 $O/c4_serial_${mode}_kernel_stats.csv | head -6
  rm -rf $O/stats_$mode
done
