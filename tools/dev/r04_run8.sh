#!/bin/bash
# round-4 GPU call 8: config 5 (eGeMAPSv02) kernel stats, serial and default; plugin test after the HNR fix
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_run8
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_plugin.py -m gpu -x -q > $O/pytest_plugin.txt 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_plugin.txt
tail -3 $O/pytest_plugin.txt
cd /tmp && export TMPDIR=/tmp
for mode in serial default; do
  if [ $mode = serial ]; then export SMILEHIP_SERIAL=1; else unset SMILEHIP_SERIAL; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$mode -- python $R/bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5_$mode.json 2> $O/stats_$mode.log
  cut -c1-200 $O/bench_c5_$mode.json
  f=$(find $O/stats_$mode -name '*kernel_stats.csv' | head -1)
  cp $f $O/c5_${mode}_kernel_stats.csv
  cut -c1-130 $O/c5_${mode}_kernel_stats.csv | head -16
  rm -rf $O/stats_$mode
done
