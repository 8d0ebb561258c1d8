#!/bin/bash
# round-4 GPU call 6: lld_jitter_runs as persistent waves with an item counter -- parity, phase timing, kernel stats (serial + default)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_run6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_f0.py tests/test_gpu_compare_full.py tests/test_gpu_egemaps.py tests/test_gpu_is10.py -m gpu -x -q > $O/pytest_f0.txt 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_f0.txt
tail -3 $O/pytest_f0.txt
SMILEHIP_SERIAL=1 python tools/ubench/phase_timing_jitter.py 12500 > $O/phase_runs.txt 2>&1
tail -8 $O/phase_runs.txt
cd /tmp && export TMPDIR=/tmp
for mode in serial default; do
  if [ $mode = serial ]; then export SMILEHIP_SERIAL=1; else unset SMILEHIP_SERIAL; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$mode -- python $R/bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4_$mode.json 2> $O/stats_$mode.log
  cut -c1-200 $O/bench_c4_$mode.json
  f=$(find $O/stats_$mode -name '*kernel_stats.csv' | head -1)
  cp $f $O/c4_${mode}_kernel_stats.csv
  cut -c1-150 $O/c4_${mode}_kernel_stats.csv | head -8
  rm -rf $O/stats_$mode
done
