#!/bin/bash
# round-4 GPU call 7: jitter -- energy chains in rounds of 16 terms, wave capacity from the plan's minimum pitch
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_run7
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_f0.py tests/test_gpu_compare_full.py tests/test_gpu_egemaps.py tests/test_gpu_is10.py tests/test_gpu_plugin.py -m gpu -x -q > $O/pytest_f0.txt 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_f0.txt
tail -3 $O/pytest_f0.txt
SMILEHIP_SERIAL=1 python tools/ubench/phase_timing_jitter.py 12500 > $O/phase_runs.txt 2>&1
tail -8 $O/phase_runs.txt
cd /tmp && export TMPDIR=/tmp
for mode in serial; do
  export SMILEHIP_SERIAL=1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$mode -- python $R/bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4_$mode.json 2> $O/stats_$mode.log
  cut -c1-200 $O/bench_c4_$mode.json
  f=$(find $O/stats_$mode -name '*kernel_stats.csv' | head -1)
  cp $f $O/c4_${mode}_kernel_stats.csv
  cut -c1-150 $O/c4_${mode}_kernel_stats.csv | head -4
  rm -rf $O/stats_$mode
done
