#!/bin/bash
# round-4 GPU call 1: the F0 front end after the scratch diet -- parity tests of the chains that contain it, kernel stats
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_run1
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_f0.py tests/test_gpu_compare_full.py tests/test_gpu_egemaps.py tests/test_gpu_rates.py -m gpu -x -q > $O/pytest_f0.txt 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_f0.txt
tail -5 $O/pytest_f0.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -- python $R/bench.py --config 4 --utts 1000 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4_1000.json 2> $O/stats_c4.log
cat $O/bench_c4_1000.json | cut -c1-300
f=$(find $O/stats_c4 -name '*kernel_stats.csv' | head -1)
cp $f $O/c4_1000_kernel_stats.csv
cut -c1-150 $O/c4_1000_kernel_stats.csv | head -16
