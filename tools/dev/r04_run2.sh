#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_run2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -- python $R/bench.py --config 4 --utts 1000 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4_1000.json 2> $O/stats_c4.log
f=$(find $O/stats_c4 -name '*kernel_stats.csv' | head -1)
cp $f $O/c4_1000_kernel_stats.csv
cut -c1-150 $O/c4_1000_kernel_stats.csv | head -8
cd $R
( time timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
tail -3 $O/bench_default.time
cut -c1-400 $O/bench_default.json
timeout 600 python -m pytest tests/test_bench_ranks.py -m gpu -x -q > $O/pytest_ranks.txt 2>&1
tail -5 $O/pytest_ranks.txt
