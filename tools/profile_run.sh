#!/bin/bash
# usage: tools/profile_run.sh <tag>  (run on the GPU box via gpurun): the default bench line, the same
# command under rocprofv3 --kernel-trace --stats, the SQ/LDS counter passes and the two HBM-traffic
# passes (FETCH_SIZE, WRITE_SIZE: each alone, as MI355X_MICROARCH.md prescribes).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R && python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline > $O/stats_bench.json 2> $O/stats.log
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc/pf -- $BENCH > $O/pf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc/pw -- $BENCH > $O/pw.log 2>&1
cd $R && ./tools/pmc_kernel.sh $1/pmc > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc $O/pmc_summary.txt > /dev/null 2>&1
cat $O/bench.json | cut -c1-600
find $O -name '*kernel_stats.csv' | head -2
