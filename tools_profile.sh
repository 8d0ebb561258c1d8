#!/bin/bash
# usage: tools_profile.sh <tag>  (run on the GPU box via gpurun): bench line + rocprofv3 kernel stats + PMC passes
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R && python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/stats_bench.json 2> $O/stats.log
cd $R && ./tools_pmc.sh $1/pmc > /dev/null 2>&1
find $O -name '*stats*.csv' | head
