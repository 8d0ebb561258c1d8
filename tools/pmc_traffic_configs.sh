#!/bin/bash
# usage (on the GPU box, via gpurun): tools/pmc_traffic_configs.sh <tag> [configs, default "3 4 5"]
# HBM traffic of BASELINE configs 3-5 per kernel: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, each in a pass of its own
# (MI355X_MICROARCH.md, HBM section), on one step of bench.py --config N at a batch that still exceeds the 256 MiB Infinity
# Cache several times over; tools/pmc_traffic_configs.py turns the passes into profiles/pmc_traffic_c<N>.json (per frame).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
declare -A UTTS=( [3]=3000 [4]=2000 [5]=10000 )
for c in ${2:-3 4 5}; do
  BENCH="python $R/bench.py --config $c --utts ${UTTS[$c]} --steps 1 --warmup 1 --no-cpu-baseline"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c$c/pf -- $BENCH > $O/c${c}_pf.json 2> $O/c${c}_pf.log
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c$c/pw -- $BENCH > $O/c${c}_pw.json 2> $O/c${c}_pw.log
  python $R/tools/pmc_traffic_configs.py $O $c > $O/pmc_traffic_c$c.json 2> $O/c${c}_sum.log
  cut -c1-400 $O/pmc_traffic_c$c.json
  rm -rf $O/c$c
done
