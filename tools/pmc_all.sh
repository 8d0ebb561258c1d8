#!/bin/bash
# usage (on the GPU box): tools/pmc_all.sh  -> gpurun_out/pmc_<set>.txt: per-kernel counter averages for the bench workload
# and the other feature sets (tools/pmc_kernel.sh passes; FETCH_SIZE / WRITE_SIZE in their own passes)
R=$GRAFT_REPO_ROOT
bash $R/tools/pmc_kernel.sh pmc_mfcc > /dev/null 2>&1; cp $R/gpurun_out/pmc_mfcc/summary.txt $R/gpurun_out/pmc_mfcc.txt
for set in is09 compare_full egemaps; do
  BENCH_CMD="python $R/tools/bench_sets.py --sets $set --utts 1000 --steps 2 --func" bash $R/tools/pmc_kernel.sh pmc_$set > /dev/null 2>&1
  cp $R/gpurun_out/pmc_$set/summary.txt $R/gpurun_out/pmc_$set.txt
done
ls -la $R/gpurun_out/pmc_*.txt
