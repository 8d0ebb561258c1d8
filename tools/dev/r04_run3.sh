#!/bin/bash
# round-4 GPU call 3: cPitchJitter as independent runs of voiced frames (lld_jitter_runs) -- parity of the chains that
# contain it, kernel stats of config 4 at 1000 and 12 500 utterances, config 5 at 125 000
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_run3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_f0.py tests/test_gpu_compare_full.py tests/test_gpu_egemaps.py tests/test_gpu_is10.py tests/test_gpu_rates.py -m gpu -x -q > $O/pytest_f0.txt 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_f0.txt
tail -5 $O/pytest_f0.txt
cd /tmp && export TMPDIR=/tmp
for n in 1000 12500; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4_$n -- python $R/bench.py --config 4 --utts $n --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4_$n.json 2> $O/stats_c4_$n.log
cut -c1-200 $O/bench_c4_$n.json
f=$(find $O/stats_c4_$n -name '*kernel_stats.csv' | head -1)
cp $f $O/c4_${n}_kernel_stats.csv
cut -c1-150 $O/c4_${n}_kernel_stats.csv | head -10
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python $R/bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/stats_c5.log
cut -c1-200 $O/bench_c5.json
f=$(find $O/stats_c5 -name '*kernel_stats.csv' | head -1)
cp $f $O/c5_kernel_stats.csv
cut -c1-150 $O/c5_kernel_stats.csv | head -14
rm -rf $O/stats_c4_* $O/stats_c5
