#!/bin/bash
# usage: tools/dev/r04_measure.sh <tag>  (GPU box): the round's evidence in one call -- the whole GPU test suite + smoke, the
# default bench line (configs 2-5, PCIe figures, CPU baselines), the driver's short form, kernel stats of every config (default
# and serial), the config-2 PMC passes (profile_run.sh's), HBM traffic of configs 3-5, the plugin's config sweep.
set -u
R=$GRAFT_REPO_ROOT
T=$1
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -n "^FAILED\|passed\|failed\|pytest rc" $O/pytest_gpu.log | cut -c1-300 | head -10
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 1500 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
cut -c1-250 $O/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-h2d > $O/bench_c2_short.json 2>> $O/bench_default.err
cut -c1-250 $O/bench_c2_short.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python $R/bench.py --no-cpu-baseline --no-configs --no-h2d > $O/stats_c2.json 2> $O/stats_c2.log
cp $(find $O/stats_c2 -name '*kernel_stats.csv' | head -1) $O/c2_kernel_stats.csv; rm -rf $O/stats_c2
for c in 3 4 5; do
  for mode in default serial; do
    if [ $mode = serial ]; then export SMILEHIP_SERIAL=1; else unset SMILEHIP_SERIAL; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c$c -- python $R/bench.py --config $c --no-cpu-baseline > $O/stats_c${c}_$mode.json 2> $O/stats_c$c.log
    cp $(find $O/stats_c$c -name '*kernel_stats.csv' | head -1) $O/c${c}_${mode}_kernel_stats.csv; rm -rf $O/stats_c$c
  done
done
unset SMILEHIP_SERIAL
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-h2d"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc/pf -- $BENCH > $O/pf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc/pw -- $BENCH > $O/pw.log 2>&1
cd $R && BENCH_CMD="$BENCH" ./tools/pmc_kernel.sh $T/pmc > /dev/null 2>&1
python $R/tools/pmc_summary.py $O/pmc $O/pmc_summary.txt > /dev/null 2>&1
rm -rf $O/pmc/p*/
bash tools/pmc_traffic_configs.sh $T/pmc345 > /dev/null 2>&1
timeout 1200 python tools/plugin_config_sweep.py > $O/plugin_config_sweep.jsonl 2> $O/sweep.err
grep -c identical $O/plugin_config_sweep.jsonl
ls $O
