#!/bin/bash
# round-4 GPU call 9: lld_gemaps_harm with a tile counter; the whole GPU test suite; config 5 stats (serial)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_run9
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_all.txt 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_all.txt
tail -6 $O/pytest_all.txt
cd /tmp && export TMPDIR=/tmp
export SMILEHIP_SERIAL=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5_serial.json 2> $O/stats.log
cut -c1-200 $O/bench_c5_serial.json
f=$(find $O/stats -name '*kernel_stats.csv' | head -1)
cp $f $O/c5_serial_kernel_stats.csv
cut -c1-130 $O/c5_serial_kernel_stats.csv | head -8
rm -rf $O/stats
