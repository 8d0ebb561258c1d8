#!/bin/bash
# round-4 GPU call 12: jitter main loop on alternating registers; IS09 quad form incl. 32 ms frames; config 4 + 3 stats
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_run12
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_is09.py tests/test_gpu_compare_full.py tests/test_gpu_egemaps.py tests/test_gpu_f0.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -5 $O/pytest.txt | cut -c1-300
cd /tmp && export TMPDIR=/tmp
export SMILEHIP_SERIAL=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4_serial.json 2> $O/stats.log
cut -c1-200 $O/bench_c4_serial.json
f=$(find $O/stats -name '*kernel_stats.csv' | head -1)
cp $f $O/c4_serial_kernel_stats.csv
cut -c1-130 $O/c4_serial_kernel_stats.csv | head -5
rm -rf $O/stats
