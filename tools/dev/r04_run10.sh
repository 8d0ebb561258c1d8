#!/bin/bash
# round-4 GPU call 10: LPC tile padding, IEEE-float WAV ingest, the jitter forms test; HBM traffic of configs 3-5 (PMC passes)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_run10
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_f32_input.py tests/test_gpu_egemaps.py tests/test_gpu_compare_full.py tests/test_gemaps_subsets.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -4 $O/pytest.txt
bash tools/pmc_traffic_configs.sh r04_run10/pmc
cd /tmp && export TMPDIR=/tmp
export SMILEHIP_SERIAL=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5_serial.json 2> $O/stats.log
cut -c1-200 $O/bench_c5_serial.json
f=$(find $O/stats -name '*kernel_stats.csv' | head -1)
cp $f $O/c5_serial_kernel_stats.csv
grep -E "lpc|harm|frame20" $O/c5_serial_kernel_stats.csv | cut -c1-130
rm -rf $O/stats
