#!/bin/bash
# round-4 GPU call 11: IS09 frame kernel, sixteen lanes per frame -- parity with the wave form, config 3 kernel stats of both
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_run11
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_is09.py -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.txt
tail -12 $O/pytest.txt | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for form in quad wave; do
  if [ $form = wave ]; then export SMILEHIP_IS09=wave; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$form -- python $R/bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c3_$form.json 2> $O/stats_$form.log
  cut -c1-220 $O/bench_c3_$form.json
  f=$(find $O/stats_$form -name '*kernel_stats.csv' | head -1)
  cp $f $O/c3_${form}_kernel_stats.csv
  cut -c1-130 $O/c3_${form}_kernel_stats.csv | head -5
  rm -rf $O/stats_$form
done
