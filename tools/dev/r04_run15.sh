#!/bin/bash
# IS09 quad kernel with register-resident spectra: parity, then config 3 at 3 and at 2 waves per SIMD
cd /root/repo; O=gpurun_out/r15; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_is09.py tests/test_gpu_ooura.py -q -x -m gpu 2>&1 | tail -15 > $O/tests.log; cat $O/tests.log
for v in w3 q2; do
  if [ $v = q2 ]; then export SMILEHIP_LIB=/root/repo/tools/ubench/build/libsmilehip_q2.so; else unset SMILEHIP_LIB; fi
  timeout 300 python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c3_$v.json 2> $O/bench_c3_$v.err
  cut -c1-260 $O/bench_c3_$v.json
done
