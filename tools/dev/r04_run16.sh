#!/bin/bash
cd /root/repo; O=gpurun_out/r16; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_is09.py -q -x -m gpu 2>&1 | tail -3
for v in default nf; do
  if [ $v = default ]; then unset SMILEHIP_LIB; else export SMILEHIP_LIB=/root/repo/tools/ubench/build/libsmilehip_$v.so; fi
  echo $v; timeout 300 python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c90-200
done
