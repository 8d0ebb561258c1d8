#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_mfcc.py -q -x -m gpu 2>&1 | tail -4
for v in default nofuse; do
  unset SMILEHIP_LIB SMILEHIP_NO_FUSED_DELTA
  if [ $v = nofuse ]; then export SMILEHIP_NO_FUSED_DELTA=1; elif [ $v != default ]; then export SMILEHIP_LIB=/root/repo/tools/ubench/build/libsmilehip_$v.so; fi
  timeout 300 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-configs --no-h2d 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('$v', round(d['ms_per_step'],4), round(r['kernel_ms'],4))"
done
