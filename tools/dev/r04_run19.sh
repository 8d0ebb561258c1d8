#!/bin/bash
cd /root/repo; O=gpurun_out/r19; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mfcc.py -q -x -m gpu 2>&1 | tail -15 | tee $O/tests.log
timeout 300 python bench.py --steps 100 --warmup 30 --no-cpu-baseline 2>/dev/null | cut -c1-300 | tee $O/bench.txt
SMILEHIP_NO_FUSED_DELTA=1 timeout 300 python bench.py --steps 100 --warmup 30 --no-cpu-baseline 2>/dev/null | cut -c1-300 | tee $O/bench_nofuse.txt
