#!/bin/bash
cd /root/repo
for i in 1 2; do timeout 200 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-configs --no-h2d 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print(round(d['ms_per_step'],4), r['kernel'], round(r['kernel_ms'],4), round(r['delta_kernel_ms'],4))"; done
