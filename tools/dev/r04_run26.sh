#!/bin/bash
cd /root/repo
for v in default early; do
  if [ $v = early ]; then export SMILEHIP_EARLY_FORK=1; else unset SMILEHIP_EARLY_FORK; fi
  echo "$v: $(timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['ms_per_step'],1))")"
done
