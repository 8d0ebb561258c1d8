#!/bin/bash
cd /root/repo
for c in 4 5; do for u in full 500 100; do for mode in serial default; do
  if [ $mode = serial ]; then export SMILEHIP_SERIAL=1; else unset SMILEHIP_SERIAL; fi
  if [ $u = full ]; then U=""; else U="--utts $u"; fi
  echo "c$c utts=$u $mode: $(python bench.py --config $c $U --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['ms_per_step'],2))")"
done; done; done
