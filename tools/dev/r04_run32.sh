#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests -q -x -m gpu -k "compare or egemaps or gemaps or f0 or is10 or rates or big_sets or functionals" 2>&1 | grep -E "^FAILED|passed|failed" | head -5
for c in 4 5; do for u in full 500; do
  if [ $u = full ]; then U=""; else U="--utts $u"; fi
  echo "c$c utts=$u: $(python bench.py --config $c $U --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['ms_per_step'],2))")"
done; done
