#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests -q -x -m gpu -k "compare or func16 or is13 or other_sample_rates or big_sets or is09 or IS09 or egemaps or gemaps or rates or emobase or prosody" 2>&1 | grep -E "^FAILED|Error|assert|passed|failed" | head -12
for c in 3 4 5; do timeout 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print($c, round(d['ms_per_step'],1), round(d['roofline']['kernel_ms'],1))"; done
