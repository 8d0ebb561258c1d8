#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests -q -x -m gpu -k "f0 or compare_full or egemaps or is10 or prosody" 2>&1 | tail -4
timeout 300 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-220
