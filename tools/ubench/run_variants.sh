#!/bin/bash
# usage (on the GPU box): tools/ubench/run_variants.sh name1 name2 ...  -> kernel_ms of bench.py for each private build
for v in "$@"; do
  for rep in 1 2; do
    SMILEHIP_LIB=$PWD/tools/ubench/build/libsmilehip_$v.so python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'kernel_ms', round(r['roofline']['kernel_ms'],4), 'step_ms', round(r['ms_per_step'],4))"
  done
done
