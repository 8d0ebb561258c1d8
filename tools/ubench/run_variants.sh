#!/bin/bash
# usage (on the GPU box): tools/ubench/run_variants.sh <config 2..5> name1 name2 ...  -> kernel_ms / step / accuracy of bench.py --config N for
# each private build tools/ubench/build/libsmilehip_<name>.so (variant_any.sh); "default" = the product library
C=$1; shift
if [ $C = 2 ]; then X="--no-configs --no-h2d"; else X="--config $C"; fi
for v in "$@"; do
  if [ $v = default ]; then unset SMILEHIP_LIB; else export SMILEHIP_LIB=$PWD/tools/ubench/build/libsmilehip_$v.so; fi
  python bench.py $X --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', 'kernel_ms', round(r['roofline']['kernel_ms'],4), 'rest_ms', round(r['roofline'].get('rest_of_step_ms') or 0,3), 'step_ms', round(r['ms_per_step'],4), 'accuracy', r['accuracy'].get('pass'), r['accuracy'].get('cells_bit_identical'), r['accuracy'].get('cells'))"
done
