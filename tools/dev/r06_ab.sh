#!/bin/bash
# usage (GPU box): tools/dev/r06_ab.sh <tag> <config> <lib or -> ...   bench.py --config N with each library (- = the product's), the
# per-kernel milliseconds of each printed side by side (kernels_ms_per_step of the bench line: HIP events around every launch)
set -u
R=$GRAFT_REPO_ROOT; T=$1; C=$2; shift; shift
O=$R/gpurun_out/$T; mkdir -p $O; cd $R
for lib in "$@"; do
  name=$(basename $lib .so); [ "$lib" = - ] && name=product
  if [ "$lib" = - ]; then unset SMILEHIP_LIB; else export SMILEHIP_LIB=$R/$lib; fi
  if [ $C = 2 ]; then X="--no-configs --no-h2d"; else X="--config $C"; fi
  timeout 900 python bench.py $X --no-cpu-baseline > $O/ab_c${C}_$name.json 2> $O/ab_c${C}_$name.err || tail -3 $O/ab_c${C}_$name.err
  python - $O/ab_c${C}_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d.get("kernels_ms_per_step") or d["roofline"].get("kernels_ms_per_step") or {}
    print(sys.argv[2], "ms_per_step", round(d["ms_per_step"], 2), "bit_identical", d.get("cells_bit_identical"), "of", d.get("cells_checked"),
          {a: round(b, 2) for a, b in sorted(k.items(), key=lambda x: -x[1])[:12]})
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
done
