#!/bin/bash
# usage (GPU box): tools/dev/r06_streams_ab.sh <tag>   configs 4 and 5 with the chains' groups on one stream (the default for batches that
# fill the device) and forced onto the side streams (SMILEHIP_SERIAL=0): ms per step of each
set -u
R=$GRAFT_REPO_ROOT; T=$1; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
for C in 4 5; do for S in default 0; do
  if [ $S = default ]; then unset SMILEHIP_SERIAL; else export SMILEHIP_SERIAL=$S; fi
  timeout 900 python bench.py --config $C --no-cpu-baseline > $O/streams_c${C}_$S.json 2> $O/streams_c${C}_$S.err || tail -3 $O/streams_c${C}_$S.err
  python - $O/streams_c${C}_$S.json $C $S <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("config", sys.argv[2], "SMILEHIP_SERIAL", sys.argv[3], "ms_per_step", round(d["ms_per_step"], 2), "bit_identical", d.get("cells_bit_identical"), "of", d.get("cells_checked"))
except Exception as e:
    print("no line:", e)
PY
done; done 2>&1 | tee $O/streams_ab.txt
