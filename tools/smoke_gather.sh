#!/bin/bash
# smilextract_hip --gather on ONE node: N ranks (one per GPU) gather their functionals rows to rank 0 over RCCL.
# usage: tools/smoke_gather.sh <n_ranks> <filelist> <out.arff>   (on a single-GPU box only N = 1 can run: RCCL refuses two
# ranks on one device)
N=${1:-1}; LIST=$2; OUT=$3
cd "$(dirname "$0")/.."
pids=()
for r in $(seq 0 $((N-1))); do
  ./opensmile_amd/smilextract_hip --set egemapsv02 -filelist "$LIST" -O "$OUT" --gather --rank $r --world $N --device $r \
    --master-addr 127.0.0.1 --master-port 29433 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
exit $rc
