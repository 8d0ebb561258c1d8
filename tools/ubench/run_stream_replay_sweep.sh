#!/bin/bash
# usage (GPU box, via gpurun): tools/ubench/run_stream_replay_sweep.sh <tag>  -> gpurun_out/<tag>/stream_replay_sweep.json
# lld_f0_sweep's pass-1 stretch (a block's first 14 bins: the labels below are those of the current compiler output -- check them with
# `grep -n "^.LBB13_" <outdir>/lld_f0.s` after a change of the kernel) replayed: vector instructions only, and with its scalar
# instructions / table loads / waits when the loads walk 1 KB (mask 0), 4 KB (mask 3) or the real 32 KB (mask 31) of the table,
# every workgroup at its own place -- at 1, 2 and 4 waves per SIMD (workgroups of one wave).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; mkdir -p $O
B=$R/tools/ubench/build
K=_ZN8smilehip12lld_f0_sweepILi9EEEvNS_8F0ParamsE
FIRST=${SWEEP_FIRST:-.LBB13_19}; LAST=${SWEEP_LAST:-.LBB13_21}
[ -x $B/stream_replay_run ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-result $R/tools/ubench/stream_replay_run.hip -o $B/stream_replay_run
: > $O/stream_replay_sweep.json
for m in 0 3 31; do
  [ -f $B/stream_co/sweep_p1_m$m"_scal.co" ] || python $R/tools/ubench/stream_replay_gen.py f0 $K $FIRST $LAST $B/stream_co sweep_p1_m$m --block 64 --lds 8192 --vgprs 128 \
      --init 's_mov_b64 s[34:35], {PTR}' --init 's_add_u32 s17, {CTR}, {WG}' --init "s_and_b32 s17, s17, $m" > $O/gen_m$m.log 2>&1
  nv=$(python -c "import json; print(json.load(open('$B/stream_co/sweep_p1_m${m}_info.json'))['valu'])")
  for b in 1024 2048 4096; do
    $B/stream_replay_run $B/stream_co sweep_p1_m$m $b 64 2000 $nv | sed "s/^{/{\"table_window_blocks\": $((m + 1)), /" >> $O/stream_replay_sweep.json
  done
done
cat $O/stream_replay_sweep.json
