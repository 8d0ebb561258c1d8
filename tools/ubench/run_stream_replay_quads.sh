#!/bin/bash
# usage (GPU box, via gpurun): tools/ubench/run_stream_replay_quads.sh <tag>  -> gpurun_out/<tag>/stream_replay_quads.json
# The frame loops of the three sixteen-lanes-per-frame kernels (IS09, ComParE groups A+B, eGeMAPS 20 ms) and of lld_f0_spec replayed: their vector
# instructions alone (every block of the loop in file order; the static count is within 2 - 8 % of the counters' dynamic count per
# pass), at the kernels' own 3 waves per SIMD -- the issue time their instruction streams need.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; mkdir -p $O
B=$R/tools/ubench/build
[ -x $B/stream_replay_run ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-result $R/tools/ubench/stream_replay_run.hip -o $B/stream_replay_run
: > $O/stream_replay_quads.json
while read stem kern name; do
  asm=$B/stream_co/lld_$stem.s
  [ -f $asm ] || { cmd=$(make -C $R/opensmile_amd/csrc -n -B lld_$stem.o | grep -m1 "hipcc.* -c "); (cd $R/opensmile_amd/csrc && mkdir -p $B/stream_co && ${cmd/-c lld_$stem.hip -o lld_$stem.o/-S --cuda-device-only lld_$stem.hip -o $asm} 2>/dev/null); }
  read first last nv < <(python $R/tools/ubench/find_main_loop.py $asm $kern)
  python $R/tools/ubench/stream_replay_gen.py $stem $kern $first $last $B/stream_co $name --block 64 --lds 0 --vgprs 168 --valu-only 1 > $O/gen_$name.log 2>&1
  nvr=$(python -c "import json; print(json.load(open('$B/stream_co/${name}_info.json'))['valu'])")
  $B/stream_replay_run $B/stream_co $name 3072 64 400 $nvr | grep "_valu" | sed "s/^{/{\"loop\": \"$first .. $last\", /" >> $O/stream_replay_quads.json
done <<'L'
is09 _ZN8smilehip19lld_is09_frame_quadILi25ELb1EEEvNS_9LldParamsENS_10Is09ParamsE is09_quad
compare _ZN8smilehip22lld_compare_frame_quadENS_9LldParamsENS_13CompareParamsEi compare_quad
gemaps _ZN8smilehip23lld_gemaps_frame20_quadILi96EEEvNS_9LldParamsENS_12GemapsParamsEi frame20_quad
f0 _ZN8smilehip11lld_f0_specILb1ELb1EEEvNS_9LldParamsENS_8F0ParamsE f0_spec
L
# lld_f0_cand9's frame loop without the blocks of the paths the chain does not take (greedyPeakAlgo = 0's scan, the serial mean), 4 waves per SIMD
asm=$B/stream_co/lld_f0.s
K9=_ZN8smilehip12lld_f0_cand9ENS_9LldParamsENS_8F0ParamsE
read first last nv < <(python $R/tools/ubench/find_main_loop.py $asm $K9)
python $R/tools/ubench/stream_replay_gen.py f0 $K9 $first $last $B/stream_co f0_cand9 --block 64 --lds 0 --vgprs 128 --valu-only 1 \
    --skip ${CAND9_SKIP:-.LBB0_158,.LBB0_162,.LBB0_166,.LBB0_170,.LBB0_174,.LBB0_178,.LBB0_182,.LBB0_186,.LBB0_284,.LBB0_286} > $O/gen_f0_cand9.log 2>&1
nvr=$(python -c "import json; print(json.load(open('$B/stream_co/f0_cand9_info.json'))['valu'])")
$B/stream_replay_run $B/stream_co f0_cand9 4096 64 1000 $nvr | grep "_valu" | sed "s/^{/{\"loop\": \"$first .. $last\", /" >> $O/stream_replay_quads.json
cat $O/stream_replay_quads.json
