#!/bin/bash
# usage (GPU box, via gpurun): tools/ubench/run_valu_replay.sh <tag>   -> gpurun_out/<tag>/valu_replay.json (+ counter passes)
# The code objects are generated here or on the build host by tools/ubench/valu_replay_gen.py tools/ubench/build/valu_replay_co
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; mkdir -p $O
B=$R/tools/ubench/build
[ -f $B/valu_replay_co/valu_replay_base.co ] || python $R/tools/ubench/valu_replay_gen.py $B/valu_replay_co > $O/gen.log 2>&1
[ -x $B/valu_replay ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/tools/ubench/valu_replay.hip -o $B/valu_replay
cp $B/valu_replay_co/valu_replay_info.json $O/
$B/valu_replay $B/valu_replay_co > $O/valu_replay.json 2> $O/valu_replay.err
cat $O/valu_replay.json
[ -x $B/valu_classes ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value $R/tools/ubench/valu_classes.hip -o $B/valu_classes
$B/valu_classes > $O/valu_classes.json 2> $O/valu_classes.err
cat $O/valu_classes.json
[ "${REPLAY_PMC:-1}" = 1 ] || exit 0
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc/p$i -- $B/valu_replay $B/valu_replay_co 1500 > $O/pmc_p$i.log 2>&1
done
python $R/tools/pmc_summary.py $O/pmc $O/valu_replay_pmc.txt > /dev/null 2>&1
rm -rf $O/pmc
grep -A8 "^==" $O/valu_replay_pmc.txt | head -120
