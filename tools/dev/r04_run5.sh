#!/bin/bash
# round-4 GPU call 5: phase timing of both jitter forms at 12 500 utterances (instrumented build, serial)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_run5
mkdir -p $O
cd $R
export SMILEHIP_SERIAL=1
python tools/ubench/phase_timing_jitter.py 12500 > $O/phase_runs.txt 2>&1
tail -9 $O/phase_runs.txt
SMILEHIP_JITTER_BY_UTT=1 python tools/ubench/phase_timing_jitter.py 12500 > $O/phase_utt.txt 2>&1
tail -9 $O/phase_utt.txt
python tools/ubench/phase_timing_jitter.py 1000 > $O/phase_runs_1000.txt 2>&1
tail -9 $O/phase_runs_1000.txt
