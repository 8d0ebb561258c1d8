#!/bin/bash
# round-4 GPU call 13: the general cSpectral operator (device == oracle, plugin runs of avec2011 / avec2013 / emo_large / MediaEval)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_run13
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_spectral_general.py -m gpu -x -q > $O/pytest_spec.txt 2>&1
echo "pytest spectral rc=$?" | tee -a $O/pytest_spec.txt
tail -8 $O/pytest_spec.txt | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_plugin.py -m gpu -x -q -k "general_spectral or compare_spectral or option_sets" > $O/pytest_plug.txt 2>&1
echo "pytest plugin rc=$?" | tee -a $O/pytest_plug.txt
tail -8 $O/pytest_plug.txt | cut -c1-300
