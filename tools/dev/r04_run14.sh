#!/bin/bash
# segments algorithms: device vs oracle, plugin taps, avec sets
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_funcspec.py tests/test_gpu_plugin.py -q -x -m gpu -k "segment or avec or refuses or rejects or general_spectral or functionals" 2>&1 | tail -25 > gpurun_out/r14_tests.log
cat gpurun_out/r14_tests.log
