#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_funcspec.py tests/test_gpu_plugin.py -q -x -m gpu -k "modulation or refuses or quotients or funcspec_custom" 2>&1 | tail -15
