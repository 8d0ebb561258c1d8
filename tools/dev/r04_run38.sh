#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_host_io.py tests/test_gpu_f32_input.py -q -x -m gpu 2>&1 | tail -3
