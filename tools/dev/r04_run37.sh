#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_mfcc.py -q -x -m gpu -k "full_size" 2>&1 | tail -4
