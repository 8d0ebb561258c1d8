#!/bin/bash
cd /root/repo; O=gpurun_out/r18; mkdir -p $O
timeout 300 python tools/ubench/phase_timing_cmp.py > $O/phase_cmp.txt 2>&1; tail -9 $O/phase_cmp.txt
