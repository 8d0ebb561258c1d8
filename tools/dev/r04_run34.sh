#!/bin/bash
cd /root/repo
timeout 300 python tools/ubench/phase_timing_gm.py 2>&1 | tail -9
