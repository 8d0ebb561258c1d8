#!/bin/bash
# full GPU suite + smoke after the delta-fused MFCC kernel and the IS09 register-resident quad kernel
cd /root/repo; O=gpurun_out/r22; mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
