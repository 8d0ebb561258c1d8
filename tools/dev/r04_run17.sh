#!/bin/bash
# IS09-related GPU tests after the register-resident quad kernel, kernel stats of config 3
cd /root/repo; O=gpurun_out/r17; mkdir -p $O
timeout 900 python -m pytest tests -q -x -m gpu -k "is09 or IS09 or ooura or emobase or smoke or option_sets or prosody" 2>&1 | tail -5 | tee $O/tests.log
cd /tmp && export TMPDIR=/tmp
export SMILEHIP_SERIAL=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats -- python /root/repo/bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline > /root/repo/$O/bench_c3.json 2> /root/repo/$O/stats.log
f=$(find /root/repo/$O/stats -name '*kernel_stats.csv' | head -1); cp $f /root/repo/$O/c3_serial_kernel_stats.csv; rm -rf /root/repo/$O/stats
cut -c1-150 /root/repo/$O/c3_serial_kernel_stats.csv | head -5
