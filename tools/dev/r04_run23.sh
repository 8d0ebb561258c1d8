#!/bin/bash
cd /root/repo; O=gpurun_out/r23; mkdir -p $O
timeout 900 python -m pytest tests -q -x -m gpu -k "compare or func16 or is13 or plugin_compare or other_sample_rates or big_sets" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
SMILEHIP_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats -- python /root/repo/bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > /root/repo/$O/bench_c4.json 2> /root/repo/$O/stats.log
f=$(find /root/repo/$O/stats -name '*kernel_stats.csv' | head -1); cp $f /root/repo/$O/c4_serial_kernel_stats.csv; rm -rf /root/repo/$O/stats
cut -c1-150 /root/repo/$O/c4_serial_kernel_stats.csv | head -4
cut -c1-200 /root/repo/$O/bench_c4.json
