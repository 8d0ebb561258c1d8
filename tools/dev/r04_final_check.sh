#!/bin/bash
# the driver's end-of-round sequence on the final tree: GPU suite (-x), smoke, the default bench line
cd /root/repo; O=gpurun_out/r04_final_check; mkdir -p $O
( time timeout 1700 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench.err; tail -3 $O/bench.err; cut -c1-200 $O/bench_default.json
