cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
timeout 720 python -m pytest tests -m gpu -q > gpurun_out/r03d/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03d/pytest_gpu.log
grep -n "^FAILED\|passed\|failed\|pytest rc" gpurun_out/r03d/pytest_gpu.log | cut -c1-300 | head -20
grep -n "^E  .*ERR\|^E  .*differ\|^E  .*Assertion" gpurun_out/r03d/pytest_gpu.log | cut -c1-500 | head -12
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03d/smoke.log 2>&1; tail -1 gpurun_out/r03d/smoke.log
timeout 200 python tools/bench_stage4.py > gpurun_out/r03d/stage4.jsonl 2>/dev/null; cut -c1-160 gpurun_out/r03d/stage4.jsonl
