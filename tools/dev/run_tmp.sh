cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/is10
timeout 600 python -m pytest tests/test_gpu_is10.py -q > gpurun_out/is10/pytest_fam.log 2>&1
grep -n "^FAILED\|passed\|failed" gpurun_out/is10/pytest_fam.log | cut -c1-300 | head
grep -n "^E  " gpurun_out/is10/pytest_fam.log | cut -c1-400 | head -20
timeout 300 python tools/bench_stage4.py > gpurun_out/is10/stage4.jsonl 2> gpurun_out/is10/stage4.err; cat gpurun_out/is10/stage4.jsonl
