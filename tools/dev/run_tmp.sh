cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/is10
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/is10/pytest_all3.log 2>&1
grep -n "^FAILED\|passed\|failed" gpurun_out/is10/pytest_all3.log | cut -c1-300 | head -30
grep -n "^E  .*ERR\|^E  .*Assert\|^E  .*differ" gpurun_out/is10/pytest_all3.log | cut -c1-700 | head -30
timeout 300 python tools/bench_sets.py --sets f0,compare_full --utts 1000 --steps 10 2>/dev/null | grep set | cut -c1-200
