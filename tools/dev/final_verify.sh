cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
timeout 720 python -m pytest tests -m gpu -x -q > gpurun_out/r03c/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03c/pytest_gpu.log
tail -3 gpurun_out/r03c/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03c/smoke.log 2>&1; tail -1 gpurun_out/r03c/smoke.log
sed -i 's#r03b#r03c#g' tools/dev/final_bench.sh
bash tools/dev/final_bench.sh
