cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sweep
timeout 420 python tools/plugin_config_sweep.py > gpurun_out/sweep/sweep.jsonl 2> gpurun_out/sweep/sweep.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/sweep/sweep.jsonl')]
ok=[r for r in rows if r['plugin_rc']==0 and r['identical']]
print(len(rows),'configs;', len(ok),'complete on the GPU operators and byte-identical')
for r in rows:
    if not (r['plugin_rc']==0 and r['identical']): print(r['conf'], r['plugin_rc'], r['identical'], [x[:60] for x in r['refused'][:1]])
PY
timeout 300 python -m pytest tests/test_gpu_plugin.py -q > gpurun_out/sweep/pytest_plugin.log 2>&1; grep -n "^FAILED\|passed\|failed" gpurun_out/sweep/pytest_plugin.log | head
