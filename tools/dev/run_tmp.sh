cd $GRAFT_REPO_ROOT
timeout 60 python tools/plugin_config_sweep.py --only emobase/emobase.conf 2>/dev/null | cut -c1-700
timeout 60 python tools/plugin_config_sweep.py --only emo_large 2>/dev/null | cut -c1-700
timeout 100 python -m pytest tests/test_gpu_plugin.py tests/test_gpu_stages.py tests/test_gpu_is09.py -q -k "is09 or acf or option_sets" 2>&1 | tail -3
