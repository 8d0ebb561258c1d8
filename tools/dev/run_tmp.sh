cd $GRAFT_REPO_ROOT
timeout 100 python tools/plugin_config_sweep.py --only audspec/ 2>/dev/null | cut -c1-600
timeout 100 python -m pytest tests/test_gpu_plugin.py -q -k "plp or compare_spectral or option_sets or egemaps_whole" 2>&1 | tail -3
