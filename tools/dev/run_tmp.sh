cd $GRAFT_REPO_ROOT
timeout 100 python tools/plugin_config_sweep.py --only prosody/ 2>/dev/null | cut -c1-420
