cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_f0.py -q -k "viterbi" 2>&1 | tail -8 | cut -c1-300
