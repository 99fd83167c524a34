cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_llama_gpu.py -x -q -m gpu -k "two_sessions" 2>&1 | tail -12
