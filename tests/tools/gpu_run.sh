cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_llama_gpu.py -x -q -m gpu -k "ggjt" 2>&1 | tail -5
