cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -s 2>&1 | tail -22
