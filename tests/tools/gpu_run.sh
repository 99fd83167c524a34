cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_llama_gpu.py -q -x -k "snapshot or chain or argmax" 2>&1 | tail -5
