# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_prompt_plan_gpu.py -q -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -12
