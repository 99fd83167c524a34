#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_fused_attn_gpu.py tests/test_ref_branch_gpu.py tests/test_llama_gpu.py tests/test_c3_gpu.py -m gpu -q -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -4
export CTX_SWEEP_POSITIONS="700 1100 1800"
for cfg in "13b q4_0" "13b q8_0" "13b q5_1" "7b q5_1" "7b q8_0"; do set -- $cfg
echo "== $1 $2 defaults"
CTX_SWEEP_MODEL=$1 CTX_SWEEP_WTYPE=$2 timeout 400 python tests/tools/ctx_sweep.py 1 2>&1 | grep -v "^ROCm\|^Hostname\|amdgpu.ids" | tail -3
done
