#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export CTX_SWEEP_POSITIONS="700 1100 1800"
for cfg in "13b q4_0 1" "13b q4_0 0" "13b q8_0 1" "13b q8_0 0" "7b q5_1 1" "7b q5_1 0"; do set -- $cfg
echo "== $1 $2 FUSE_ATTN=2 FUSE_HEADS=$3"
CTX_SWEEP_MODEL=$1 CTX_SWEEP_WTYPE=$2 GGML_HIP_FUSE_ATTN=2 GGML_HIP_FUSE_HEADS=$3 timeout 400 python tests/tools/ctx_sweep.py 1 2>&1 | grep -v "^ROCm\|^Hostname\|amdgpu.ids" | tail -3
done
