#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export CTX_SWEEP_MODEL=7b CTX_SWEEP_POSITIONS="300 700 1100 1800"
for cfg in "q8_0 1 1" "q8_0 2 1" "q5_1 1 1" "q5_1 1 0"; do set -- $cfg
echo "== 7b $1 FUSE_ATTN=$2 FUSE_HEADS=$3"
CTX_SWEEP_WTYPE=$1 GGML_HIP_FUSE_ATTN=$2 GGML_HIP_FUSE_HEADS=$3 timeout 400 python tests/tools/ctx_sweep.py 1 2>&1 | grep -v "^ROCm\|^Hostname\|amdgpu.ids" | tail -4
done
