#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5g
export CTX_SWEEP_MODEL=13b CTX_SWEEP_WTYPE=q5_1 CTX_SWEEP_POSITIONS="200 400 600 900 1100 1400 1800"
for cfg in "1 1" "2 1" "2 0"; do set -- $cfg
echo "== 13b FUSE_ATTN=$1 FUSE_HEADS=$2"
GGML_HIP_FUSE_ATTN=$1 GGML_HIP_FUSE_HEADS=$2 timeout 400 python tests/tools/ctx_sweep.py 1 2>&1 | grep -v "^ROCm\|^Hostname\|amdgpu.ids" | tail -8
done
export CTX_SWEEP_MODEL=65b CTX_SWEEP_WTYPE=q8_0 CTX_SWEEP_POSITIONS="200 600 1100 1800"
for cfg in "1 1" "2 1" "2 0"; do set -- $cfg
echo "== 65b FUSE_ATTN=$1 FUSE_HEADS=$2"
GGML_HIP_FUSE_ATTN=$1 GGML_HIP_FUSE_HEADS=$2 timeout 600 python tests/tools/ctx_sweep.py 1 2>&1 | grep -v "^ROCm\|^Hostname\|amdgpu.ids" | tail -5
done
