#!/bin/bash
# STAGE_PROBE build, probe 7: barrier col = LATEST wave's entry, first col = EARLIEST wave's entry, S col = which wave
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
export GGML_HIP_LIB=$GRAFT_REPO_ROOT/llm_amd/variants/libggml_hip_probe.so
TL_PROBE=7 GGML_HIP_FUSE_ATTN=0 timeout 200 python tests/tools/timeline.py 7b 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -30 > gpurun_out/r6/run14_stage_probe.txt
cat gpurun_out/r6/run14_stage_probe.txt
