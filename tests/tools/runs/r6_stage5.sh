#!/bin/bash
# round 6, run 5: STAGE_PROBE build: columns of "avg" lines are: issued | staged | x landed (barrier col) | partial-sum barrier (first col) | scale known (dots col) | exit
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
GGML_HIP_FUSE_ATTN=0 timeout 200 python tests/tools/timeline.py 7b 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -14 > gpurun_out/r6/run5_stage_probe.txt
cat gpurun_out/r6/run5_stage_probe.txt
