#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
timeout 300 python tests/tools/wo_timeline.py 128 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tee gpurun_out/r6/wo_timeline_allheads.txt
