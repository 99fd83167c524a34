#!/bin/bash
# round 6, run 17: is the L2 warm-up of w1|w3 hit by the next launch?  gate's in-kernel stamps with / without
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
for R in 0 2048 4096; do
echo "== warm_rows $R"
GGML_HIP_WARM_ROWS=$R timeout 200 python tests/tools/timeline.py 7b 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | grep 'avg gate\|avg down\|gate15\|gap before'
done > gpurun_out/r6/run17_warm_gate.txt
cat gpurun_out/r6/run17_warm_gate.txt
