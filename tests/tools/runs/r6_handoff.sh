#!/bin/bash
# round 6: the intra-XCD hand-off probe (tests/tools/handoff_probe.hip, built into build/ before the call) + the K plan tests on the reverted tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
timeout 120 build/handoff_probe > gpurun_out/r6/handoff_probe.txt 2>&1; cat gpurun_out/r6/handoff_probe.txt
timeout 900 python -m pytest tests/test_kquant_plan_gpu.py -x -q 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -5
timeout 600 python -m pytest tests/test_fused_attn_gpu.py -x -q -k "wo_tail or 7b_width" 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -5
