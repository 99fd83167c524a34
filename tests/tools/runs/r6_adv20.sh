#!/bin/bash
# round 6, run 20: ADVICE fixes (speculation alternates, w16 release with siblings, weights generation): the affected suites
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_speculate_gpu.py tests/test_fused_timeout_gpu.py tests/test_concurrent_gpu.py tests/test_split_gpu.py tests/test_device_tools_gpu.py tests/test_kquant_plan_gpu.py -q -x 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -15 > gpurun_out/r6/run20_tests.txt
cat gpurun_out/r6/run20_tests.txt
