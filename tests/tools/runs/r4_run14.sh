#!/bin/bash
# round-4 run 14: per-slot locks / thread-local current slot, perf counters, peer-copy event: the tests that exercise them
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4
timeout 900 python -X faulthandler -m pytest tests/test_concurrent_gpu.py tests/test_split_gpu.py tests/test_pipeline_2proc_gpu.py tests/test_llama_gpu.py tests/test_ops_gpu.py tests/test_fused_attn_gpu.py -q -x -m gpu -s > gpurun_out/r4/run14_pytest.txt 2>&1
grep -v "^$" gpurun_out/r4/run14_pytest.txt | head -40 | cut -c1-250
echo ...; grep -v "^$" gpurun_out/r4/run14_pytest.txt | tail -5 | cut -c1-250
