#!/bin/bash
# round-4 run 29: 2 / 4 attention workgroups per head inside the wq|wk|wv launch for contexts beyond 512 positions (attn_consumer_split):
# its test + the fused / long-context tests, then the context sweep with the option on and off
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_fused_attn_gpu.py tests/test_llama_gpu.py tests/test_c3_gpu.py -q -m gpu -x -s > gpurun_out/r4/run29_pytest.txt 2>&1
grep -E "passed|failed|Error|error|attention workgroups|vs oracle|assert" gpurun_out/r4/run29_pytest.txt | tail -12 | cut -c1-250
for fh in 1 0 1 0; do
GGML_HIP_FUSE_HEADS=$fh CTX_SWEEP_NBATCH=8 timeout 600 python tests/tools/ctx_sweep.py 1 > gpurun_out/r4/r04_ctx_sweep_fh$fh.txt 2>&1; echo "fuse_heads=$fh: $(tail -10 gpurun_out/r4/r04_ctx_sweep_fh$fh.txt | tr '\n' ' ')"
done
