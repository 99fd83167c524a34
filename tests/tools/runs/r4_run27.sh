#!/bin/bash
# round-4 run 27: the switch to the split attention at 768 positions for the fused path (512 for k_attn_decode / the K plan): suite, sweep check
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests -q -m gpu -x > gpurun_out/r4/suite27.txt 2>&1; grep -E "passed|failed|Error|error" gpurun_out/r4/suite27.txt | head -8 | cut -c1-300
timeout 600 python tests/tools/ctx_sweep.py 1 512 > gpurun_out/r4/r04_ctx_sweep_default.txt 2>&1; tail -12 gpurun_out/r4/r04_ctx_sweep_default.txt
