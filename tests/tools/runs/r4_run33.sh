#!/bin/bash
# round-4 run 33: A/B on one box: the tree's kernels (scores / exps / sums of the cached positions under the weight stream, in both
# attention consumers) against the previous commit's build (GGML_HIP_LIB=tests/tools/ab/libggml_hip_v2.so), context sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
for v in new old new old; do
if [ $v = old ]; then export GGML_HIP_LIB=$R/tests/tools/ab/libggml_hip_v2.so; else unset GGML_HIP_LIB; fi
CTX_SWEEP_NBATCH=8 timeout 600 python tests/tools/ctx_sweep.py 1 > gpurun_out/r4/r04_ctx_sweep_ab_$v.txt 2>&1; echo "$v: $(tail -10 gpurun_out/r4/r04_ctx_sweep_ab_$v.txt | tr '\n' ' ')"
done
unset GGML_HIP_LIB
timeout 600 python -X faulthandler -m pytest tests/test_fused_attn_gpu.py -q -m gpu -x 2>&1 | tail -2
