#!/bin/bash
# round-5 run 5: the WO form of k_qkv_attn (wo + residual as the mat-vec workgroups' second phase): bit-identity tests, A/B decode rate
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fused_attn_gpu.py tests/test_fused_timeout_gpu.py tests/test_llama_gpu.py tests/test_c3_gpu.py -m gpu -q -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -25 > gpurun_out/r5/pytest_run5.txt
tail -12 gpurun_out/r5/pytest_run5.txt
for wo in 1 0 1 0; do
GGML_HIP_FUSE_WO=$wo timeout 300 python bench.py --weights blocks --no-cpu-baseline --prefill-steps 0 --steps 128 > gpurun_out/r5/bench_wo$wo.json 2> gpurun_out/r5/bench_wo$wo.err
tail -2 gpurun_out/r5/bench_wo$wo.err | cut -c1-300
python - <<PY
import json
d=json.loads(open('gpurun_out/r5/bench_wo$wo.json').read().strip().splitlines()[-1])
r=d['roofline']
print('fuse_wo=$wo', d['value'], d['ms_per_step_min_median_max'], {k:(v['launches'], v['us_per_launch']) for k,v in r['per_kind'].items()}, d['config']['long_context']['tokens_per_s'], d['parity_check']['passed'])
PY
done
