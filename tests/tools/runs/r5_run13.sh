#!/bin/bash
# round-5 run 13: k_qkv_attn_k (K-quant wq|wk|wv + attention in one launch): K plan tests, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kquant_plan_gpu.py tests/test_kquant_gpu.py -m gpu -q -x 2>&1 | tail -12
for fa in 1 0 1 0; do
GGML_HIP_FUSE_ATTN=$fa timeout 300 python bench.py --wtype q4_k --no-cpu-baseline --prefill-steps 0 --steps 64 > gpurun_out/r5/bench_q4_k_fa$fa.json 2> gpurun_out/r5/bench_q4_k_fa$fa.err
tail -n 2 gpurun_out/r5/bench_q4_k_fa$fa.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r5/bench_q4_k_fa$fa.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('q4_k fuse_attn=$fa', d['value'], d['ms_per_step_min_median_max'], {k:(v['us_per_launch']) for k,v in r['per_kind'].items()}, r['class_ms_per_token'], r['class_launches_per_token'], d['parity_check']['passed'], d['config']['long_context']['tokens_per_s'])
except Exception as e: print('failed', e)
PY
done
GGML_HIP_FUSE_ATTN=1 timeout 300 python bench.py --wtype q6_k --no-cpu-baseline --prefill-steps 0 --steps 64 > gpurun_out/r5/bench_q6_k_fa1.json 2> /dev/null; python -c "
import json
d=json.loads(open('gpurun_out/r5/bench_q6_k_fa1.json').read().strip().splitlines()[-1]); print('q6_k', d['value'], d['parity_check']['passed'])"
