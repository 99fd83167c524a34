#!/bin/bash
# round-5 run 14: k_mmvq_kbig / k_qkv_attn_k for all five K types: tests, the q5_k / q3_k / q2_k decode lines
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kquant_plan_gpu.py -m gpu -q -x 2>&1 | tail -3
for wt in q5_k q3_k q2_k; do for kb in 1 0; do
GGML_HIP_KBIG=$kb timeout 300 python bench.py --wtype $wt --no-cpu-baseline --prefill-steps 0 --steps 64 > gpurun_out/r5/bench_${wt}_kbig$kb.json 2> gpurun_out/r5/bench_${wt}_kbig$kb.err
tail -n 2 gpurun_out/r5/bench_${wt}_kbig$kb.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r5/bench_${wt}_kbig$kb.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('$wt kbig=$kb', d['value'], d['ms_per_step_min_median_max'], {k:(v['us_per_launch']) for k,v in r['per_kind'].items()}, r['class_launches_per_token'], d['parity_check']['passed'], d['config']['long_context']['tokens_per_s'])
except Exception as e: print('$wt kbig=$kb failed', e)
PY
done; done
