#!/bin/bash
# round 6: the K plan's fused launch warms wo + the first rows of w1|w3 for the launches behind it (warm_mb), A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6; T=kwarm
for wt in q4_k q6_k; do for MB in 0 16 24 28 0 24; do
  GGML_HIP_WARM_MB=$MB timeout 300 python bench.py --wtype $wt --steps 96 --no-cpu-baseline --prefill-steps 0 --headline-only > $O/${T}_${wt}_$MB.json 2> $O/${T}_${wt}_$MB.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/${T}_${wt}_$MB.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('$wt warm $MB', d['value'], 'all_matvecs_ms', r['all_matvecs_per_token']['ms'], {k:v['us_per_launch'] for k,v in r['per_kind'].items()}, d['parity_check'].get('passed'))
except Exception as e: print('$wt $MB failed', e)
PY
done; done
timeout 900 python -m pytest tests/test_kquant_plan_gpu.py -x -q 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -5
timeout 600 python -m pytest tests/test_fused_attn_gpu.py -x -q -k "wo_tail or 7b_width" 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -5
