#!/bin/bash
# round 6: XCD-affine dealing of wq|wk|wv in the fused launch (rows through the XCD's own L2), tests + timeline + A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6
timeout 900 python -m pytest tests/test_fused_attn_gpu.py tests/test_fused_timeout_gpu.py -x -q 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -5
for A in 1 0; do echo "== affine $A"; GGML_HIP_AFFINE=$A timeout 300 python tests/tools/wo_timeline.py 128 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tee $O/wo_timeline_affine$A.txt; done
for A in 0 1 0 1; do
  GGML_HIP_AFFINE=$A timeout 300 python bench.py --steps 128 --no-cpu-baseline --prefill-steps 0 --headline-only --weights blocks > $O/affine_$A.json 2> $O/affine_$A.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/affine_$A.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('affine $A', d['value'], 'all_matvecs_ms', r['all_matvecs_per_token']['ms'], {k:(v['us_per_launch'], v.get('in_sequence_us_per_launch')) for k,v in r['per_kind'].items()}, d['parity_check'].get('passed'), d['parity_check'].get('per_layer',{}).get('worst_max'))
except Exception as e: print('$A failed', e)
PY
done
