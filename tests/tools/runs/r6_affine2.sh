#!/bin/bash
# round 6: warm-up size again, now with the XCD-affine rows (the heads publish ~0.8 us earlier: into the warm-up's backlog)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6
for MB in 24 8 12 16 20 24 28 16; do
  GGML_HIP_WARM_MB=$MB timeout 300 python bench.py --steps 128 --no-cpu-baseline --prefill-steps 0 --headline-only --weights blocks --no-per-layer-check > $O/affw_$MB.json 2> $O/affw_$MB.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/affw_$MB.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('affine warm $MB', d['value'], 'all_matvecs_ms', r['all_matvecs_per_token']['ms'], {k:(v['us_per_launch'], v.get('in_sequence_us_per_launch')) for k,v in r['per_kind'].items()}, d['parity_check'].get('passed'))
except Exception as e: print('$MB failed', e)
PY
done
