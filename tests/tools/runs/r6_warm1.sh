#!/bin/bash
# round 6, run 1: L2 warm-up of w1|w3 from the WO form's idle window (option warm_rows): bit-identity, tok/s sweep, timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6
GGML_HIP_WARM_ROWS=4096 timeout 600 python -m pytest tests/test_fused_attn_gpu.py -q -x 2>&1 | tail -3 > $O/warm_fused_tests.txt
cat $O/warm_fused_tests.txt
for R in 0 1024 2048 3072 4096 6144 8192 11008 0; do
  GGML_HIP_WARM_ROWS=$R timeout 300 python bench.py --steps 64 --no-cpu-baseline --prefill-steps 0 --headline-only --weights blocks > $O/warm_$R.json 2> $O/warm_$R.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/warm_$R.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('warm_rows $R', d['value'], {k:v['us_per_launch'] for k,v in r['per_kind'].items()}, d['parity_check'].get('passed'))
except Exception as e: print('warm $R failed', e)
PY
done
for R in 0 4096; do
GGML_HIP_WARM_ROWS=$R timeout 200 python tests/tools/wo_timeline.py 128 > $O/wo_timeline_warm$R.txt 2>&1
echo "== timeline warm_rows $R"; cat $O/wo_timeline_warm$R.txt
done
