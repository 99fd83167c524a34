#!/bin/bash
# round 6, run 18: the warm-up issued right behind the workgroup's last wq|wk|wv pair (before the wo rows are unpacked)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6; T=run18
for R in 0 1024 2048 3072 4096 0 2048 3072; do
  GGML_HIP_WARM_ROWS=$R timeout 300 python bench.py --steps 128 --no-cpu-baseline --prefill-steps 0 --headline-only --weights blocks --no-per-layer-check > $O/${T}_$R.json 2> $O/${T}_$R.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/${T}_$R.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('warm $R', d['value'], 'all_matvecs_ms', r['all_matvecs_per_token']['ms'], {k:v['us_per_launch'] for k,v in r['per_kind'].items()}, d['parity_check'].get('passed'))
except Exception as e: print('$R failed', e)
PY
done
for R in 0 3072; do echo "== warm $R"; GGML_HIP_WARM_ROWS=$R timeout 200 python tests/tools/wo_timeline.py 128 2>&1 | grep -v '^ROCm\|^Host\|^Librccl'; done
