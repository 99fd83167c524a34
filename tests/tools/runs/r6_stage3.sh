#!/bin/bash
# round 6, run 3: where does the NORM staging wait? plain timeline (4 sampled workgroups: all stamps) + distribution + replay totals
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6; T=${1:-run3}
GGML_HIP_FUSE_ATTN=0 timeout 200 python tests/tools/timeline.py 7b 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -12 > $O/${T}_timeline4.txt
cat $O/${T}_timeline4.txt
GGML_HIP_FUSE_ATTN=0 timeout 200 python tests/tools/timeline.py 7b 256 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | head -40 > $O/${T}_timeline_plain.txt
grep 'staged\|exit' $O/${T}_timeline_plain.txt
for R in 0 3072 0 3072; do
  GGML_HIP_WARM_ROWS=$R timeout 300 python bench.py --steps 128 --no-cpu-baseline --prefill-steps 0 --headline-only --weights blocks > $O/${T}_warm_$R.json 2> $O/${T}_warm_$R.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/${T}_warm_$R.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('warm_rows $R', d['value'], 'all_matvecs_ms', r['all_matvecs_per_token']['ms'], {k:v['us_per_launch'] for k,v in r['per_kind'].items()}, d['parity_check'].get('passed'))
except Exception as e: print('warm $R failed', e)
PY
done
