#!/bin/bash
# round 6, run 4: the next launch's norm weights warmed into L2 (BigArgs::nwarm): staged times with / without, replay totals
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6; T=${1:-run4}
for WN in 1 0; do
echo "== warm_norm $WN"
GGML_HIP_WARM_NORM=$WN GGML_HIP_FUSE_ATTN=0 timeout 200 python tests/tools/timeline.py 7b 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -10 > $O/${T}_timeline4_wn$WN.txt
cat $O/${T}_timeline4_wn$WN.txt
GGML_HIP_WARM_NORM=$WN GGML_HIP_FUSE_ATTN=0 timeout 200 python tests/tools/timeline.py 7b 256 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | head -40 > $O/${T}_timeline_plain_wn$WN.txt
grep 'staged\|exit' $O/${T}_timeline_plain_wn$WN.txt
done
for cfg in "1 0" "1 3072" "0 0" "0 3072" "1 0" "1 3072" "1 4096" "1 2048"; do set -- $cfg
  GGML_HIP_WARM_NORM=$1 GGML_HIP_WARM_ROWS=$2 timeout 300 python bench.py --steps 128 --no-cpu-baseline --prefill-steps 0 --headline-only --weights blocks > $O/${T}_wn$1_wr$2.json 2> $O/${T}_wn$1_wr$2.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/${T}_wn$1_wr$2.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('warm_norm $1 warm_rows $2', d['value'], 'all_matvecs_ms', r['all_matvecs_per_token']['ms'], {k:v['us_per_launch'] for k,v in r['per_kind'].items()}, d['parity_check'].get('passed'))
except Exception as e: print('$1 $2 failed', e)
PY
done
