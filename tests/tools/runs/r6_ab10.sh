#!/bin/bash
# round 6, run 10: same-box A/B: round-5 library (base) vs this tree (main), with / without the w1|w3 L2 warm-up; wo timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6; T=run10
for cfg in "base 0" "main 0" "main 3072" "base 0" "main 0" "main 3072" "main 4096" "main 2048"; do set -- $cfg
  if [ $1 = main ]; then unset GGML_HIP_LIB; else export GGML_HIP_LIB=$GRAFT_REPO_ROOT/llm_amd/variants/libggml_hip_$1.so; fi
  GGML_HIP_WARM_ROWS=$2 timeout 300 python bench.py --steps 128 --no-cpu-baseline --prefill-steps 0 --headline-only --weights blocks > $O/${T}_$1_$2.json 2> $O/${T}_$1_$2.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/${T}_$1_$2.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('$1 warm $2', d['value'], 'all_matvecs_ms', r['all_matvecs_per_token']['ms'], {k:v['us_per_launch'] for k,v in r['per_kind'].items()}, d['parity_check'].get('passed'))
except Exception as e: print('$1 $2 failed', e)
PY
done
unset GGML_HIP_LIB
for R in 0 3072; do
GGML_HIP_WARM_ROWS=$R timeout 200 python tests/tools/wo_timeline.py 128 > $O/${T}_wo_timeline_warm$R.txt 2>&1
echo "== timeline warm_rows $R"; grep -v '^ROCm\|^Host\|^Librccl' $O/${T}_wo_timeline_warm$R.txt
done
timeout 200 python tests/tools/timeline.py 7b 256 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | head -40 > $O/${T}_timeline_fused.txt
grep 'staged\|exit' $O/${T}_timeline_fused.txt
