#!/bin/bash
# round 6, run 2: NORM staging on all 16 waves + the L2 warm-up: correctness subset, staged times, all-mat-vec replay sweep
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6
GGML_HIP_WARM_ROWS=3072 timeout 900 python -m pytest tests/test_fused_attn_gpu.py tests/test_llama_gpu.py tests/test_layer_chain_gpu.py tests/test_ref_branch_gpu.py -q -x 2>&1 | tail -3 > $O/run2_tests.txt
cat $O/run2_tests.txt
GGML_HIP_FUSE_ATTN=0 timeout 200 python tests/tools/timeline.py 7b 256 2>&1 | grep -v '^ROCm\|^Host' | head -40 > $O/run2_timeline_plain.txt
grep 'staged\|exit' $O/run2_timeline_plain.txt
for R in 0 2048 3072 4096 0 3072; do
  GGML_HIP_WARM_ROWS=$R timeout 300 python bench.py --steps 128 --no-cpu-baseline --prefill-steps 0 --headline-only --weights blocks > $O/run2_warm_$R.json 2> $O/run2_warm_$R.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/run2_warm_$R.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('warm_rows $R', d['value'], 'all_matvecs_ms', r['all_matvecs_per_token']['ms'], {k:v['us_per_launch'] for k,v in r['per_kind'].items()}, d['parity_check'].get('passed'))
except Exception as e: print('warm $R failed', e)
PY
done
for R in 0 3072; do
GGML_HIP_WARM_ROWS=$R timeout 200 python tests/tools/wo_timeline.py 128 > $O/run2_wo_timeline_warm$R.txt 2>&1
echo "== timeline warm_rows $R"; cat $O/run2_wo_timeline_warm$R.txt
done
