#!/bin/bash
# round 6, run 15: same-box A/B: base (round 5) | ring (counted waits) | main (+ sum of squares before the ring requests)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6; T=${1:-run15}; VARS=${2:-"base ring main ring main"}
for V in $VARS; do
  if [ $V = main ]; then unset GGML_HIP_LIB; else export GGML_HIP_LIB=$GRAFT_REPO_ROOT/llm_amd/variants/libggml_hip_$V.so; fi
  timeout 300 python bench.py --steps 128 --no-cpu-baseline --prefill-steps 0 --headline-only --weights blocks > $O/${T}_$V.json 2> $O/${T}_$V.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/${T}_$V.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('$V', d['value'], 'all_matvecs_ms', r['all_matvecs_per_token']['ms'], {k:v['us_per_launch'] for k,v in r['per_kind'].items()}, d['parity_check'].get('passed'))
except Exception as e: print('$V failed', e)
PY
done
unset GGML_HIP_LIB
timeout 200 python tests/tools/wo_timeline.py 128 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' > $O/${T}_wo_timeline.txt; cat $O/${T}_wo_timeline.txt
timeout 200 python tests/tools/timeline.py 7b 256 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | head -40 > $O/${T}_timeline_fused.txt
grep 'staged\|exit' $O/${T}_timeline_fused.txt | grep -v '0.00   0.00'
GGML_HIP_FUSE_ATTN=0 timeout 200 python tests/tools/timeline.py 7b 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -10 | head -5
