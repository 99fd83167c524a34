#!/bin/bash
# round 6, run 6: the register ring with counted waits (unconditional refills, scales as aligned words, no maybe-pending store):
# correctness subset, plain-launch timeline, replay totals with / without the L2 warm-up
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6; T=${1:-run6}
timeout 900 python -m pytest tests/test_fused_attn_gpu.py tests/test_llama_gpu.py tests/test_layer_chain_gpu.py tests/test_ref_branch_gpu.py tests/test_ops_gpu.py -q -x 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -5 > $O/${T}_tests.txt
cat $O/${T}_tests.txt
GGML_HIP_FUSE_ATTN=0 timeout 200 python tests/tools/timeline.py 7b 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -10 > $O/${T}_timeline4.txt
cat $O/${T}_timeline4.txt
GGML_HIP_FUSE_ATTN=0 timeout 200 python tests/tools/timeline.py 7b 256 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | head -40 > $O/${T}_timeline_plain.txt
grep 'staged\|exit' $O/${T}_timeline_plain.txt
for cfg in "0" "3072" "0" "3072" "4096" "2048"; do set -- $cfg
  GGML_HIP_WARM_ROWS=$1 timeout 300 python bench.py --steps 128 --no-cpu-baseline --prefill-steps 0 --headline-only --weights blocks > $O/${T}_wr$1.json 2> $O/${T}_wr$1.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/${T}_wr$1.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('warm_rows $1', d['value'], 'all_matvecs_ms', r['all_matvecs_per_token']['ms'], {k:v['us_per_launch'] for k,v in r['per_kind'].items()}, d['parity_check'].get('passed'))
except Exception as e: print('$1 failed', e)
PY
done
