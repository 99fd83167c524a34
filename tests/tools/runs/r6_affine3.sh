#!/bin/bash
# round 6: the XCD-affine rows for 2-4 attention workgroups per head too (long contexts), tests + A/B at 1800 positions
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6
timeout 900 python -m pytest tests/test_fused_attn_gpu.py tests/test_fused_timeout_gpu.py tests/test_ref_branch_gpu.py -x -q 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -4
for A in 0 1 0 1; do
  GGML_HIP_AFFINE=$A timeout 300 python bench.py --steps 64 --no-cpu-baseline --prefill-steps 0 --weights blocks --no-per-layer-check > $O/aff3_$A.json 2> $O/aff3_$A.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/aff3_$A.json').read().strip().splitlines()[-1]); c=d['config']
    print('affine $A', d['value'], 'long', c['long_context'], d['parity_check'].get('passed'))
except Exception as e: print('$A failed', e)
PY
done
