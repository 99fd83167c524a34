#!/bin/bash
# round 6, run 7: ring depth sweep with counted waits (variants built with -DBIG_PF1 -DBIG_PF2); the NaN test first
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6; T=run9
timeout 900 python -m pytest tests/test_fused_attn_gpu.py tests/test_llama_gpu.py tests/test_layer_chain_gpu.py tests/test_ref_branch_gpu.py tests/test_ops_gpu.py -q -x 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -5 > $O/${T}_tests.txt
cat $O/${T}_tests.txt
for V in main pf43 pf42 pf32 pf33 pf63 pf64 pf22 main; do
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
