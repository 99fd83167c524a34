#!/bin/bash
# round 6: the next token's graph matched while the device runs (ggml_hip_graph_prepare), tests + A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_speculate_gpu.py tests/test_concurrent_gpu.py tests/test_split_gpu.py tests/test_abi.py -x -q 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -5
for A in 0 1 0 1; do
  GGML_HIP_PREPARE=$A timeout 300 python bench.py --steps 128 --no-cpu-baseline --prefill-steps 0 --headline-only --weights blocks --no-per-layer-check > $O/prep_$A.json 2> $O/prep_$A.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/prep_$A.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('prepare $A', d['value'], d['ms_per_step'], 'all_matvecs_ms', r['all_matvecs_per_token']['ms'], d['config']['host_split_per_token'], d['parity_check'].get('passed'))
except Exception as e: print('$A failed', e)
PY
done
