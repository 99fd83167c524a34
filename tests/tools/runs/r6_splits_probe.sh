#!/bin/bash
# round 6: would a 3/4/8-way K split of wo / w2 (partial tiles, on 256-tiles or on 128-tiles) beat the 2-way split on 128-tiles?
# (timing-only probe in plan_launch_prompt's gemm(): the consumers still add two partials)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r6
for T in 1 2; do for SP in 0 3 4 8; do
  GGML_HIP_MMQ_T256=$T GGML_HIP_SPLITS_PROBE=$SP timeout 300 python bench.py --mode prefill --no-cpu-baseline --no-parity-check > gpurun_out/r6/sp_${T}_$SP.json 2> gpurun_out/r6/sp_${T}_$SP.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r6/sp_${T}_$SP.json').read().strip().splitlines()[-1])
    pf=d['config'].get('prefill') or d
    print('t256=$T splits=$SP', d.get('value'), d.get('ms_per_step'), json.dumps(d.get('roofline',{}).get('kernel_launch_counts')), json.dumps(d.get('class_ms_per_step') or d['config'].get('class_ms_per_step')))
except Exception as e: print('t256=$T splits=$SP failed', e)
PY
done; done
