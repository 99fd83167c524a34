#!/bin/bash
# round 6: what a hit in the 256 MB Infinity Cache is worth to the decode mat-vec launches: the roofline replays with every layer on layer 0's weights
# (GGML_HIP_BENCH_SAME_LAYER=1: a 25-50 MB matrix is re-read by the next replayed launch from the memory-side cache, L2 cannot hold it)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
for SL in 0 1 0 1; do
  GGML_HIP_BENCH_SAME_LAYER=$SL timeout 300 python bench.py --headline-only --no-cpu-baseline --no-parity-check --prefill-steps 0 > gpurun_out/r6/mall_$SL.json 2> gpurun_out/r6/mall_$SL.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r6/mall_$SL.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('same_layer=$SL', d['value'], {k:(v['us_per_launch'], v.get('in_sequence_us_per_launch')) for k,v in r['per_kind'].items()}, 'all', r['all_matvecs_per_token']['ms'])
except Exception as e: print('same_layer=$SL failed', e)
PY
done
