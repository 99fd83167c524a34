#!/bin/bash
# round 6: wo_tail's waves without a granule in the second poll batch skip it (they polled granule 0)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fused_attn_gpu.py tests/test_llama_gpu.py -x -q 2>&1 | grep -v "^ROCm\|^Host\|^Librccl\|^HIP\|^RCCL" | tail -3
for i in 1 2 3; do
  timeout 300 python bench.py --headline-only --no-cpu-baseline --no-parity-check --prefill-steps 0 > gpurun_out/r6/pollskip_$i.json 2> gpurun_out/r6/pollskip_$i.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r6/pollskip_$i.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('run $i', d['value'], d['ms_per_step'], 'qkv', r['per_kind']['qkv']['us_per_launch'], r['per_kind']['qkv'].get('in_sequence_us_per_launch'), 'all', r['all_matvecs_per_token']['ms'])
except Exception as e: print('run $i failed', e)
PY
done
timeout 200 python tests/tools/wo_timeline.py 128 2>&1 | grep -v "^ROCm\|^Host\|^Librccl\|^HIP\|^RCCL" | tail -22
