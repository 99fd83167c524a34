#!/bin/bash
# round 6: k_p_attn's query tiles dealt so that a CU's two workgroups add up to the same number of keys (r1 = 8 at 7B / 512 tokens) against longest-first throughout (r1 = 16)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
for R1 in 16 0 16 0 4 12; do
  E=""; [ $R1 != 0 ] && E="GGML_HIP_PATTN_R1=$R1"
  env $E timeout 300 python bench.py --mode prefill --no-cpu-baseline --no-parity-check > gpurun_out/r6/pattn_r1_$R1.json 2> gpurun_out/r6/pattn_r1_$R1.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r6/pattn_r1_$R1.json').read().strip().splitlines()[-1])
    print('r1=$R1 (0 = rule)', d.get('value'), d.get('ms_per_step'), json.dumps(d.get('class_ms_per_step')))
except Exception as e: print('r1=$R1 failed', e)
PY
done
timeout 200 python tests/tools/pattn_timeline.py 512 0 2>&1 | grep -v "^ROCm\|^Host\|^Librccl\|^HIP\|^RCCL" | head -24
