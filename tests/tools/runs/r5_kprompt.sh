#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5g
timeout 900 python -m pytest tests/test_kquant_plan_gpu.py -m gpu -q -x -k "prompt_plan" -s 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -25
for wt in q4_k q6_k; do
timeout 500 python bench.py --wtype $wt --mode prefill --no-cpu-baseline > gpurun_out/r5g/r05_prefill_$wt.json 2> gpurun_out/r5g/r05_prefill_$wt.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r5g/r05_prefill_$wt.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('$wt prefill', d['value'], d['ms_per_step'], r['frac'], r['kernel_launch_counts'], r['class_ms_per_step'], r['class_launches_per_step'])
except Exception as e: print('$wt failed', e); print(open('gpurun_out/r5g/r05_prefill_$wt.err').read()[-800:])
PY
done
timeout 300 python bench.py --wtype q4_k --mode feed --n-batch 48 --steps 3 2>/dev/null | tail -1
