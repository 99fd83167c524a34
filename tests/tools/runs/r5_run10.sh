#!/bin/bash
# round-5 run 10: the big-workgroup K mat-vec (k_mmvq_kbig): K plan tests, A/B of the Q4_K / Q6_K decode lines
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kquant_plan_gpu.py tests/test_kquant_gpu.py -m gpu -q -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -12
for kb in 1 0; do for wt in q4_k q6_k; do
GGML_HIP_KBIG=$kb timeout 300 python bench.py --wtype $wt --no-cpu-baseline --prefill-steps 0 --steps 64 > gpurun_out/r5/bench_${wt}_kbig$kb.json 2> gpurun_out/r5/bench_${wt}_kbig$kb.err
tail -2 gpurun_out/r5/bench_${wt}_kbig$kb.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r5/bench_${wt}_kbig$kb.json').read().strip().splitlines()[-1])
    r=d['roofline']
    print('$wt kbig=$kb', d['value'], d['ms_per_step_min_median_max'], {k:(v['launches'], v['us_per_launch'], v['frac']) for k,v in r['per_kind'].items()}, r['class_ms_per_token'], r['class_launches_per_token'], d['parity_check']['passed'], d['config']['long_context'])
except Exception as e: print('$wt kbig=$kb failed', e)
PY
done; done
