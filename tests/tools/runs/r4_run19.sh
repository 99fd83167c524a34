#!/bin/bash
# round-4 run 19: the K plan (K-quant LLaMA decode as 13 launches per layer from a hipGraph): its tests, the K-quant suite,
# bench lines for Q4_K / Q6_K with the plan on and off
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_kquant_plan_gpu.py tests/test_kquant_gpu.py -q -x -m gpu -s > gpurun_out/r4/run19_pytest.txt 2>&1
grep -E "passed|failed|Error|error|worst|assert" gpurun_out/r4/run19_pytest.txt | head -40 | cut -c1-250
for cfg in "q4_k 1" "q4_k 2" "q4_k 0" "q6_k 1"; do set -- $cfg
GGML_HIP_PLAN_K=$2 timeout 400 python bench.py --wtype $1 --no-cpu-baseline --prefill-steps 0 --steps 64 --warmup 4 > gpurun_out/r4/bench19_$1_$2.json 2> gpurun_out/r4/bench19_$1_$2.err; tail -3 gpurun_out/r4/bench19_$1_$2.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r4/bench19_$1_$2.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print('$1 plan_k=$2', d.get('value'), d.get('ms_per_step'), r.get('frac'), {k: (v['launches'], v['us_per_launch'], v['GBps']) for k, v in (r.get('per_kind') or {}).items()}, r.get('class_ms_per_token'), r.get('class_launches_per_token'), (d.get('parity_check') or {}).get('max_over_std'), (d.get('parity_check') or {}).get('passed'))
except Exception as e: print('$1 plan_k=$2 failed', e)
PY
done
