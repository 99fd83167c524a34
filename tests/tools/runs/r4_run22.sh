#!/bin/bash
# round-4 run 22: K plan for prompt chunks of up to 8 tokens: tests, feed rate with the plan on / off, decode line again
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_kquant_plan_gpu.py tests/test_kquant_gpu.py tests/test_selftest_gpu.py -q -m gpu -s > gpurun_out/r4/run22_pytest.txt 2>&1
grep -E "passed|failed|Error|error|worst|assert" gpurun_out/r4/run22_pytest.txt | tail -30 | cut -c1-250
for pk in 1 0; do
GGML_HIP_PLAN_K=$pk timeout 400 python bench.py --mode feed --wtype q4_k --steps 5 > gpurun_out/r4/feed22_q4_k_$pk.json 2> gpurun_out/r4/feed22_q4_k_$pk.err; tail -2 gpurun_out/r4/feed22_q4_k_$pk.err | cut -c1-300; cat gpurun_out/r4/feed22_q4_k_$pk.json | cut -c1-400
done
timeout 400 python bench.py --wtype q4_k --no-cpu-baseline --prefill-steps 0 --steps 64 --warmup 4 > gpurun_out/r4/bench22_q4_k.json 2> gpurun_out/r4/bench22_q4_k.err; tail -3 gpurun_out/r4/bench22_q4_k.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench22_q4_k.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print(d.get('value'), r.get('frac'), d['config']['prompt_feed'], d['config'].get('long_context'))
PY
