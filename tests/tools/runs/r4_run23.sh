#!/bin/bash
# round-4 run 23 (second version of k_attn_split_one: two tiny exchanges instead of a gather of the score row):
# long-context tests, the K plan at long context; long-context decode rate A/B (attn_one 1 / 0), Q4_0 and Q4_K
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_fused_attn_gpu.py tests/test_kquant_plan_gpu.py tests/test_kquant_gpu.py tests/test_llama_gpu.py -q -m gpu -x -s > gpurun_out/r4/run23_pytest.txt 2>&1
grep -E "passed|failed|Error|error|long context|assert" gpurun_out/r4/run23_pytest.txt | tail -12 | cut -c1-250
for one in 1 0 1 0; do
GGML_HIP_ATTN_ONE=$one timeout 400 python bench.py --no-cpu-baseline --prefill-steps 0 --steps 32 > gpurun_out/r4/bench23_one$one.json 2> gpurun_out/r4/bench23_one$one.err; tail -2 gpurun_out/r4/bench23_one$one.err | cut -c1-300
python - <<PY
import json
d=json.loads(open('gpurun_out/r4/bench23_one$one.json').read().strip().splitlines()[-1])
print('attn_one=$one', d['value'], d['config']['long_context'], d['parity_check']['passed'])
PY
done
timeout 400 python bench.py --wtype q4_k --no-cpu-baseline --prefill-steps 0 --steps 32 --warmup 4 > gpurun_out/r4/bench23_q4_k.json 2> gpurun_out/r4/bench23_q4_k.err; tail -3 gpurun_out/r4/bench23_q4_k.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench23_q4_k.json').read().strip().splitlines()[-1])
print('q4_k', d.get('value'), d['config'].get('long_context'))
PY
