#!/bin/bash
# round-4 run 31: attn_consumer with the cached positions' scores / maximum / exps / sum computed under the weight stream
# (speculative softmax): bit-identity tests, timeline, decode line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_fused_attn_gpu.py tests/test_llama_gpu.py -q -m gpu -x > gpurun_out/r4/run31_pytest.txt 2>&1
grep -E "passed|failed|Error|error|assert" gpurun_out/r4/run31_pytest.txt | tail -6 | cut -c1-250
timeout 300 python tests/tools/fused_timeline.py 128 > gpurun_out/r4/r04_fused_timeline_128_v2.txt 2>&1; tail -14 gpurun_out/r4/r04_fused_timeline_128_v2.txt
for i in 1 2; do
timeout 400 python bench.py --no-cpu-baseline --prefill-steps 0 > gpurun_out/r4/bench31_$i.json 2> gpurun_out/r4/bench31_$i.err; tail -2 gpurun_out/r4/bench31_$i.err | cut -c1-300
python - <<PY
import json
d=json.loads(open('gpurun_out/r4/bench31_$i.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step_min_median_max'], d['parity_check']['passed'], {k: v['us_per_launch'] for k, v in d['roofline']['per_kind'].items()}, d['config']['long_context'])
PY
done
