#!/bin/bash
# round-4 run 30: whole suite + default bench line after the attention-variant refactor (2 / 3 / 4 attention workgroups per head
# inside the wq|wk|wv launch beyond 576 positions)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests -q -m gpu > gpurun_out/r4/suite30.txt 2>&1; grep -E "passed|failed|Error|error" gpurun_out/r4/suite30.txt | head -8 | cut -c1-300
timeout 600 python bench.py > gpurun_out/r4/bench30.json 2> gpurun_out/r4/bench30.err; tail -2 gpurun_out/r4/bench30.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench30.json').read().strip().splitlines()[-1]); c=d['config']
print(d['value'], d['ms_per_step_min_median_max'], d['roofline']['frac'], d['parity_check']['passed'])
print(c['prefill']['tokens_per_s'], c['prompt_feed']['steady']['tokens_per_s'], c.get('long_context'), d['cpu_baseline']['value'])
PY
