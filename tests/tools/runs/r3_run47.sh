#!/bin/bash
# round-3 run 47: final green-state check — smoke(), whole GPU suite, default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | cut -c1-200
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu > gpurun_out/r3/suite47.txt 2>&1; grep -v "^  File" gpurun_out/r3/suite47.txt | tail -4 | cut -c1-300
timeout 900 python bench.py > gpurun_out/r3/bench47.json 2> gpurun_out/r3/bench47.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench47.json').read().strip().splitlines()[-1]); c=d['config']
print(d['value'], d['ms_per_step_min_median_max'], d['roofline']['frac'], d['parity_check']['passed'], c['host_split_per_token']['host_phases_us'])
print(c['prefill']['tokens_per_s'], c['prompt_feed']['steady']['tokens_per_s'])
PY
tail -2 gpurun_out/r3/bench47.err
