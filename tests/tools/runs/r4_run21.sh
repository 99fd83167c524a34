#!/bin/bash
# round-4 run 21: K plan tests, rest of the suite, default bench line with the pinned / first-touched CPU baseline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests -q -m gpu > gpurun_out/r4/suite21.txt 2>&1; grep -E "passed|failed|Error|error" gpurun_out/r4/suite21.txt | head -8 | cut -c1-300
nproc; lscpu | grep -E "Model name|Socket|NUMA node\(s\)|Thread" | head -5
timeout 600 python bench.py --prefill-steps 0 --cpu-secs 25 > gpurun_out/r4/bench21.json 2> gpurun_out/r4/bench21.err; tail -2 gpurun_out/r4/bench21.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench21.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], json.dumps(d['cpu_baseline'])[:900])
PY
