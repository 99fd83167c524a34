#!/bin/bash
# round-4 run 1: activation quantizer on ggml's AVX2 branch (oracle mode 3), one-float tensor_split, per-layer teacher forcing:
# new tests first (verbose), then the whole GPU suite, then the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4
timeout 900 python -X faulthandler -m pytest tests/test_ref_branch_gpu.py -q -x -s -m gpu > gpurun_out/r4/s1_ref_branch.txt 2>&1; tail -25 gpurun_out/r4/s1_ref_branch.txt | cut -c1-250
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu --deselect tests/test_ref_branch_gpu.py > gpurun_out/r4/suite1.txt 2>&1; grep -v "^  File" gpurun_out/r4/suite1.txt | tail -15 | cut -c1-300
timeout 900 python bench.py > gpurun_out/r4/bench1.json 2> gpurun_out/r4/bench1.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench1.json').read().strip().splitlines()[-1]); c=d['config']
print(d['value'], d['ms_per_step_min_median_max'], d['roofline']['frac'], d['parity_check'])
print(c['prefill']['tokens_per_s'], c['prompt_feed']['steady']['tokens_per_s'])
PY
tail -3 gpurun_out/r4/bench1.err
