#!/bin/bash
# round-5 run 4: whole GPU suite (all failures listed), the default bench line with the new fields
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -40 > gpurun_out/r5/pytest_run4.txt
tail -15 gpurun_out/r5/pytest_run4.txt
timeout 600 python bench.py > gpurun_out/r5/bench_run4.json 2> gpurun_out/r5/bench_run4.err
tail -3 gpurun_out/r5/bench_run4.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_run4.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step_min_median_max'], r['kernel_kind'], r['frac'], r['traffic_over_algo'])
print({k:(v['us_per_launch'],v['frac']) for k,v in r['per_kind'].items()})
print(d['config']['call_sequence']['reference_call_sequence'])
print(d['config']['host_split_per_token']['host_phases_us'])
print(d['config']['prompt_feed']['steady'], d['config']['long_context'], d['config']['prefill']['tokens_per_s'], d['config']['prefill']['roofline']['frac'])
print(d['cpu_baseline']['value'], d['parity_check']['passed'], d['parity_check']['max_over_std'], d['parity_check']['bound_over_std'])
PY
