#!/bin/bash
# round-5 run 8: whole suite on the current tree; sessions with more hardware queues; the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -30 > gpurun_out/r5/pytest_run8.txt
tail -8 gpurun_out/r5/pytest_run8.txt
for q in 4 8; do
GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --mode sessions --sessions 1,2,3,4,6 --weights blocks --steps 128 > gpurun_out/r5/sessions_q$q.json 2> gpurun_out/r5/sessions_q$q.err
python -c "
import json
d=json.loads(open('gpurun_out/r5/sessions_q$q.json').read().strip().splitlines()[-1])
print('hw queues $q:', [(r['sessions'], r['aggregate_tokens_per_s'], r['vs_one_session']) for r in d['runs']])
"
done
timeout 600 python bench.py > gpurun_out/r5/bench_run8.json 2> gpurun_out/r5/bench_run8.err
tail -3 gpurun_out/r5/bench_run8.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_run8.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step_min_median_max'], r['kernel_kind'], r['frac'], r['traffic_over_algo'], r['whole_token'])
print({k:(v['launches'],v['us_per_launch'],v['frac']) for k,v in r['per_kind'].items()})
print(d['config']['call_sequence']['reference_call_sequence'])
print(d['config']['prompt_feed']['steady'], d['config']['long_context'], d['config']['prefill']['tokens_per_s'], d['config']['prefill']['roofline']['frac'])
print(d['cpu_baseline']['value'], d['parity_check']['passed'], d['parity_check']['max_over_std'], d['parity_check']['bound_over_std'])
PY
