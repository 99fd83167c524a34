#!/bin/bash
# round-5 run 7: speculative next token: tests + the three call sequences in one bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_speculate_gpu.py tests/test_fused_attn_gpu.py tests/test_llama_gpu.py -m gpu -q -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -25 > gpurun_out/r5/pytest_run7.txt
tail -12 gpurun_out/r5/pytest_run7.txt
timeout 300 python bench.py --weights blocks --no-cpu-baseline --prefill-steps 0 --steps 128 > gpurun_out/r5/bench_run7.json 2> gpurun_out/r5/bench_run7.err
tail -3 gpurun_out/r5/bench_run7.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_run7.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step_min_median_max'])
print(json.dumps(d['config']['call_sequence']['reference_call_sequence'], indent=1))
PY
