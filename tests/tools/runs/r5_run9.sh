#!/bin/bash
# round-5 run 9: k_argmax_next with its loads in flight; speculation and chain tests; the bench line's call-sequence legs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_speculate_gpu.py tests/test_llama_gpu.py tests/test_fused_attn_gpu.py tests/test_device_tools_gpu.py -m gpu -q -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -6
timeout 300 python bench.py --weights blocks --no-cpu-baseline --prefill-steps 0 --steps 128 > gpurun_out/r5/bench_run9.json 2> gpurun_out/r5/bench_run9.err
tail -3 gpurun_out/r5/bench_run9.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5/bench_run9.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step_min_median_max'])
print(json.dumps(d['config']['call_sequence']['reference_call_sequence'])[:900])
print(d['config']['device_sampling'])
print(d['roofline']['kernel'][:200])
PY
