#!/bin/bash
# round-3 run 24: K tokens per graph launch in the greedy chain
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3
timeout 600 python -X faulthandler -m pytest tests/test_llama_gpu.py -q -m gpu -x -k "chain" 2>&1 | tail -4 | cut -c1-300
timeout 400 python bench.py --weights blocks --steps 128 --no-cpu-baseline --prefill-steps 0 --no-parity-check > gpurun_out/r3/bench24.json 2>gpurun_out/r3/bench24.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench24.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['device_sampling'])
PY
tail -3 gpurun_out/r3/bench24.err
