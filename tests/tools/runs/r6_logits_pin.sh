#!/bin/bash
# round 6 (NOT KEPT, the option is gone): lm_head stored the logits to a pinned host mirror (no copy node behind the token): A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_attn_gpu.py tests/test_llama_gpu.py tests/test_speculate_gpu.py tests/test_fused_timeout_gpu.py tests/test_device_tools_gpu.py -x -q 2>&1 | grep -v "^ROCm\|^Host\|^Librccl\|^HIP\|^RCCL" | tail -3
for PIN in 0 1 0 1; do
  GGML_HIP_LOGITS_PIN=$PIN timeout 300 python bench.py --no-cpu-baseline --no-parity-check --prefill-steps 0 > gpurun_out/r6/pin_$PIN.json 2> gpurun_out/r6/pin_$PIN.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r6/pin_$PIN.json').read().strip().splitlines()[-1])
    c=d['config']; rs=c['call_sequence']['reference_call_sequence']
    print('pin=$PIN', d['value'], d['ms_per_step'], 'ref', rs['tokens_per_s'], 'spec', rs['with_backend_speculation']['tokens_per_s'], 'wait_copy', c['host_split_per_token']['host_phases_us']['compute_end_wait_and_copy'], 'long', c['long_context']['tokens_per_s'])
except Exception as e: print('pin=$PIN failed', e)
PY
done
