#!/bin/bash
# round 6 (NOT KEPT): a speculating token's result copies on a copy stream + the sampled id by a pinned store: tests + the bench's speculation legs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_speculate_gpu.py tests/test_llama_gpu.py tests/test_device_tools_gpu.py tests/test_fused_timeout_gpu.py -x -q 2>&1 | grep -v "^ROCm\|^Host\|^Librccl\|^HIP\|^RCCL" | tail -4
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-parity-check --prefill-steps 0 > gpurun_out/r6/specs_$i.json 2> gpurun_out/r6/specs_$i.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r6/specs_$i.json').read().strip().splitlines()[-1])
    c=d['config']; rs=c['call_sequence']['reference_call_sequence']; sp=rs['with_backend_speculation']
    print('run $i value', d['value'], 'ref', rs['tokens_per_s'], 'spec', sp['tokens_per_s'], 'hits', sp['hits'], sp['of'], 'spec+begin/end', sp['and_begin_end_sequence_tokens_per_s'], 'dev', c['device_sampling']['tokens_per_s'])
except Exception as e: print('run $i failed', e)
PY
done
