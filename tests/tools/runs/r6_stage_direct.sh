#!/bin/bash
# round 6: a stage's first layer reads stage_in and its last w2 launch writes stage_out (no copy nodes at the stage's ends): tests + split bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_split_gpu.py tests/test_pipeline_2proc_gpu.py tests/test_ref_branch_gpu.py tests/test_prompt_plan_gpu.py::test_prompt_plan_stage_of_a_layer_split tests/test_fused_timeout_gpu.py tests/test_llama_gpu.py -x -q 2>&1 | grep -v "^ROCm\|^Host\|^Librccl\|^HIP\|^RCCL" | tail -4
for G in 2 4 8; do
  timeout 300 python bench.py --mode split --split $G --weights blocks --steps 128 > gpurun_out/r6/direct_split$G.json 2> gpurun_out/r6/direct_split$G.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r6/direct_split$G.json').read().strip().splitlines()[-1])
    print('split $G', d['value'], 'unsplit', d['unsplit']['tokens_per_s'], 'hop us', d['overhead_per_hop_us'], 'same ids', d['same_greedy_ids'])
except Exception as e: print('split $G failed', e); print(open('gpurun_out/r6/direct_split$G.err').read()[-600:])
PY
done
timeout 300 python bench.py --no-cpu-baseline --prefill-steps 0 --headline-only 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('decode', d['value'], 'per-layer parity', d['parity_check']['per_layer']['passed'], d['parity_check']['per_layer']['worst_max'])"
