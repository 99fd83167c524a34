#!/bin/bash
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
echo "== fused attention tests"; timeout 600 python -m pytest tests/test_prompt_plan_gpu.py -q -x -k "fused_prompt_attention or bit_identical_to_the_node" 2>&1 | tail -15
echo "== c3 decode"; timeout 600 python -m pytest tests/test_c3_gpu.py -q -s -k "decode" 2>&1 | grep -E "7B|layer-0|passed|failed|Error|assert" | head
echo "== prefill bench fused / unfused"
for f in 1 0; do GGML_HIP_ATTN_FUSED=$f timeout 600 python bench.py --mode prefill --steps 5 --warmup 2 --weights blocks > gpurun_out/r3/prefill_f$f.json 2> gpurun_out/r3/prefill_f$f.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r3/prefill_f$f.json').read().strip().split('\n')[-1])
print("fused=$f", d['value'], d['ms_per_step'], d['roofline']['class_ms_per_step'], d['roofline']['achieved'])
PY
done
echo "== full suite"; timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8
