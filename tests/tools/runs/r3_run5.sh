#!/bin/bash
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
echo "== cols + attention tests"; timeout 600 python -m pytest tests/test_mmq_cols_gpu.py tests/test_prompt_plan_gpu.py -q -x 2>&1 | tail -12
echo "== llama tests"; timeout 600 python -m pytest tests/test_llama_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -8
echo "== decode bench (prompt feed cols vs big8)"
for c in 1 0; do GGML_HIP_MMQ_COLS=$c timeout 600 python bench.py --steps 32 --warmup 4 --weights blocks --no-cpu-baseline --prefill-steps 2 --no-parity-check > gpurun_out/r3/dec_cols$c.json 2> gpurun_out/r3/dec_cols$c.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r3/dec_cols$c.json').read().strip().split('\n')[-1])
print("cols=$c", d['value'], d['config']['prompt_feed'], d['config']['prefill']['tokens_per_s'], d['config']['prefill']['class_ms_per_step'])
PY
tail -3 gpurun_out/r3/dec_cols$c.err
done
echo "== full suite"; timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8
