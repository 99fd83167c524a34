#!/bin/bash
# round 6: feed_prompt's chunks only enqueued (all but the last): tests + A/B of the prompt feed / prefill legs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6
timeout 1500 python -m pytest tests/test_llama_gpu.py tests/test_prompt_plan_gpu.py tests/test_kquant_plan_gpu.py tests/test_c3_gpu.py tests/test_concurrent_gpu.py tests/test_split_gpu.py tests/test_mmq_cols_gpu.py -x -q 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -5
for A in 0 1 0 1; do
  LLM_HOST_PIPELINE_CHUNKS=$A timeout 300 python bench.py --mode feed --weights blocks --steps 5 > $O/pipe_feed_$A.json 2> $O/pipe_feed_$A.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/pipe_feed_$A.json').read().strip().splitlines()[-1])
    print('pipeline $A feed', d['value'], d.get('ms_per_step'), str(d['config'])[:300])
except Exception as e: print('$A failed', e)
PY
done
for A in 0 1; do
  LLM_HOST_PIPELINE_CHUNKS=$A timeout 400 python bench.py --steps 32 --no-cpu-baseline --weights blocks --no-per-layer-check > $O/pipe_all_$A.json 2> $O/pipe_all_$A.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/pipe_all_$A.json').read().strip().splitlines()[-1]); c=d['config']
    print('pipeline $A', d['value'], 'prefill', c['prefill']['tokens_per_s'], c['prefill']['ms_per_step'], 'feed', c['prompt_feed']['tokens_per_s'], c['prompt_feed']['steady'])
except Exception as e: print('$A failed', e)
PY
done
