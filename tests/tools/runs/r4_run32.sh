#!/bin/bash
# round-4 run 32: attn_consumer_split with speculative exps + early sum exchange: tests, sweep, bench long-context leg
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_fused_attn_gpu.py tests/test_llama_gpu.py -q -m gpu -x -s > gpurun_out/r4/run32_pytest.txt 2>&1
grep -E "passed|failed|Error|error|attention workgroups|vs oracle|assert" gpurun_out/r4/run32_pytest.txt | tail -8 | cut -c1-250
for fh in 1 1; do
GGML_HIP_FUSE_HEADS=$fh CTX_SWEEP_NBATCH=8 timeout 600 python tests/tools/ctx_sweep.py 1 > gpurun_out/r4/r04_ctx_sweep_fh${fh}_v3.txt 2>&1; echo "fuse_heads=$fh: $(tail -10 gpurun_out/r4/r04_ctx_sweep_fh${fh}_v3.txt | tr '\n' ' ')"
done
timeout 400 python bench.py --no-cpu-baseline --prefill-steps 0 > gpurun_out/r4/bench32.json 2> gpurun_out/r4/bench32.err; tail -2 gpurun_out/r4/bench32.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench32.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step_min_median_max'], d['parity_check']['passed'], d['config']['long_context'])
PY
