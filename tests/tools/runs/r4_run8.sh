#!/bin/bash
# round-4 run 8: k_qkv_attn's mat-vec workgroups warm the L2 for wo / w1|w3 while the attention workgroups finish: size sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4
timeout 600 python -X faulthandler -m pytest tests/test_fused_attn_gpu.py -q -x -m gpu 2>&1 | tail -3 | cut -c1-250
for d in -1 0 4 8 12 16; do
GGML_HIP_FUSE_PREFETCH=$d timeout 600 python bench.py --no-cpu-baseline --prefill-steps 0 --steps 96 > gpurun_out/r4/bench8_p$d.json 2> gpurun_out/r4/bench8_p$d.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r4/bench8_p$d.json').read().strip().splitlines()[-1]); c=d['config']
print('fuse_prefetch=$d', d['value'], d['ms_per_step_min_median_max'], d['parity_check']['passed'], c['long_context']['tokens_per_s'])
PY
done
GGML_HIP_FUSE_PREFETCH=8 timeout 300 python tests/tools/fused_timeline.py 128 2>&1 | tail -14
GGML_HIP_FUSE_PREFETCH=8 timeout 250 python tests/tools/timeline.py 7b 2>&1 | grep -E "^avg|^gap|^token|^attention"
