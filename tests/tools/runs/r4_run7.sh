#!/bin/bash
# round-4 run 7: k_qkv_attn with the single-wave softmax; K/V request delay sweep; honest per-kind replay (store_at)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4
timeout 600 python -X faulthandler -m pytest tests/test_fused_attn_gpu.py -q -x -m gpu 2>&1 | tail -3 | cut -c1-250
for d in 0 4 8 16; do
GGML_HIP_FUSE_KV_DELAY=$d timeout 600 python bench.py --no-cpu-baseline --prefill-steps 0 --steps 96 > gpurun_out/r4/bench7_d$d.json 2> gpurun_out/r4/bench7_d$d.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r4/bench7_d$d.json').read().strip().splitlines()[-1]); c=d['config']
print('kv_delay=$d', d['value'], d['ms_per_step_min_median_max'], d['roofline']['frac'], d['parity_check']['passed'], {k: v['us_per_launch'] for k, v in d['roofline']['per_kind'].items()}, d['roofline']['all_matvecs_per_token']['ms'])
PY
done
GGML_HIP_FUSE_KV_DELAY=0 timeout 300 python tests/tools/fused_timeline.py 128 2>&1 | tail -14
