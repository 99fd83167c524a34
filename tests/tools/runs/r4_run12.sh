#!/bin/bash
# round-4 run 12: k_qkv_attn with the context-sized 512-position register window: tests, bench A/B against fuse_attn=0, timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4
timeout 600 python -X faulthandler -m pytest tests/test_fused_attn_gpu.py -q -x -m gpu 2>&1 | tail -3 | cut -c1-250
for f in 1 0 1 0; do
GGML_HIP_FUSE_ATTN=$f timeout 600 python bench.py --no-cpu-baseline --prefill-steps 0 --steps 128 > gpurun_out/r4/bench12_f$f.json 2> gpurun_out/r4/bench12_f$f.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r4/bench12_f$f.json').read().strip().splitlines()[-1]); c=d['config']
print('fuse_attn=$f', d['value'], d['ms_per_step_min_median_max'], d['parity_check']['passed'], {k: v['us_per_launch'] for k, v in d['roofline']['per_kind'].items()}, d['roofline']['all_matvecs_per_token']['frac'], d['roofline']['whole_token']['frac'])
PY
done
timeout 300 python tests/tools/fused_timeline.py 128 2>&1 | tail -14
timeout 300 python tests/tools/fused_timeline.py 400 2>&1 | tail -14
