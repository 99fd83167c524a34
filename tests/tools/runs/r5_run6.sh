#!/bin/bash
# round-5 run 6: timeline of the WO form; A/B decode rate with the single-lane first wait
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 200 python tests/tools/wo_timeline.py 128 > gpurun_out/r5/wo_timeline_128.txt 2>&1; cat gpurun_out/r5/wo_timeline_128.txt | tail -16
timeout 300 python -m pytest tests/test_fused_attn_gpu.py -m gpu -q -x 2>&1 | tail -3
for wo in 1 0 1; do
GGML_HIP_FUSE_WO=$wo timeout 300 python bench.py --weights blocks --no-cpu-baseline --prefill-steps 0 --steps 128 > gpurun_out/r5/bench_b_wo$wo.json 2> gpurun_out/r5/bench_b_wo$wo.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r5/bench_b_wo$wo.json').read().strip().splitlines()[-1])
r=d['roofline']
print('fuse_wo=$wo', d['value'], d['ms_per_step_min_median_max'], {k:(v['launches'], v['us_per_launch']) for k,v in r['per_kind'].items()}, d['config']['long_context']['tokens_per_s'], d['parity_check']['passed'])
PY
done
