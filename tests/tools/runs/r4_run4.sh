#!/bin/bash
# round-4 run 4: first contact of k_qkv_attn (wq|wk|wv + attention in one launch): its own tests, the per-layer teacher-forced
# test with the oracle's band, the decode tests of the suite, bench with the option on and off
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4
timeout 600 python -X faulthandler -m pytest tests/test_fused_attn_gpu.py -q -x -m gpu > gpurun_out/r4/s4_fused.txt 2>&1; tail -15 gpurun_out/r4/s4_fused.txt | cut -c1-250
timeout 600 python -X faulthandler -m pytest tests/test_ref_branch_gpu.py -q -x -s -m gpu -k every_layer 2>&1 | grep -E "^layer|oracle fwd|per-layer|passed|failed" | cut -c1-200
timeout 900 python -X faulthandler -m pytest tests/test_llama_gpu.py tests/test_c3_gpu.py tests/test_fullsize_gpu.py -q -x -m gpu > gpurun_out/r4/s4_llama.txt 2>&1; tail -5 gpurun_out/r4/s4_llama.txt | cut -c1-250
for f in 1 0; do
GGML_HIP_FUSE_ATTN=$f timeout 600 python bench.py --no-cpu-baseline --prefill-steps 0 > gpurun_out/r4/bench4_f$f.json 2> gpurun_out/r4/bench4_f$f.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r4/bench4_f$f.json').read().strip().splitlines()[-1]); c=d['config']
print('fuse_attn=$f', d['value'], d['ms_per_step_min_median_max'], d['roofline']['frac'], d['parity_check']['passed'], d['parity_check']['max_over_std'])
print({k: v['us_per_launch'] for k, v in d['roofline']['per_kind'].items()}, d['roofline'].get('all_matvecs_per_token'), c.get('long_context'))
PY
tail -2 gpurun_out/r4/bench4_f$f.err
done
