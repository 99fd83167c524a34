#!/bin/bash
# round-5 run 11: k_mmvq_kbig dealing over 16 waves vs the even dealing; K plan tests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kquant_plan_gpu.py -m gpu -q -x 2>&1 | tail -3
for kw in 16 0 16 0; do
GGML_HIP_KBIG_WAVES=$kw timeout 300 python bench.py --wtype q4_k --no-cpu-baseline --prefill-steps 0 --steps 64 --no-parity-check > gpurun_out/r5/bench_q4_k_kw$kw.json 2> gpurun_out/r5/bench_q4_k_kw$kw.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r5/bench_q4_k_kw$kw.json').read().strip().splitlines()[-1])
r=d['roofline']
print('kbig waves $kw', d['value'], d['ms_per_step_min_median_max'], {k:(v['us_per_launch']) for k,v in r['per_kind'].items()})
PY
done
