#!/bin/bash
# the Q5 fifth-bit spread on v_mul_u32_u24: the whole suite, then the 13B Q5_1 line (decode, prompt feed at n_batch 8) and the 7B Q5_1 feed
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5l
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -30 > gpurun_out/r5l/r05_suite_final.txt
tail -3 gpurun_out/r5l/r05_suite_final.txt
timeout 600 python bench.py --model 13b --wtype q5_1 --weights blocks --no-cpu-baseline --prefill-steps 0 --steps 48 > gpurun_out/r5l/r05_bench_13b_q5_1.json 2> gpurun_out/r5l/r05_bench_13b_q5_1.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5l/r05_bench_13b_q5_1.json').read().strip().splitlines()[-1]); r=d['roofline']
print('13b q5_1', d['value'], {k:(v['launches'], v['us_per_launch'], v['frac']) for k,v in r['per_kind'].items()}, r['whole_token']['frac'], d['parity_check']['passed'], d['config']['long_context']['tokens_per_s'], d['config']['prompt_feed']['steady']['tokens_per_s'], d['config']['call_sequence']['reference_call_sequence']['tokens_per_s'])
PY
timeout 300 python bench.py --wtype q5_1 --mode feed --n-batch 8 --steps 3 2>/dev/null | tail -1 | cut -c1-130
