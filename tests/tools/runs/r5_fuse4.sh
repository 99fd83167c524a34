#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5g
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -8
for cfg in "13b q5_1" "65b q8_0"; do set -- $cfg
timeout 600 python bench.py --model $1 --wtype $2 --weights blocks --no-cpu-baseline --prefill-steps 0 --steps 48 > gpurun_out/r5g/r05_bench_$1_$2.json 2> gpurun_out/r5g/r05_bench_$1_$2.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r5g/r05_bench_$1_$2.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('$1 $2', d['value'], r['kernel_kind'], r['frac'], {k:(v['launches'], v['us_per_launch'], v['frac']) for k,v in r['per_kind'].items()}, r['whole_token']['frac'], d['parity_check']['passed'], d['config']['long_context']['tokens_per_s'], d['config']['decode_launches']['qkv_and_attention_in_one_launch_tokens'], d['config']['decode_launches']['wo_in_the_attention_launch_tokens'], d['config']['call_sequence']['reference_call_sequence']['tokens_per_s'])
except Exception as e: print('$1 $2 failed', e)
PY
done
