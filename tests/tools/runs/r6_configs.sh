#!/bin/bash
# round 6: the other configs on one GPU (13B Q5_1, 65B Q8_0, 7B Q8_0 / Q5_0), K-quant lines, sessions, split
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6
for cfg in "13b q5_1" "65b q8_0" "7b q8_0" "7b q5_0" "7b q4_1"; do set -- $cfg
timeout 900 python bench.py --model $1 --wtype $2 --weights blocks --no-cpu-baseline --prefill-steps 0 --steps 48 > $O/r06_bench_$1_$2.json 2> $O/r06_bench_$1_$2.err; tail -n 1 $O/r06_bench_$1_$2.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open('$O/r06_bench_$1_$2.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('$1 $2', d['value'], r['kernel_kind'], r['frac'], {k:(v['launches'], v['us_per_launch'], v['frac']) for k,v in r['per_kind'].items()}, 'whole', r['whole_token']['frac'], 'parity', d['parity_check']['passed'], d['parity_check'].get('per_layer',{}).get('worst_max'), d['parity_check'].get('per_layer',{}).get('bound_max'), 'long', d['config']['long_context'], 'ref-seq', d['config']['call_sequence']['reference_call_sequence']['tokens_per_s'])
except Exception as e: print('$1 $2 failed', e)
PY
done
for wt in q4_k q6_k q5_k q3_k; do timeout 400 python bench.py --wtype $wt --no-cpu-baseline --prefill-steps 0 --steps 64 > $O/r06_bench_$wt.json 2> $O/r06_bench_$wt.err; python - <<PY
import json
try:
    d=json.loads(open('$O/r06_bench_$wt.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('$wt', d['value'], {k:(v['us_per_launch'], v['frac']) for k,v in r['per_kind'].items()}, 'whole', r['whole_token']['frac'], 'parity', d['parity_check']['passed'], d['parity_check'].get('per_layer',{}).get('worst_max'))
except Exception as e: print('$wt failed', e)
PY
done
