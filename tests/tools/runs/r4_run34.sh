#!/bin/bash
# round-4 run 34: the other BASELINE configs on one GPU with this round's kernels (13B Q5_1, 65B Q8_0 decode lines; random valid
# blocks), the in-process split harness (7B over 2 and 4 virtual slots) and the one-rank selftest
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
for cfg in "13b q5_1" "65b q8_0"; do set -- $cfg
timeout 600 python bench.py --model $1 --wtype $2 --weights blocks --no-cpu-baseline --prefill-steps 0 --steps 48 > gpurun_out/r4/r04_bench_$1_$2.json 2> gpurun_out/r4/r04_bench_$1_$2.err; tail -2 gpurun_out/r4/r04_bench_$1_$2.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r4/r04_bench_$1_$2.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('$1 $2', d['value'], r['frac'], r['all_matvecs_per_token']['frac'], r['whole_token']['frac'], d['parity_check']['passed'], d['config']['long_context'], d['config']['prompt_feed']['steady']['tokens_per_s'])
except Exception as e: print('$1 $2 failed', e)
PY
done
for G in 2 4; do
timeout 400 python bench.py --mode split --split $G --weights blocks --steps 64 > gpurun_out/r4/r04_split$G.json 2> gpurun_out/r4/r04_split$G.err; tail -2 gpurun_out/r4/r04_split$G.err | cut -c1-200; cut -c1-700 gpurun_out/r4/r04_split$G.json
done
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 timeout 300 python -m llm_amd.pipeline --selftest 2>&1 | tail -1 | cut -c1-500
