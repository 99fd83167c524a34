#!/bin/bash
# round-5 final measurement set: whole GPU suite + smoke, the end-of-round profile set (bench line, rocprofv3 kernel stats, PMC
# traffic incl. the fused launch, prefill counters, feed trace, timelines), the other configs on one GPU, sessions, K-quant lines
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5f; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -30 > gpurun_out/r5f/r05_suite_final.txt
tail -5 gpurun_out/r5f/r05_suite_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tests/tools/profile_round.sh r05 > gpurun_out/r5f/profile_round.log 2>&1
mv gpurun_out/r05_* gpurun_out/r5f/ 2>/dev/null
tail -c 900 gpurun_out/r5f/r05_bench_final.json
timeout 200 python tests/tools/wo_timeline.py 128 > gpurun_out/r5f/r05_wo_timeline_128.txt 2>&1
timeout 300 python bench.py --mode sessions --sessions 1,2,3 --weights blocks --steps 128 > gpurun_out/r5f/r05_sessions.json 2> gpurun_out/r5f/r05_sessions.err
for wt in q4_k q6_k; do timeout 300 python bench.py --wtype $wt --no-cpu-baseline --prefill-steps 0 --steps 64 > gpurun_out/r5f/r05_bench_$wt.json 2> gpurun_out/r5f/r05_bench_$wt.err; done
for cfg in "13b q5_1" "65b q8_0"; do set -- $cfg
timeout 600 python bench.py --model $1 --wtype $2 --weights blocks --no-cpu-baseline --prefill-steps 0 --steps 48 > gpurun_out/r5f/r05_bench_$1_$2.json 2> gpurun_out/r5f/r05_bench_$1_$2.err; tail -n 2 gpurun_out/r5f/r05_bench_$1_$2.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r5f/r05_bench_$1_$2.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('$1 $2', d['value'], r['kernel_kind'], r['frac'], {k:(v['launches'], v['us_per_launch'], v['frac']) for k,v in r['per_kind'].items()}, r['whole_token']['frac'], d['parity_check']['passed'], d['config']['long_context'], d['config']['call_sequence']['reference_call_sequence']['tokens_per_s'])
except Exception as e: print('$1 $2 failed', e)
PY
done
python - <<'PY'
import json
for f in ('r05_sessions','r05_bench_q4_k','r05_bench_q6_k'):
    try:
        d=json.loads(open(f'gpurun_out/r5f/{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], [(r['sessions'], r['aggregate_tokens_per_s']) for r in d.get('runs',[])])
    except Exception as e: print(f, 'failed', e)
PY
