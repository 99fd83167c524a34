#!/bin/bash
# the two PMC traffic passes of profile_round.sh alone (every decode dispatch at the roofline's context) + a bench line that reads them
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5f; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/pmc_f /tmp/pmc_w
A="--steps 16 --warmup 2 --prompt 248 --headline-only --no-parity-check --no-cpu-baseline --prefill-steps 0 --weights blocks --roofline-steps 1"
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f -o f -- python $R/bench.py $A > /dev/null 2>&1
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_w -o w -- python $R/bench.py $A > /dev/null 2>&1
cd $R
python tests/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w "bench.py --prompt 248 --warmup 2 --steps 16 --headline-only --weights blocks: single-token decode dispatches at 248...266 positions of context (+ one roofline replay each)" > gpurun_out/r5f/r05_pmc_traffic.json 2> gpurun_out/r5f/r05_pmc_traffic.err
cat gpurun_out/r5f/r05_pmc_traffic.json; cat gpurun_out/r5f/r05_pmc_traffic.err | tail -3
cp gpurun_out/r5f/r05_pmc_traffic.json profiles/r05_pmc_traffic.json
timeout 600 python bench.py > gpurun_out/r5f/r05_bench_final.json 2> gpurun_out/r5f/r05_bench_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5f/r05_bench_final.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], r['kernel_kind'], r['frac'], r['traffic'], r['traffic_over_algo'], r['context_positions'], r['algo_bytes_per_launch'], r['traffic_per_kind'])
print(d['config']['call_sequence']['reference_call_sequence']['tokens_per_s'], d['config']['prefill']['tokens_per_s'], d['parity_check']['passed'])
PY
