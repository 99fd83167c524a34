#!/bin/bash
# round 5, last call: kernel traces of the K-quant legs and of the 13B line on the final tree, the suite's output, the driver-format line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5l; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -30 > gpurun_out/r5l/r05_suite_final.txt
tail -3 gpurun_out/r5l/r05_suite_final.txt
cd /tmp; rm -rf /tmp/pk_d /tmp/pk_p /tmp/p13
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk_d -o d -- python $R/bench.py --wtype q4_k --steps 48 --warmup 8 --no-cpu-baseline --prefill-steps 0 --headline-only > $R/gpurun_out/r5l/r05_q4_k_line_under_rocprof.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk_p -o p -- python $R/bench.py --wtype q4_k --mode prefill --no-cpu-baseline > $R/gpurun_out/r5l/r05_q4_k_prefill_line_under_rocprof.json 2> /dev/null
GGML_HIP_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p13 -o d -- python $R/bench.py --model 13b --wtype q5_1 --weights blocks --steps 48 --warmup 8 --no-cpu-baseline --prefill-steps 0 --headline-only > $R/gpurun_out/r5l/r05_13b_q5_1_line_under_rocprof.json 2> /dev/null
cd $R
python tests/tools/kstats.py /tmp/pk_d > gpurun_out/r5l/r05_decode7b_q4_k_kernel_stats.txt 2>&1
python tests/tools/kstats.py /tmp/pk_p > gpurun_out/r5l/r05_prefill7b_q4_k_kernel_stats.txt 2>&1
python tests/tools/kstats.py /tmp/p13 > gpurun_out/r5l/r05_decode13b_q5_1_kernel_stats.txt 2>&1
head -8 gpurun_out/r5l/r05_decode7b_q4_k_kernel_stats.txt; head -8 gpurun_out/r5l/r05_prefill7b_q4_k_kernel_stats.txt; head -8 gpurun_out/r5l/r05_decode13b_q5_1_kernel_stats.txt
timeout 600 python bench.py > gpurun_out/r5l/r05_bench_final.json 2> gpurun_out/r5l/r05_bench_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5l/r05_bench_final.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], r['kernel_kind'], r['frac'], r['traffic_over_algo'], d['parity_check']['passed'], d['config']['prefill']['tokens_per_s'], d['config']['call_sequence']['reference_call_sequence']['tokens_per_s'], d['cpu_baseline']['value'])
PY
