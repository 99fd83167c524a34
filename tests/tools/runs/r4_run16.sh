#!/bin/bash
# round-4 run 16 (re-entry after the container was re-created): smoke, whole GPU suite, driver-format bench line, rocprofv3
# kernel stats of the decode leg (fused k_qkv_attn path) and of the prefill leg
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | cut -c1-200
timeout 900 python -X faulthandler -m pytest tests -q -m gpu > gpurun_out/r4/suite16.txt 2>&1; grep -v "^  File" gpurun_out/r4/suite16.txt | tail -4 | cut -c1-300
timeout 600 python bench.py > gpurun_out/r4/r04_bench_a.json 2> gpurun_out/r4/r04_bench_a.err; tail -2 gpurun_out/r4/r04_bench_a.err
cd /tmp; rm -rf /tmp/prof_d /tmp/prof_p
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o d -- python $R/bench.py --steps 64 --warmup 8 --no-cpu-baseline --prefill-steps 0 --weights blocks > $R/gpurun_out/r4/r04_bench_line_under_rocprof.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o p -- python $R/bench.py --mode prefill --weights blocks --no-cpu-baseline > $R/gpurun_out/r4/r04_prefill_line_under_rocprof.json 2> /dev/null
cd $R
python tests/tools/kstats.py /tmp/prof_d > gpurun_out/r4/r04_decode7b_kernel_stats.txt 2>&1
python tests/tools/kstats.py /tmp/prof_p > gpurun_out/r4/r04_prefill7b_kernel_stats.txt 2>&1
head -14 gpurun_out/r4/r04_decode7b_kernel_stats.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/r04_bench_a.json').read().strip().splitlines()[-1]); c=d['config']
print(d['value'], d['ms_per_step_min_median_max'], d['roofline']['frac'], d['parity_check']['passed'])
print({k: v['us_per_launch'] for k, v in d['roofline']['per_kind'].items()}, d['roofline']['all_matvecs_per_token']['frac'], d['roofline']['whole_token']['frac'])
print(c['prefill']['tokens_per_s'], c['prompt_feed']['steady']['tokens_per_s'], c.get('long_context'))
PY
