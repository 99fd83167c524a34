#!/bin/bash
# round-4 run 11: green-state check with k_qkv_attn as the default decode launch: smoke, whole GPU suite, default bench line,
# rocprofv3 kernel stats of the decode leg
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | cut -c1-200
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu > gpurun_out/r4/suite11.txt 2>&1; grep -v "^  File" gpurun_out/r4/suite11.txt | tail -6 | cut -c1-300
timeout 900 python bench.py > gpurun_out/r4/bench11.json 2> gpurun_out/r4/bench11.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench11.json').read().strip().splitlines()[-1]); c=d['config']
print(d['value'], d['ms_per_step_min_median_max'], d['roofline']['frac'], d['parity_check']['passed'], d['parity_check']['max_over_std'], c['decode_launches']['qkv_and_attention_in_one_launch_tokens'])
print({k: v['us_per_launch'] for k, v in d['roofline']['per_kind'].items()}, d['roofline']['all_matvecs_per_token']['frac'], d['roofline']['whole_token']['frac'])
print(c['prefill']['tokens_per_s'], c['prompt_feed']['steady']['tokens_per_s'], c['long_context'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
tail -2 gpurun_out/r4/bench11.err
cd /tmp; rm -rf /tmp/prof_d
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o d -- python $R/bench.py --steps 64 --warmup 8 --no-cpu-baseline --prefill-steps 0 --weights blocks > $R/gpurun_out/r4/bench_line_under_rocprof.json 2> /dev/null
cd $R; python tests/tools/kstats.py /tmp/prof_d > gpurun_out/r4/decode7b_kernel_stats.txt 2>&1; head -14 gpurun_out/r4/decode7b_kernel_stats.txt | cut -c1-200
