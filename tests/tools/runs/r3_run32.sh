#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_prompt_plan_gpu.py tests/test_c3_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED" | head
cd /tmp; rm -rf /tmp/prof_p
timeout 200 python $R/tests/tools/pattn_timeline.py plan 512 2>&1 | sed -n "1,3p;17,19p"; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o p -- python $R/bench.py --mode prefill --weights blocks --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
cd $R; python tests/tools/kstats.py /tmp/prof_p 2>&1 | head -8 | cut -c1-50,100-175
