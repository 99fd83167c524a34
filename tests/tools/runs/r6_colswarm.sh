#!/bin/bash
# round 6: the chunk plan's norm launches warm the next k_mmq_cols launch's leading groups (ColsWarm), tests + A/B of the prompt feed
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6
timeout 1500 python -m pytest tests/test_llama_gpu.py tests/test_mmq_cols_gpu.py tests/test_c3_gpu.py tests/test_split_gpu.py -x -q 2>&1 | grep -v '^ROCm\|^Host\|^Librccl' | tail -4
for MB in 0 24 0 24 12 36; do
  GGML_HIP_WARM_MB=$MB timeout 300 python bench.py --mode feed --weights blocks --steps 5 > $O/cw_feed_$MB.json 2> $O/cw_feed_$MB.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/cw_feed_$MB.json').read().strip().splitlines()[-1])
    print('cols warm $MB feed', d['value'], d.get('ms_per_chunk'))
except Exception as e: print('$MB failed', e)
PY
done
cd /tmp; rm -rf /tmp/prof_cw; GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_cw -o f -- python $GRAFT_REPO_ROOT/bench.py --mode feed --weights blocks --steps 5 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python tests/tools/kstats.py /tmp/prof_cw 2>&1 | head -12
