#!/bin/bash
# round-4 run 17: where the K-quant models stand before they get a decode plan (generic executor, node by node)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
for wt in q4_k q6_k; do
timeout 400 python bench.py --wtype $wt --no-cpu-baseline --prefill-steps 0 --steps 32 --warmup 4 > gpurun_out/r4/bench17_$wt.json 2> gpurun_out/r4/bench17_$wt.err; tail -3 gpurun_out/r4/bench17_$wt.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r4/bench17_$wt.json').read().strip().splitlines()[-1])
    print('$wt', d['value'], d['ms_per_step'], json.dumps(d.get('roofline'))[:600], d.get('parity_check'))
except Exception as e: print('$wt failed', e)
PY
done
