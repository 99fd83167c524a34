#!/bin/bash
# round 6: what the token costs when its launches go to the stream one by one instead of as one hipGraphLaunch (first kernel starts earlier, host enqueues ahead of the device)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
for GR in 1 0 1 0; do
  GGML_HIP_GRAPH=$GR timeout 300 python bench.py --headline-only --no-cpu-baseline --no-parity-check --prefill-steps 0 > gpurun_out/r6/graph_$GR.json 2> gpurun_out/r6/graph_$GR.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r6/graph_$GR.json').read().strip().splitlines()[-1])
    print('graph=$GR', d['value'], d['ms_per_step'], json.dumps(d['config']['host_split_per_token']))
except Exception as e: print('graph=$GR failed', e)
PY
done
