#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6; export TMPDIR=/tmp
O=gpurun_out/r6
for MB in 0 8 12 16 20 0 8 12 16 20 0 12; do
  GGML_HIP_WARM_MB=$MB timeout 300 python bench.py --mode feed --weights blocks --steps 8 > $O/cw2_feed_$MB.json 2> $O/cw2_feed_$MB.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/cw2_feed_$MB.json').read().strip().splitlines()[-1])
    print('cols warm $MB feed', d['value'], d.get('ms_per_chunk'))
except Exception as e: print('$MB failed', e)
PY
done
