#!/bin/bash
# round-4 run 18: after retiring k_mmq_dma / k_mmq_dma_p / the X8 variants / the prefetch warm-up: whole GPU suite; K-quant
# bench lines on the generic executor with the oracle's reversed super-block order as the band
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests -q -m gpu -x > gpurun_out/r4/suite18.txt 2>&1; grep -E "passed|failed|Error|error" gpurun_out/r4/suite18.txt | head -8 | cut -c1-300
for wt in q4_k q6_k; do
timeout 400 python bench.py --wtype $wt --no-cpu-baseline --prefill-steps 0 --steps 32 --warmup 4 > gpurun_out/r4/bench18_$wt.json 2> gpurun_out/r4/bench18_$wt.err; tail -3 gpurun_out/r4/bench18_$wt.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r4/bench18_$wt.json').read().strip().splitlines()[-1])
    print('$wt', d.get('value'), d.get('ms_per_step'), json.dumps(d.get('roofline'))[:700], json.dumps(d.get('parity_check'))[:500])
except Exception as e: print('$wt failed', e)
PY
done
