#!/bin/bash
# round-4 run 15: tail-split prompt GEMM: parity tests, prefill bench A/B (mmq_sk 0 / rule at 92 % / rule at 80 %)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4
timeout 900 python -X faulthandler -m pytest tests/test_tailsplit_gpu.py tests/test_mmq256_gpu.py tests/test_c3_gpu.py -q -x -m gpu -s > gpurun_out/r4/run15_pytest.txt 2>&1
grep -E "passed|failed|Error|error|tail split vs|giveup|assert" gpurun_out/r4/run15_pytest.txt | head -20 | cut -c1-250
for cfg in "0 92" "1 92" "0 92" "1 92"; do set -- $cfg
GGML_HIP_MMQ_SK=$1 GGML_HIP_MMQ_SK_PCT=$2 timeout 600 python bench.py --mode prefill --steps 10 --warmup 3 > gpurun_out/r4/bench15_sk$1_$2.json 2> gpurun_out/r4/bench15_sk$1_$2.err; tail -2 gpurun_out/r4/bench15_sk$1_$2.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r4/bench15_sk$1_$2.json').read().strip().splitlines()[-1]); r=d['roofline']
    print('sk=$1 pct=$2', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['class_ms_per_step'])
except Exception as e: print('sk=$1 failed', e)
PY
done
