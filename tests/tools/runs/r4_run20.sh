#!/bin/bash
# round-4 run 20: K plan tests again (merged wq|wk|wv / w1|w3 launches held bit-exact to one launch per matrix), whole suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests/test_kquant_plan_gpu.py -q -m gpu -s > gpurun_out/r4/run20_pytest.txt 2>&1
grep -E "passed|failed|Error|error|worst|assert" gpurun_out/r4/run20_pytest.txt | head -50 | cut -c1-250
timeout 900 python -X faulthandler -m pytest tests -q -m gpu -x > gpurun_out/r4/suite20.txt 2>&1; grep -E "passed|failed|Error|error" gpurun_out/r4/suite20.txt | head -8 | cut -c1-300
