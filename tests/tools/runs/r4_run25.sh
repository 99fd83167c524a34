#!/bin/bash
# round-4 run 25: the end-of-round measurement set (profile_round.sh r04) + the whole suite + smoke
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | cut -c1-200
timeout 900 python -X faulthandler -m pytest tests -q -m gpu > gpurun_out/r04_suite_final.txt 2>&1; grep -E "passed|failed|Error" gpurun_out/r04_suite_final.txt | head -5 | cut -c1-300
bash tests/tools/profile_round.sh r04 > gpurun_out/profile_round_r04.log 2>&1
tail -40 gpurun_out/profile_round_r04.log | cut -c1-400
