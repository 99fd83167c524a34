#!/bin/bash
# round-3 run 18: chunks of 9..31 tokens in passes; whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3; export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_mmq_cols_gpu.py -q -m gpu -x > gpurun_out/r3/s18_cols.txt 2>&1; grep -v "^  File" gpurun_out/r3/s18_cols.txt | tail -6 | cut -c1-300
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu > gpurun_out/r3/suite18.txt 2>&1; grep -v "^  File" gpurun_out/r3/suite18.txt | tail -8 | cut -c1-300
for nb in 16 24 31; do timeout 300 python bench.py --mode feed --weights blocks --steps 3 --n-batch $nb 2>/dev/null | cut -c1-220; done
