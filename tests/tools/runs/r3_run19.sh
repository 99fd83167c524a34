#!/bin/bash
# round-3 run 19: whole GPU suite with the library's last words kept in a file
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3; export TMPDIR=/tmp
rm -f gpurun_out/r3/fatal.log; export GGML_HIP_FATAL_LOG=$R/gpurun_out/r3/fatal.log
timeout 600 python -X faulthandler -m pytest tests/test_mmq_cols_gpu.py -q -m gpu -x > gpurun_out/r3/s19_cols.txt 2>&1; grep -v "^  File" gpurun_out/r3/s19_cols.txt | tail -4 | cut -c1-300
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu > gpurun_out/r3/suite19.txt 2>&1; grep -v "^  File" gpurun_out/r3/suite19.txt | tail -8 | cut -c1-300
echo "== fatal log"; cat gpurun_out/r3/fatal.log 2>/dev/null
