#!/bin/bash
# round-5 run 2: the safety changes (error word read back with every token, per-layer hand-off buffers, arena event log, weights
# shared by the slots of one device) under the whole GPU suite; sessions of one model on sibling slots
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -25 > gpurun_out/r5/pytest_run2.txt
tail -12 gpurun_out/r5/pytest_run2.txt
timeout 300 python bench.py --mode sessions --sessions 1,2,3,4 --weights blocks --steps 192 > gpurun_out/r5/sessions_shared.json 2> gpurun_out/r5/sessions_shared.err
tail -2 gpurun_out/r5/sessions_shared.err | cut -c1-300; cut -c1-1500 gpurun_out/r5/sessions_shared.json
