#!/bin/bash
# round-4 run 28: does the session's n_batch (the size of the host arena a decode graph is built in) show in the decode rate?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
for nb in 8 512 8 512; do
CTX_SWEEP_NBATCH=$nb timeout 600 python tests/tools/ctx_sweep.py 1 > gpurun_out/r4/r04_ctx_sweep_nb$nb.txt 2>&1; tail -11 gpurun_out/r4/r04_ctx_sweep_nb$nb.txt | tr '\n' ' '; echo
done
