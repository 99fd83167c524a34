#!/bin/bash
# where a 48-token prompt chunk of a K-quant model spends its 6 ms on the prompt plan (kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_s
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python $R/bench.py --wtype q4_k --mode feed --n-batch 48 --steps 4 > /dev/null 2>&1
cd $R; python tests/tools/kstats.py /tmp/prof_s 2>&1 | head -24
