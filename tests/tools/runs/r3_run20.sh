#!/bin/bash
# round-3 run 20: per-kernel times of the 512-token prefill step
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r3; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_p
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o p -- python $R/bench.py --mode prefill --weights blocks --no-cpu-baseline > $R/gpurun_out/r3/prefill20_line.json 2> /dev/null
cd $R; python tests/tools/kstats.py /tmp/prof_p > gpurun_out/r3/prefill20_kernel_stats.txt 2>&1; head -24 gpurun_out/r3/prefill20_kernel_stats.txt | cut -c1-60,100-175
