#!/bin/bash
# round-4 run 26: where the switch between the fused short-context attention and the split attention belongs now (ctx_sweep.py);
# rocprofv3 kernel stats of the K plan (Q4_K decode)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 600 python tests/tools/ctx_sweep.py > gpurun_out/r4/r04_ctx_sweep.txt 2>&1; tail -14 gpurun_out/r4/r04_ctx_sweep.txt
cd /tmp; rm -rf /tmp/prof_k
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o k -- python $R/bench.py --wtype q4_k --steps 48 --warmup 4 --no-cpu-baseline --prefill-steps 0 --no-parity-check > $R/gpurun_out/r4/r04_q4_k_line_under_rocprof.json 2> /dev/null
cd $R; python tests/tools/kstats.py /tmp/prof_k > gpurun_out/r4/r04_decode7b_q4_k_kernel_stats.txt 2>&1; head -16 gpurun_out/r4/r04_decode7b_q4_k_kernel_stats.txt | cut -c1-150
