#!/bin/bash
# round-4 run 24: in-kernel timeline of k_qkv_attn at 128 and 400 positions (what the attention tail behind the mat-vec consists of)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4; export TMPDIR=/tmp
timeout 300 python tests/tools/fused_timeline.py 128 > gpurun_out/r4/r04_fused_timeline_128.txt 2>&1; tail -20 gpurun_out/r4/r04_fused_timeline_128.txt
timeout 300 python tests/tools/fused_timeline.py 400 > gpurun_out/r4/r04_fused_timeline_400.txt 2>&1; tail -20 gpurun_out/r4/r04_fused_timeline_400.txt
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null
