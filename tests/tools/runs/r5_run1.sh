#!/bin/bash
# round-5 run 1: baseline of the r04 tree on today's box — in-kernel timelines of the decode launches (all workgroups), the
# aggregate rate of concurrent sessions on virtual slots (one model copy each), the counter list, two instruction-side PMC passes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 200 python tests/tools/fused_timeline.py 128 > gpurun_out/r5/fused_timeline_128.txt 2>&1
timeout 200 python tests/tools/timeline.py 7b 256 > gpurun_out/r5/timeline_256.txt 2>&1
tail -30 gpurun_out/r5/timeline_256.txt
SESSIONS_SHARED=0 timeout 300 python tests/tools/sessions_probe.py 1 2 4 > gpurun_out/r5/sessions_unshared.txt 2>&1
tail -4 gpurun_out/r5/sessions_unshared.txt | cut -c1-600
cd /tmp
rocprofv3 -L > $R/gpurun_out/r5/counters.txt 2>&1
rm -rf /tmp/pmc_i1 /tmp/pmc_i2
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace -d /tmp/pmc_i1 -o p -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --prefill-steps 0 --weights blocks --roofline-steps 1 --no-parity-check > /dev/null 2> $R/gpurun_out/r5/pmc_i1.err
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS --kernel-trace -d /tmp/pmc_i2 -o p -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --prefill-steps 0 --weights blocks --roofline-steps 1 --no-parity-check > /dev/null 2> $R/gpurun_out/r5/pmc_i2.err
cd $R
for k in k_qkv_attn k_mmvq_big; do
  python tests/tools/pmc_kernel.py /tmp/pmc_i1 "%$k%" >> gpurun_out/r5/pmc_insts.txt 2>&1
  python tests/tools/pmc_kernel.py /tmp/pmc_i2 "%$k%" >> gpurun_out/r5/pmc_insts.txt 2>&1
done
head -60 gpurun_out/r5/pmc_insts.txt
tail -3 gpurun_out/r5/pmc_i1.err gpurun_out/r5/pmc_i2.err | cut -c1-300
