#!/bin/bash
# round-5 run 12: counters of k_mmvq_kbig (Q4_K decode): is it the VALU, LDS conflicts, or waiting for memory?
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r5; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pmc_k1 /tmp/pmc_k2
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/pmc_k1 -o p -- python $R/bench.py --wtype q4_k --steps 16 --warmup 2 --no-cpu-baseline --prefill-steps 0 --roofline-steps 1 --no-parity-check > /dev/null 2> $R/gpurun_out/r5/pmc_k1.err
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM --kernel-trace -d /tmp/pmc_k2 -o p -- python $R/bench.py --wtype q4_k --steps 16 --warmup 2 --no-cpu-baseline --prefill-steps 0 --roofline-steps 1 --no-parity-check > /dev/null 2> $R/gpurun_out/r5/pmc_k2.err
cd $R
python tests/tools/pmc_kernel.py /tmp/pmc_k1 "%k_mmvq_kbig%" > gpurun_out/r5/pmc_kbig.txt 2>&1
python tests/tools/pmc_kernel.py /tmp/pmc_k2 "%k_mmvq_kbig%" >> gpurun_out/r5/pmc_kbig.txt 2>&1
cat gpurun_out/r5/pmc_kbig.txt | cut -c1-130
tail -n 2 gpurun_out/r5/pmc_k1.err; tail -n 2 gpurun_out/r5/pmc_k2.err
