#!/bin/bash
# round-3 run 12: where the time of k_mmq_cols goes (in-kernel timeline + counters)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3; export TMPDIR=/tmp
timeout 300 python tests/tools/cols_timeline.py 64 8 > gpurun_out/r3/cols_timeline.txt 2>&1; cat gpurun_out/r3/cols_timeline.txt | tail -30
cd /tmp; rm -rf /tmp/pmc_c1 /tmp/pmc_c2
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/pmc_c1 -o c -- python $R/bench.py --mode feed --weights blocks --steps 1 > /dev/null 2>$R/gpurun_out/r3/pmc_c1.err
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM --kernel-trace -d /tmp/pmc_c2 -o c -- python $R/bench.py --mode feed --weights blocks --steps 1 > /dev/null 2>$R/gpurun_out/r3/pmc_c2.err
cd $R
python tests/tools/pmc_kernel.py /tmp/pmc_c1 '%k_mmq_cols%' > gpurun_out/r3/cols_pmc.txt 2>&1
python tests/tools/pmc_kernel.py /tmp/pmc_c2 '%k_mmq_cols%' >> gpurun_out/r3/cols_pmc.txt 2>&1
python tests/tools/pmc_kernel.py /tmp/pmc_c1 '%k_rmsnorm_quant%' >> gpurun_out/r3/cols_pmc.txt 2>&1
cat gpurun_out/r3/cols_pmc.txt; tail -3 gpurun_out/r3/pmc_c1.err gpurun_out/r3/pmc_c2.err
