# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pmc_g /tmp/pmc_g2
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_g -o g -- python $R/bench.py --mode prefill --steps 1 --warmup 1 --weights blocks --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM --kernel-trace -d /tmp/pmc_g2 -o g -- python $R/bench.py --mode prefill --steps 1 --warmup 1 --weights blocks --no-cpu-baseline > /dev/null 2>&1
cd $R
O=gpurun_out/r02_prefill_mmq_w16_pmc.txt
echo "# rocprofv3 --pmc (two passes) --kernel-trace -- python bench.py --mode prefill --steps 1 --warmup 1 --weights blocks   (round 2; k_mmq_w16_p8, LLaMA-7B Q4_0, 512-token batch); per-dispatch averages over counter instances" > $O
python tests/tools/pmc_kernel.py /tmp/pmc_g '%k_mmq_w16_p8%' >> $O 2>&1
python tests/tools/pmc_kernel.py /tmp/pmc_g2 '%k_mmq_w16_p8%' >> $O 2>&1
cat $O
