cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof_pre /tmp/pmc1
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_pre -o bench -- python bench.py --mode prefill --steps 3 --warmup 1 > gpurun_out/prof_pre.log 2>&1
python tests/tools/shape_stats.py /tmp/prof_pre '%k_mmq%'
python tests/tools/kstats.py /tmp/prof_pre > gpurun_out/prof_pre_stats.txt; head -14 gpurun_out/prof_pre_stats.txt
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d /tmp/pmc1 -o bench -- python bench.py --mode prefill --steps 1 --warmup 1 > gpurun_out/pmc1.log 2>&1
python tests/tools/pmc_kernel.py /tmp/pmc1 '%k_mmq%' | tee gpurun_out/pmc_mmq_dma.txt
