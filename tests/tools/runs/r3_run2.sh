#!/bin/bash
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
echo "== shapes 512"; timeout 300 python tests/tools/gemm_shapes.py 512 2>&1 | tee gpurun_out/r3/shapes2.txt | tail -40
echo "== shapes big"; timeout 300 python tests/tools/gemm_shapes.py 4096 big 2>&1 | tee gpurun_out/r3/shapes2_big.txt | tail -20
echo "== c3 tests"; timeout 900 python -m pytest tests/test_c3_gpu.py -q -s 2>&1 | tee gpurun_out/r3/c3.txt | grep -E "7B|x.*N=512|passed|failed|Error|assert" | head -40
echo "== full suite"; timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_c3_gpu.py 2>&1 | tail -8
echo "== bench"; timeout 900 python bench.py --steps 128 --warmup 8 > gpurun_out/r3/bench2.json 2> gpurun_out/r3/bench2.err; tail -c 2500 gpurun_out/r3/bench2.json; tail -5 gpurun_out/r3/bench2.err
echo "== pmc lm_head t256"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --pmc $pass --kernel-trace -d /tmp/pmc_$i -o p -- python $R/tests/tools/gemm_one.py 32000 4096 512 4 mmq_t256=2 > /dev/null 2>&1
  python $R/tests/tools/pmcstats.py /tmp/pmc_$i 2>&1 | grep -i "mmq_w16_256" | awk '{print $1, $(NF-5), $(NF-4), $(NF-3), $NF}' | tee -a $R/gpurun_out/r3/pmc_t256_lmhead.txt
done
