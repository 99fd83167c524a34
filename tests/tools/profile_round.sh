#!/bin/bash
# End-of-round measurement set (run on the MI355X box through gpurun): the driver's bench line, the rocprofv3 kernel
# traces of the decode and prefill legs, and the PMC traffic passes.  Outputs under gpurun_out/; the summaries are
# copied into profiles/ by hand.   bash tests/tools/profile_round.sh r02
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rNN}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/${TAG}_bench_final.json 2> gpurun_out/${TAG}_bench_final.err
cd /tmp
rm -rf /tmp/prof_d /tmp/prof_p /tmp/pmc_f /tmp/pmc_w
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o d -- python $R/bench.py --steps 64 --warmup 8 --no-cpu-baseline --prefill-steps 0 --weights blocks > $R/gpurun_out/${TAG}_bench_line_under_rocprof.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o p -- python $R/bench.py --mode prefill --weights blocks --no-cpu-baseline > $R/gpurun_out/${TAG}_prefill_line_under_rocprof.json 2> /dev/null
# PMC passes: --headline-only + a 248-token prompt = every decode dispatch at 248...266 positions, the context the default
# line's roofline bytes are stated for (128 + 8 + 128 tokens)
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f -o f -- python $R/bench.py --steps 16 --warmup 2 --prompt 248 --headline-only --no-parity-check --no-cpu-baseline --prefill-steps 0 --weights blocks --roofline-steps 1 > /dev/null 2>&1
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_w -o w -- python $R/bench.py --steps 16 --warmup 2 --prompt 248 --headline-only --no-parity-check --no-cpu-baseline --prefill-steps 0 --weights blocks --roofline-steps 1 > /dev/null 2>&1
rm -rf /tmp/pmc_p1 /tmp/pmc_p2 /tmp/prof_f
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --kernel-trace -d /tmp/pmc_p1 -o p -- python $R/bench.py --mode prefill --steps 1 --warmup 1 --weights blocks --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum SQ_INSTS_VMEM_RD --kernel-trace -d /tmp/pmc_p2 -o p -- python $R/bench.py --mode prefill --steps 1 --warmup 1 --weights blocks --no-cpu-baseline > /dev/null 2>&1
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o f -- python $R/bench.py --mode feed --weights blocks --steps 5 > $R/gpurun_out/${TAG}_feed8_line_under_rocprof.json 2> /dev/null
cd $R
for k in k_mmq_w16_256 k_mmq_w16_p8 k_p_attn; do
  python tests/tools/pmc_kernel.py /tmp/pmc_p1 "%$k%" >> gpurun_out/${TAG}_prefill_pmc.txt 2>&1
  python tests/tools/pmc_kernel.py /tmp/pmc_p2 "%$k%" >> gpurun_out/${TAG}_prefill_pmc.txt 2>&1
done
python tests/tools/kstats.py /tmp/prof_f > gpurun_out/${TAG}_feed8_kernel_stats.txt 2>&1
timeout 200 python tests/tools/cols_timeline.py 64 8 > gpurun_out/${TAG}_cols_timeline.txt 2>&1
timeout 200 python tests/tools/pattn_timeline.py 512 0 > gpurun_out/${TAG}_pattn_timeline.txt 2>&1
python tests/tools/kstats.py /tmp/prof_d > gpurun_out/${TAG}_decode7b_kernel_stats.txt 2>&1
python tests/tools/kstats.py /tmp/prof_p > gpurun_out/${TAG}_prefill7b_kernel_stats.txt 2>&1
python tests/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w "bench.py --prompt 248 --warmup 2 --steps 16 --headline-only --weights blocks: single-token decode dispatches at 248...266 positions of context (+ one roofline replay each)" 257 > gpurun_out/${TAG}_pmc_traffic.json 2> gpurun_out/${TAG}_pmc_traffic.err
head -12 gpurun_out/${TAG}_decode7b_kernel_stats.txt
head -12 gpurun_out/${TAG}_prefill7b_kernel_stats.txt
cat gpurun_out/${TAG}_pmc_traffic.json | head -40
cat gpurun_out/${TAG}_prefill_pmc.txt | head -40
tail -c 1500 gpurun_out/${TAG}_bench_final.json
