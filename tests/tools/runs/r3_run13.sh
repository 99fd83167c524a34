#!/bin/bash
# round-3 run 13: cols prologue by DMA behind the ring, one-pass rmsnorm_quant: tests, timeline, feed rate
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3; export TMPDIR=/tmp
for f in tests/test_mmq_cols_gpu.py tests/test_llama_gpu.py tests/test_fullsize_gpu.py; do
  timeout 900 python -X faulthandler -m pytest $f -x -q -m gpu > gpurun_out/r3/s13_$(basename $f).txt 2>&1
  echo "== $f"; grep -v "^  File" gpurun_out/r3/s13_$(basename $f).txt | tail -4 | cut -c1-400
done
timeout 300 python tests/tools/cols_timeline.py 64 8 > gpurun_out/r3/cols_timeline13.txt 2>&1; cat gpurun_out/r3/cols_timeline13.txt | tail -16
timeout 300 python bench.py --mode feed --weights blocks --steps 5 > gpurun_out/r3/feed8_13.json 2>gpurun_out/r3/feed8_13.err; cat gpurun_out/r3/feed8_13.json
cd /tmp; rm -rf /tmp/prof_f
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o f -- python $R/bench.py --mode feed --weights blocks --steps 5 > /dev/null 2>&1
cd $R; python tests/tools/kstats.py /tmp/prof_f > gpurun_out/r3/feed8_13_kernel_stats.txt 2>&1; head -9 gpurun_out/r3/feed8_13_kernel_stats.txt | cut -c1-20,100-175
