#!/bin/bash
# round-3 run 11: K-quant llama test, 2-process pipeline test, prompt-feed profile at n_batch = 8
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3; export TMPDIR=/tmp
for f in tests/test_kquant_gpu.py tests/test_pipeline_2proc_gpu.py tests/test_llama_gpu.py; do
  timeout 900 python -X faulthandler -m pytest $f -x -q -m gpu -s > gpurun_out/r3/s11_$(basename $f).txt 2>&1
  echo "== $f"; grep -v "^  File" gpurun_out/r3/s11_$(basename $f).txt | tail -6 | cut -c1-400
done
timeout 300 python bench.py --mode feed --weights blocks --steps 5 > gpurun_out/r3/feed8.json 2>gpurun_out/r3/feed8.err; cat gpurun_out/r3/feed8.json
cd /tmp; rm -rf /tmp/prof_f
GGML_HIP_GRAPH=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o f -- python $R/bench.py --mode feed --weights blocks --steps 5 > $R/gpurun_out/r3/feed8_rocprof.json 2>/dev/null
cd $R; python tests/tools/kstats.py /tmp/prof_f > gpurun_out/r3/feed8_kernel_stats.txt 2>&1; head -25 gpurun_out/r3/feed8_kernel_stats.txt | cut -c1-175
