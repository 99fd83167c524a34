#!/bin/bash
# round-3 run 17: cols with batched kernarg loads; experiment: kernarg preload build (libggml_hip_kpl.so) on decode
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mmq_cols_gpu.py -q -m gpu > gpurun_out/r3/s17_cols.txt 2>&1; tail -4 gpurun_out/r3/s17_cols.txt | cut -c1-300
timeout 300 python tests/tools/cols_timeline.py 64 8 > gpurun_out/r3/cols_timeline17.txt 2>&1; tail -7 gpurun_out/r3/cols_timeline17.txt
timeout 300 python bench.py --mode feed --weights blocks --steps 5 > gpurun_out/r3/feed8_17.json 2>gpurun_out/r3/feed8_17.err; cut -c1-200 gpurun_out/r3/feed8_17.json
echo "== decode, production build"
timeout 300 python bench.py --weights blocks --steps 128 --no-cpu-baseline --prefill-steps 0 --no-parity-check 2>/dev/null | cut -c1-260
echo "== decode, kernarg preload build"
GGML_HIP_LIB=$R/llm_amd/libggml_hip_kpl.so timeout 300 python bench.py --weights blocks --steps 128 --no-cpu-baseline --prefill-steps 0 --no-parity-check 2>gpurun_out/r3/kpl.err | cut -c1-260; tail -3 gpurun_out/r3/kpl.err
GGML_HIP_LIB=$R/llm_amd/libggml_hip_kpl.so timeout 300 python bench.py --mode feed --weights blocks --steps 5 2>/dev/null | cut -c1-200
GGML_HIP_LIB=$R/llm_amd/libggml_hip_kpl.so timeout 600 python -m pytest tests/test_llama_gpu.py tests/test_ops_gpu.py -q -m gpu -x 2>&1 | tail -3 | cut -c1-300
