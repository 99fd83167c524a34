#!/bin/bash
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
for f in tests/test_split_gpu.py tests/test_mmq_cols_gpu.py tests/test_abi.py tests/test_c3_gpu.py tests/test_device_tools_gpu.py tests/test_fullsize_gpu.py tests/test_gpt2_gpu.py tests/test_kquant_gpu.py tests/test_layer_chain_gpu.py tests/test_llama_gpu.py tests/test_mmq256_gpu.py tests/test_ops_gpu.py tests/test_prompt_plan_gpu.py; do
  echo "== $f"; timeout 600 python -X faulthandler -m pytest $f -q -m gpu 2>&1 > gpurun_out/r3/s7_$(basename $f).txt; tail -1 gpurun_out/r3/s7_$(basename $f).txt; grep -m3 "^FAILED\|Fatal Python\|libggml_hip:" gpurun_out/r3/s7_$(basename $f).txt
done
echo "== whole suite in one process"; timeout 900 python -X faulthandler -m pytest tests -q -m gpu 2>&1 > gpurun_out/r3/suite7.txt; tail -2 gpurun_out/r3/suite7.txt; grep -n -m3 "Fatal Python\|libggml_hip:\|^FAILED" gpurun_out/r3/suite7.txt; grep -n -A14 "Fatal Python error" gpurun_out/r3/suite7.txt | grep "File.*tests" | head -5
