#!/bin/bash
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
for f in tests/test_split_gpu.py tests/test_mmq_cols_gpu.py tests/test_prompt_plan_gpu.py; do
  echo "== $f"; timeout 600 python -X faulthandler -m pytest $f -q -m gpu > gpurun_out/r3/s8_$(basename $f).txt 2>&1; tail -n 1 gpurun_out/r3/s8_$(basename $f).txt; grep -m6 "^FAILED\|Fatal Python\|libggml_hip:" gpurun_out/r3/s8_$(basename $f).txt
done
echo "== whole suite in one process"; timeout 900 python -X faulthandler -m pytest tests -q -m gpu > gpurun_out/r3/suite8.txt 2>&1; tail -n 2 gpurun_out/r3/suite8.txt; grep -n -m5 "^FAILED" gpurun_out/r3/suite8.txt; grep -n -B25 "Fatal Python error" gpurun_out/r3/suite8.txt | head -60
