#!/bin/bash
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
echo "== full suite"; timeout 900 python -X faulthandler -m pytest tests -q -m gpu -x -v 2>&1 > gpurun_out/r3/suite6.txt; grep -n "Fatal\|FAILED\|passed\|failed\|Segmentation\|Aborted\|libggml_hip:" gpurun_out/r3/suite6.txt | head; grep -n "PASSED\|FAILED" gpurun_out/r3/suite6.txt | tail -3; grep -n -A12 "Fatal Python error\|Current thread" gpurun_out/r3/suite6.txt | head -50
