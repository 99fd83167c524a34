#!/bin/bash
mkdir -p gpurun_out/r3
export PYTHONUNBUFFERED=1
echo "== attn debug"; timeout 300 python tests/tools/attn_debug.py 2>&1 | tail -20
echo "== full suite (no fused attention)"; GGML_HIP_ATTN_FUSED=0 timeout 900 python -m pytest tests -q -m gpu -x 2>&1 > gpurun_out/r3/suite4.txt; grep -n -m5 "Fatal\|Error\|FAILED\|passed\|failed\|Segmentation\|Aborted" gpurun_out/r3/suite4.txt; grep -n -B2 -A25 "Fatal Python error" gpurun_out/r3/suite4.txt | head -60
