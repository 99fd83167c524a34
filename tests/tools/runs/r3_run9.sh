#!/bin/bash
# round-3 run 9: split test with the library's own stderr visible, then the files fixed after run 8
mkdir -p gpurun_out/r3
timeout 300 python -X faulthandler -m pytest tests/test_split_gpu.py -x -q -m gpu -s > gpurun_out/r3/s9_split.txt 2>&1
grep -v "^  File" gpurun_out/r3/s9_split.txt | head -40
for f in tests/test_mmq_cols_gpu.py tests/test_prompt_plan_gpu.py; do
  timeout 600 python -m pytest $f -q -m gpu > gpurun_out/r3/s9_$(basename $f).txt 2>&1; tail -3 gpurun_out/r3/s9_$(basename $f).txt
done
