#!/bin/bash
# round-3 run 10: K-quant completion + split fix, whole GPU suite, default bench
mkdir -p gpurun_out/r3
for f in tests/test_kquant_gpu.py tests/test_split_gpu.py; do
  timeout 900 python -X faulthandler -m pytest $f -x -q -m gpu -s > gpurun_out/r3/s10_$(basename $f).txt 2>&1
  echo "== $f"; grep -v "^  File" gpurun_out/r3/s10_$(basename $f).txt | tail -8 | cut -c1-300
done
timeout 1200 python -X faulthandler -m pytest tests -q -m gpu > gpurun_out/r3/suite10.txt 2>&1
echo "== suite"; grep -v "^  File" gpurun_out/r3/suite10.txt | tail -12 | cut -c1-300
timeout 900 python bench.py > gpurun_out/r3/bench10.json 2> gpurun_out/r3/bench10.err
echo "== bench"; cut -c1-1500 gpurun_out/r3/bench10.json; tail -5 gpurun_out/r3/bench10.err
