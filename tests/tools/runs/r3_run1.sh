#!/bin/bash
# round-3 GPU call 1: new kernel + new tests, shape timings, whole suite, bench
mkdir -p gpurun_out/r3
cd $GRAFT_REPO_ROOT 2>/dev/null || true
export PYTHONUNBUFFERED=1
echo "== mmq256 tests"; timeout 600 python -m pytest tests/test_mmq256_gpu.py -x -q 2>&1 | tail -15
echo "== shapes"; timeout 300 python tests/tools/gemm_shapes.py 512 2>&1 | tee gpurun_out/r3/shapes1.txt | tail -8
echo "== c3 tests"; timeout 900 python -m pytest tests/test_c3_gpu.py -x -q -s 2>&1 | tee gpurun_out/r3/c3.txt | tail -40
echo "== full suite"; timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
echo "== bench"; timeout 900 python bench.py --steps 128 --warmup 8 > gpurun_out/r3/bench1.json 2> gpurun_out/r3/bench1.err; tail -c 3000 gpurun_out/r3/bench1.json; tail -5 gpurun_out/r3/bench1.err
