# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tests/tools/launch_probe.py 7b > gpurun_out/r02_probe1.txt 2>&1; echo "probe rc=$?"
GGML_HIP_BIG=2 timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r02_pytest_big2.txt; cat gpurun_out/r02_pytest_big2.txt
timeout 300 python tests/tools/timeline.py 7b 256 > gpurun_out/r02_timeline_big1.txt 2>&1
GGML_HIP_BIG=2 timeout 300 python tests/tools/timeline.py 7b 256 > gpurun_out/r02_timeline_big2.txt 2>&1
GGML_HIP_BIG=2 timeout 300 python tests/tools/timeline.py 7b > gpurun_out/r02_timeline4_big2.txt 2>&1
HIP_FORCE_DEV_KERNARG=0 timeout 300 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/r02_bench_kernarg0.json 2>/dev/null
HIP_FORCE_DEV_KERNARG=1 timeout 300 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/r02_bench_kernarg1.json 2>/dev/null
GGML_HIP_BIG=2 timeout 300 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/r02_bench_big2.json 2>/dev/null
tail -40 gpurun_out/r02_probe1.txt
