# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tests/tools/launch_probe.py 7b q4_0 2>&1 | tail -9 | tee gpurun_out/r02_launch_probe_b.txt
timeout 300 python tests/tools/timeline.py 7b 256 2>&1 | tail -38 | tee gpurun_out/r02_timeline_b.txt
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_ops_gpu.py -q -x 2>&1 | tail -3
