# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -x -k dma 2>&1 | tail -15
timeout 600 python tests/tools/launch_probe.py 7b > gpurun_out/r02_probe2.txt 2>&1; echo "probe rc=$?"
tail -22 gpurun_out/r02_probe2.txt
GGML_HIP_BIG=3 timeout 300 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/r02_bench_big3.json 2>gpurun_out/r02_bench_big3.err; echo "bench rc=$?"; tail -3 gpurun_out/r02_bench_big3.err
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r02_bench_big3.json")); print(d["value"], d["ms_per_step"]); print({k:(v["us_per_launch_incl_boundary"]) for k,v in d["roofline"]["per_kind"].items()})
except Exception as e: print("no bench json", e)
PY
