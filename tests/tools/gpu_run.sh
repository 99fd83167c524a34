# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for r in 3 4 6; do GGML_HIP_LIB=$PWD/llm_amd/libggml_hip_ring$r.so timeout 300 python bench.py --mode prefill --steps 5 --warmup 2 --weights blocks > gpurun_out/r02_prefill_ring$r.json 2>gpurun_out/r02_prefill.err; python - <<PY
import json
d=json.load(open("gpurun_out/r02_prefill_ring$r.json")); print("ring=$r", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["class_ms_per_step"])
PY
done
GGML_HIP_LIB=$PWD/llm_amd/libggml_hip_ring6.so timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "prefill" 2>&1 | tail -4
