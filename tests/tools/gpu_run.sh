# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "prefill" 2>&1 | tail -12
timeout 600 python -m pytest tests/test_llama_gpu.py -q -x -k "prefill or gqa" 2>&1 | tail -6
for i8 in 1 0; do GGML_HIP_MMQ_I8=$i8 timeout 300 python bench.py --mode prefill --steps 5 --warmup 2 --weights blocks > gpurun_out/r02_prefill_i8_$i8.json 2>gpurun_out/r02_prefill.err; python - <<PY
import json
d=json.load(open("gpurun_out/r02_prefill_i8_$i8.json")); print("i8=$i8", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["class_ms_per_step"])
PY
done
