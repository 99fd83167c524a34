cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_entry_gpu.py -q -x 2>&1 | tail -5
timeout 300 python bench.py --no-cpu-baseline --steps 128 > gpurun_out/bench_decode.json 2>gpurun_out/bench.err || tail -3 gpurun_out/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_decode.json"))
print(d["value"], "tok/s", d["ms_per_step"], "ms; device", d["config"]["host_split_per_token"]["device_wait_ms"], d["config"]["device_sampling"])
PY
