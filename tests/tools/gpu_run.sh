cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_llama_gpu.py tests/test_entry_gpu.py -q -x 2>&1 | tail -2
for f in 0 1 0 1; do
  GGML_HIP_ATTN_WO=$f timeout 300 python bench.py --no-cpu-baseline --steps 128 > gpurun_out/bench_f$f.json 2>gpurun_out/bench_f$f.err || tail -3 gpurun_out/bench_f$f.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_f$f.json")); r=d["roofline"]["per_kind"]
print("attn_wo $f:", d["value"], "tok/s", d["ms_per_step"], "ms; device", d["config"]["host_split_per_token"]["device_wait_ms"])
PY
done
timeout 300 python tests/tools/timeline.py 7b > gpurun_out/timeline.txt 2>&1; grep "attn+wo\|token span" gpurun_out/timeline.txt
