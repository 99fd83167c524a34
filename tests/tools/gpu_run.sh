cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_fullsize_gpu.py tests/test_entry_gpu.py -q -x 2>&1 | tail -3
for w in 16 0; do
  GGML_HIP_BIG_WAVES=$w timeout 300 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/bench_w$w.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_w$w.json")); r=d["roofline"]["per_kind"]
print("waves $w:", d["value"], "tok/s", d["ms_per_step"], "ms; device", d["config"]["host_split_per_token"]["device_wait_ms"], {k:(v["us_per_launch_incl_boundary"], v["us_in_kernel"]) for k,v in r.items()})
PY
done
timeout 300 python tests/tools/timeline.py 7b > gpurun_out/timeline.txt 2>&1; grep "^avg\|^gap\|token span\|attention" gpurun_out/timeline.txt
