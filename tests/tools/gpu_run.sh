cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for pf in 0 16 32 64; do
  GGML_HIP_PREFETCH=$pf timeout 300 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/bench_pf$pf.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_pf$pf.json")); r=d["roofline"]["per_kind"]
print("prefetch $pf MB:", d["value"], "tok/s", d["ms_per_step"], "ms; device", d["config"]["host_split_per_token"]["device_wait_ms"], {k:v["us_per_launch_incl_boundary"] for k,v in r.items()})
PY
done
GGML_HIP_PREFETCH=32 timeout 300 python tests/tools/timeline.py 7b > gpurun_out/timeline_pf32.txt 2>&1; grep "^avg\|^gap\|token span\|attention" gpurun_out/timeline_pf32.txt
