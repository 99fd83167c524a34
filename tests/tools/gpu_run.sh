cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for lib in libggml_hip.so libggml_hip_xf9.so libggml_hip_xf31.so; do
  GGML_HIP_LIB=$GRAFT_REPO_ROOT/llm_amd/$lib timeout 300 python bench.py --no-cpu-baseline --steps 128 > gpurun_out/bench_$lib.json 2>gpurun_out/bench.err || tail -3 gpurun_out/bench.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_$lib.json")); r=d["roofline"]["per_kind"]
print("$lib:", d["value"], "tok/s; device", d["config"]["host_split_per_token"]["device_wait_ms"], {k:(v["us_per_launch_incl_boundary"], v["us_in_kernel"]) for k,v in r.items()})
PY
done
GGML_HIP_LIB=$GRAFT_REPO_ROOT/llm_amd/libggml_hip_xf31.so timeout 300 python tests/tools/timeline.py 7b > gpurun_out/timeline.txt 2>&1; grep "^avg\|token span" gpurun_out/timeline.txt
