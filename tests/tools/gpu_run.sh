cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/bench_decode.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/bench_decode.json")); print(d["value"], d["ms_per_step"]); print(d["config"]["host_split_per_token"])
PY
