cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python bench.py --model 13b --wtype q5_1 --no-cpu-baseline --steps 64 > gpurun_out/bench_13b_q5_1.json 2> gpurun_out/bench_13b.err; echo "rc=$?"; tail -2 gpurun_out/bench_13b.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_13b_q5_1.json")); r=d["roofline"]; c=d["config"]
print(d["metric"], d["value"], d["ms_per_step"], "gate:", r["achieved"], r["frac"], r["achieved_in_kernel"], {k:(v["us_per_launch_incl_boundary"], v["us_in_kernel"]) for k,v in r["per_kind"].items()}, c["long_context"], c["prep"])
PY
