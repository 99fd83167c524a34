# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for p in 1 0; do
GGML_HIP_PLAN_PROMPT=$p timeout 300 python bench.py --mode prefill --weights blocks --no-cpu-baseline > gpurun_out/r02_prefill_plan$p.json 2> gpurun_out/r02_prefill_plan$p.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02_prefill_plan$p.json").read().strip().splitlines()[-1])
print("plan_prompt=$p", d["value"], d["unit"], d["ms_per_step"], "ms/step", json.dumps(d["roofline"].get("class_ms_per_step", d["config"].get("class_ms_per_step"))), d["roofline"]["frac"])
PY
done
