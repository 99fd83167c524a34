# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_prompt_plan_gpu.py -q -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -15
for w in 1 0; do
GGML_HIP_MMQ_W16=$w timeout 300 python bench.py --mode prefill --weights blocks --no-cpu-baseline > gpurun_out/r02_prefill_w16_$w.json 2> gpurun_out/r02_prefill_w16_$w.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02_prefill_w16_$w.json").read().strip().splitlines()[-1])
print("w16=$w:", d["value"], d["unit"], d["ms_per_step"], "ms/step", json.dumps(d["roofline"].get("class_ms_per_step", d["config"].get("class_ms_per_step"))), d["roofline"]["frac"])
PY
tail -2 gpurun_out/r02_prefill_w16_$w.err
done
