# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_prompt_plan_gpu.py tests/test_ops_gpu.py -q -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -25 > gpurun_out/r02_pytest_prompt.txt
tail -6 gpurun_out/r02_pytest_prompt.txt
for w in 8 4; do
GGML_HIP_MMQ_WAVES=$w timeout 300 python bench.py --mode prefill --weights blocks --no-cpu-baseline > gpurun_out/r02_prefill_w$w.json 2> gpurun_out/r02_prefill_w$w.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02_prefill_w$w.json").read().strip().splitlines()[-1])
print("waves=$w:", d["value"], d["unit"], d["ms_per_step"], "ms/step", json.dumps(d["roofline"].get("class_ms_per_step", d["config"].get("class_ms_per_step"))), d["roofline"]["frac"])
PY
done
