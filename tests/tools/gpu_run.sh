# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -15 > gpurun_out/r02_pytest_c.txt
tail -6 gpurun_out/r02_pytest_c.txt
timeout 300 python bench.py --mode prefill --weights blocks --no-cpu-baseline > gpurun_out/r02_prefill_c.json 2> gpurun_out/r02_prefill_c.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r02_prefill_c.json").read().strip().splitlines()[-1])
print("prefill:", d["value"], d["unit"], d["ms_per_step"], "ms/step", json.dumps(d["roofline"].get("class_ms_per_step", d["config"].get("class_ms_per_step"))), d["roofline"]["frac"])
PY
