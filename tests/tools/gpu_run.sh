# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -x -m gpu 2>&1 | tail -15 > gpurun_out/r02_pytest.txt; cat gpurun_out/r02_pytest.txt
echo skip bench
python - <<"PY" || true
import json
try:
    d=json.load(open("gpurun_out/r02_bench1.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["all_matvecs_per_token"], d["config"]["prefill"], d["cpu_baseline"], d["config"]["prep"])
except Exception as e: print("no bench json", e)
PY
