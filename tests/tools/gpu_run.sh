# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_prompt_plan_gpu.py -q -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -5
cd /tmp
rm -rf /tmp/prof_p
timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_p -o p -- python $R/bench.py --mode prefill --weights blocks --no-cpu-baseline > $R/gpurun_out/r02_prefill_rocprof_line.json 2> $R/gpurun_out/r02_prefill_rocprof.err
cd $R
python - <<PY
import json
d=json.loads(open("gpurun_out/r02_prefill_rocprof_line.json").read().strip().splitlines()[-1])
print("under rocprof:", d["value"], d["unit"], d["ms_per_step"], "ms/step", json.dumps(d["roofline"].get("class_ms_per_step", d["config"].get("class_ms_per_step"))), d["roofline"]["frac"])
PY
python tests/tools/kstats.py /tmp/prof_p > gpurun_out/r02_prefill_kstats.txt 2>&1
head -10 gpurun_out/r02_prefill_kstats.txt
