# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python bench.py --model 13b --wtype q5_1 --weights blocks --no-cpu-baseline > gpurun_out/r02_bench_13b_q5_1.json 2> gpurun_out/err13.txt
timeout 600 python bench.py --model 65b --wtype q8_0 --weights blocks --no-cpu-baseline --steps 32 --prefill-steps 2 > gpurun_out/r02_bench_65b_q8_0.json 2> gpurun_out/err65.txt
timeout 300 python bench.py --model 7b --wtype q8_0 --weights blocks --no-cpu-baseline > gpurun_out/r02_bench_7b_q8_0.json 2> gpurun_out/err7.txt
python - <<PY
import json
for f in ("r02_bench_13b_q5_1","r02_bench_65b_q8_0","r02_bench_7b_q8_0"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        pf=d["config"].get("prefill") or {}
        print(f, d["value"], "tok/s", d["ms_per_step"], "ms; w1|w3 frac", d["roofline"]["frac"], "all-matvec frac", d["roofline"]["all_matvecs_per_token"]["frac"], "| prefill", pf.get("tokens_per_s"), pf.get("ms_per_step"), (pf.get("roofline") or {}).get("frac"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 gpurun_out/err65.txt
