#!/bin/bash
# round-6 final measurement set: whole GPU suite + smoke, the end-of-round profile set (bench line, rocprofv3 kernel stats, PMC
# traffic incl. the fused launch, prefill counters, feed trace, timelines), sessions (explicit slots and the unchanged caller), split
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6f; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP version" | tail -30 > gpurun_out/r6f/r06_suite_final.txt
tail -3 gpurun_out/r6f/r06_suite_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tests/tools/profile_round.sh r06 > gpurun_out/r6f/profile_round.log 2>&1
mv gpurun_out/r06_* gpurun_out/r6f/ 2>/dev/null
tail -c 900 gpurun_out/r6f/r06_bench_final.json
timeout 200 python tests/tools/wo_timeline.py 128 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" > gpurun_out/r6f/r06_wo_timeline_128.txt
timeout 200 python tests/tools/timeline.py 7b 256 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | head -40 > gpurun_out/r6f/r06_timeline_256.txt
timeout 300 python bench.py --mode sessions --sessions 1,2,3 --weights blocks --steps 128 > gpurun_out/r6f/r06_sessions.json 2> gpurun_out/r6f/r06_sessions.err
timeout 300 python bench.py --mode sessions --sessions 1,2,3 --sessions-unchanged-caller --weights blocks --steps 128 > gpurun_out/r6f/r06_sessions_unchanged_caller.json 2> gpurun_out/r6f/r06_sessions_unchanged_caller.err
for G in 2 4 8; do timeout 300 python bench.py --mode split --split $G --weights blocks --steps 128 > gpurun_out/r6f/r06_split$G.json 2> gpurun_out/r6f/r06_split$G.err; done
timeout 300 python bench.py --mode feed --weights blocks > gpurun_out/r6f/r06_feed8.json 2> gpurun_out/r6f/r06_feed8.err
python - <<'PY'
import json
for f in ('r06_sessions','r06_sessions_unchanged_caller','r06_split2','r06_split4','r06_split8','r06_feed8'):
    try:
        d=json.loads(open(f'gpurun_out/r6f/{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], [(r['sessions'], r['aggregate_tokens_per_s']) for r in d.get('runs',[])], d.get('overhead_per_hop_us'))
    except Exception as e: print(f, 'failed', e)
PY
