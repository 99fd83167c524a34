#!/bin/bash
# round-5 run 3: whole GPU suite again; concurrent sessions with and without the fused attention launches, up to 8 sessions
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -25 > gpurun_out/r5/pytest_run3.txt
tail -12 gpurun_out/r5/pytest_run3.txt
GGML_HIP_FUSE_ATTN=0 timeout 300 python bench.py --mode sessions --sessions 1,2,3,4,6,8 --weights blocks --steps 128 > gpurun_out/r5/sessions_shared_unfused.json 2> gpurun_out/r5/sessions_shared_unfused.err
tail -2 gpurun_out/r5/sessions_shared_unfused.err | cut -c1-300; python -c "
import json
d=json.loads(open('gpurun_out/r5/sessions_shared_unfused.json').read().strip().splitlines()[-1])
for r in d['runs']: print(r['sessions'], r['aggregate_tokens_per_s'], r['vs_one_session'], max(r['per_session_ms_per_token']))
"
timeout 300 python bench.py --mode sessions --sessions 3,6,8 --weights blocks --steps 128 > gpurun_out/r5/sessions_shared_b.json 2> gpurun_out/r5/sessions_shared_b.err
tail -2 gpurun_out/r5/sessions_shared_b.err | cut -c1-300; python -c "
import json
d=json.loads(open('gpurun_out/r5/sessions_shared_b.json').read().strip().splitlines()[-1])
for r in d['runs']: print(r['sessions'], r['aggregate_tokens_per_s'], r['vs_one_session'], max(r['per_session_ms_per_token']))
"
