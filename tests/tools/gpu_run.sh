cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_llama_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python tests/tools/timeline.py 7b 2>&1 | tail -10
timeout 600 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
