cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_llama_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -2
for v in 1 0; do
GGML_HIP_PREFETCH=$v timeout 600 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prefetch=$v', d['value'], d['ms_per_step'], d['config']['host_split_per_token'])"
done
