cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_llama_gpu.py -x -q -m gpu -k "prefill or attention" 2>&1 | tail -2
for x in 0 1; do
echo "XCDN=$x"; GGML_HIP_MMQ_XCDN=$x timeout 600 python bench.py --mode prefill --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['roofline']['class_ms_per_step'])"
done
