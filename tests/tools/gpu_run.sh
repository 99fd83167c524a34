cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "prefill" 2>&1 | tail -6
timeout 300 python -m pytest tests/test_llama_gpu.py -x -q -m gpu -k "prefill" 2>&1 | tail -3
for d in 2 1; do
echo "DMA=$d"; GGML_HIP_MMQ_DMA=$d timeout 300 python bench.py --mode prefill --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['class_ms_per_step'])"
done
