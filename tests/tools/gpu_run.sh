set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "prefill or attention" > gpurun_out/t_mmq.log 2>&1; echo "mmq rc=$?" 
tail -5 gpurun_out/t_mmq.log
timeout 600 python -m pytest tests/test_llama_gpu.py -q -m gpu -k "prefill or layer_split" -s > gpurun_out/t_llama.log 2>&1; echo "llama rc=$?"
grep -E "prefill N=|passed|failed|Error|assert" gpurun_out/t_llama.log | head -30
timeout 600 python bench.py --mode prefill --steps 3 --warmup 1 > gpurun_out/prefill.json 2> gpurun_out/prefill.err; echo "prefill rc=$?"
cat gpurun_out/prefill.json; tail -5 gpurun_out/prefill.err
