set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_layer_chain_gpu.py -x -q -m gpu > gpurun_out/t_llama.log 2>&1; echo "llama rc=$?"
tail -5 gpurun_out/t_llama.log
timeout 600 python tests/tools/timeline.py 7b > gpurun_out/timeline.txt 2>&1; echo "rc=$?"
head -1 gpurun_out/timeline.txt; tail -10 gpurun_out/timeline.txt
timeout 600 python bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/dec_big.json 2> gpurun_out/dec_big.err; echo "bench rc=$?"
cat gpurun_out/dec_big.json | cut -c1-330; tail -5 gpurun_out/dec_big.err
