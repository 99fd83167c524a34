cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_fullsize_gpu.py tests/test_entry_gpu.py -q -x 2>&1 | tail -3
timeout 300 python tests/tools/timeline.py 7b > gpurun_out/timeline.txt 2>&1; grep "^avg\|^gap\|token span\|attention" gpurun_out/timeline.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_decode.json 2> gpurun_out/bench_decode.err; echo "bench rc=$?"; cut -c1-200 gpurun_out/bench_decode.json
