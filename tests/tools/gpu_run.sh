set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; echo "all rc=$?"
tail -4 gpurun_out/t_all.log
timeout 600 python bench.py > gpurun_out/bench_decode.json 2> gpurun_out/bench_decode.err; echo "bench rc=$?"
cat gpurun_out/bench_decode.json; tail -5 gpurun_out/bench_decode.err
rm -rf gpurun_out/prof_dec
GGML_HIP_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_dec -o bench -- python bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/prof_dec.log 2>&1; echo "prof rc=$?"
grep '"metric"' gpurun_out/prof_dec.log > gpurun_out/prof_dec_bench_line.json
python tests/tools/kstats.py gpurun_out/prof_dec > gpurun_out/prof_dec_stats.txt; head -12 gpurun_out/prof_dec_stats.txt
