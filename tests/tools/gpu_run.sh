cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/bench_decode.json 2> gpurun_out/bench_decode.err; echo "bench rc=$?"; cut -c1-200 gpurun_out/bench_decode.json; tail -2 gpurun_out/bench_decode.err
timeout 600 python bench.py --mode prefill --steps 5 --warmup 2 > gpurun_out/bench_prefill.json 2>/dev/null; cut -c1-200 gpurun_out/bench_prefill.json
rm -rf /tmp/prof_dec
GGML_HIP_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_dec -o bench -- python bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/prof_dec.log 2>&1; echo "prof rc=$?"
grep '"metric"' gpurun_out/prof_dec.log > gpurun_out/prof_dec_bench_line.json
python tests/tools/kstats.py /tmp/prof_dec > gpurun_out/prof_dec_stats.txt; head -12 gpurun_out/prof_dec_stats.txt
