# scratch script of the last `gpurun` call (see README.md in this directory); the round-end validation was:
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/bench_decode.json 2> gpurun_out/bench_decode.err; echo "bench rc=$?"
timeout 600 python bench.py --mode prefill --steps 5 --warmup 2 > gpurun_out/bench_prefill.json 2>/dev/null
rm -rf /tmp/prof_dec
GGML_HIP_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_dec -o bench -- python bench.py --steps 32 --warmup 4 --no-cpu-baseline > gpurun_out/prof_dec.log 2>&1
python tests/tools/kstats.py /tmp/prof_dec > gpurun_out/prof_dec_stats.txt
