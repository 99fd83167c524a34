# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -8 > gpurun_out/r02_pytest_final.txt
tail -4 gpurun_out/r02_pytest_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tests/tools/profile_round.sh r02 > gpurun_out/r02_profile_round.log 2>&1
tail -c 900 gpurun_out/r02_bench_final.json
head -6 gpurun_out/r02_prefill7b_kernel_stats.txt
