# scratch script of the last `gpurun` call (see README.md in this directory): the round-end validation
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl" | tail -8 > gpurun_out/pytest_gpu.txt
tail -4 gpurun_out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tests/tools/profile_round.sh rNN > gpurun_out/profile_round.log 2>&1
tail -c 600 gpurun_out/rNN_bench_final.json
