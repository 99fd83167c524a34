# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tests/tools/decode_repro.py 7b q4_k 2>&1 | tail -12
timeout 300 python tests/tools/decode_repro.py 7b q6_k 2>&1 | tail -6
