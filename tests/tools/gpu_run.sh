# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 -o /tmp/overlap_probe3 tests/tools/overlap_probe3.hip 2>/dev/null
NREG=64 timeout 120 /tmp/overlap_probe3 > gpurun_out/r02_overlap_probe3.txt 2>&1
cat gpurun_out/r02_overlap_probe3.txt
