# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_rate tests/tools/dma_rate.hip 2>&1 | grep -E "error"
timeout 60 /tmp/dma_rate 2>&1 | head -12 > gpurun_out/r02_ibsts.txt; cat gpurun_out/r02_ibsts.txt
