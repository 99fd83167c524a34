# scratch script of the last `gpurun` call (see README.md in this directory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 16 --warmup 4 > gpurun_out/r02_bench_2rank_1gpu.json 2> gpurun_out/r02_bench_2rank_1gpu.err
echo rc=$?
tail -c 1200 gpurun_out/r02_bench_2rank_1gpu.json
tail -5 gpurun_out/r02_bench_2rank_1gpu.err
