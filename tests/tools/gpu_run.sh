cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 16 --warmup 4 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err; echo "rc=$?"; tail -3 gpurun_out/bench_2rank.err | cut -c1-300; cut -c1-700 gpurun_out/bench_2rank.json
