cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export LLM_PIPELINE_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 16 --warmup 4 --model 7b > gpurun_out/bench2.json 2> gpurun_out/bench2.err; echo "rc=$?"
cat gpurun_out/bench2.json; tail -5 gpurun_out/bench2.err
