#!/bin/bash
# round-3 run 25: green-state check — smoke(), 2-rank bench on one GPU (gloo hop), whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | cut -c1-300
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --model tiny --steps 16 --warmup 2 > gpurun_out/r3/bench_2rank_tiny.json 2> gpurun_out/r3/bench_2rank_tiny.err; cut -c1-900 gpurun_out/r3/bench_2rank_tiny.json; tail -2 gpurun_out/r3/bench_2rank_tiny.err | cut -c1-300
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu > gpurun_out/r3/suite25.txt 2>&1; grep -v "^  File" gpurun_out/r3/suite25.txt | tail -5 | cut -c1-300
