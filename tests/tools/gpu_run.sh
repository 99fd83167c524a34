cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_llama_gpu.py tests/test_layer_chain_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['host_split_per_token']); print({k:(v['us_in_kernel'],v['us_per_launch_incl_boundary']) for k,v in d['roofline']['per_kind'].items()}, d['roofline']['attn_us_in_kernel'])"
