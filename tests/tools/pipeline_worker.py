"""One rank of a layer-split greedy decode with REAL stages (llm_amd.pipeline.GpuStage): launched by
tests/test_pipeline_2proc_gpu.py through torch.distributed.run.  All ranks share GPU 0 when the box has fewer GPUs than
ranks (residual over gloo host copies: RCCL refuses two ranks on one device); with a GPU per rank the hop runs through
RCCL inside the library, as in bench.py --gpus N.  The last rank writes the produced token ids as JSON to argv[1]."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))


def main():
    out_path, wtype, n_prompt, n_decode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    n_dev = torch.cuda.device_count()
    os.environ["GGML_HIP_DEVICE"] = str(local_rank % max(n_dev, 1))
    dist.init_process_group("gloo")
    from llm_amd import ggml, pipeline, synth
    hp0 = dict(n_vocab=256, n_embd=128, n_head=4, n_head_kv=4, n_layer=6, n_rot=32, n_ff=352, n_mult=32)
    lb, le = pipeline.layer_range(hp0["n_layer"], rank, world)
    hp, w = synth.make_llama(hp0, wtype, seed=23)  # every rank makes the whole model and keeps its layers
    stage = pipeline.GpuStage(hp, w, (lb, le), 128, n_batch=8)
    use_rccl = n_dev >= world and os.environ.get("LLM_PIPELINE_BACKEND", "rccl") == "rccl"
    comm_ranks = 0
    if use_rccl:
        import ctypes
        idb = torch.zeros(ggml.COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            buf = (ctypes.c_ubyte * ggml.COMM_ID_BYTES)()
            ggml.lib().ggml_hip_comm_unique_id(buf)
            idb = torch.tensor(list(buf), dtype=torch.uint8)
        dist.broadcast(idb, src=0)
        raw = (ctypes.c_ubyte * ggml.COMM_ID_BYTES)(*idb.tolist())
        comm_ranks = ggml.lib().ggml_hip_comm_init(rank, world, raw)
        ok = torch.tensor([1 if comm_ranks == world else 0], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # all ranks take the same transport
        stage.comm_ready = int(ok.item()) == 1
        if not stage.comm_ready:
            if comm_ranks > 0:
                ggml.lib().ggml_hip_comm_destroy()
            use_rccl, comm_ranks = False, 0
    n_seq = world
    for s in range(n_seq):
        stage.new_sequence(s)
    pipe = pipeline.Pipeline(stage, dist, rank, world, None)
    rng = np.random.default_rng(77)
    prompts = [rng.integers(0, hp["n_vocab"], n_prompt).astype(np.int32) for _ in range(n_seq)]
    items = [[p[i:i + 8] for i in range(0, len(p), 8)] + [None] * n_decode for p in prompts]
    produced = pipe.run(items)
    ranges = [None] * world
    dist.all_gather_object(ranges, (lb, le, int(os.environ["GGML_HIP_DEVICE"])))
    if rank == world - 1:
        with open(out_path, "w") as f:
            json.dump({"tokens": [[int(t) for t in produced[s]] for s in range(n_seq)], "layer_ranges": ranges,
                       "comm_ranks_seen_by_rccl": comm_ranks, "hop": "rccl" if stage.comm_ready else "gloo host copies"}, f)
    stage.free()
    if use_rccl:
        ggml.lib().ggml_hip_comm_destroy()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
