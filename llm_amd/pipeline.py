"""Layer split across the GPUs of one node: one process per GPU, rank r owns the contiguous layers
[r·L/G, (r+1)·L/G) and their K/V memory; the only data that crosses a stage boundary is the residual [n_embd × N]
f32.  On GPUs it goes through RCCL INSIDE the library: ggml_hip_comm_recv into the session's hand-off buffer before
the stage's graph and ggml_hip_comm_send of its output after it, both enqueued on the backend's stream (ordered with
the kernels, no host copy, no host synchronisation for the hop, no torch tensor).  torch.distributed ("gloo") is only
the launcher's rendezvous: it carries the RCCL unique id, the 4-byte sampled token id from the last rank back to rank
0, and the timing barrier.  The CPU tests (gloo, stub stage) and a 1-GPU box (RCCL refuses two ranks on one device)
move the residual through torch.distributed instead.

Batch-1 decode through a layer split is a sequential pipeline: one sequence cannot be faster on G GPUs
than on one.  The driver therefore keeps G independent sequences in flight, one per stage — the reference's
"several InferenceSessions on one Model" use — so every GPU is busy every micro-step:

    work item (sequence s, step j) runs on stage r at micro-step  t = j·G + s + r

Each micro-step a rank (1) posts the receive for this micro-step's input, (2) waits for last micro-step's send to
drain, (3) waits for the input, (4) evaluates its layers, (5) posts the send of its output.  Posting the receive
first is what makes the ring (last rank → rank 0 for the token) deadlock-free.

The stage compute is behind a tiny interface (`Stage`) so that tests/test_pipeline_cpu.py can run the schedule and
the message protocol on CPU (gloo, world sizes 2 and 3) with a stub stage; real stages of a split model run on one GPU in
tests/test_llama_gpu.py (test_layer_split_stages_match_whole_model; the RCCL hop itself in
test_layer_split_hop_through_rccl_inside_the_library, as a 1-rank communicator sending to itself) and
tests/test_prompt_plan_gpu.py (prompt batches through a split).
"""
import json
import os
import time

import numpy as np


class Stage:
    """What a rank contributes.  Residuals are float32 numpy arrays [N, n_embd] at this interface; GpuStage moves
    them to/from the device hand-off buffers of the session (llm_session_stage_buffers)."""
    n_embd = 0
    is_first = True
    is_last = True

    def new_sequence(self, s):
        raise NotImplementedError

    def evaluate(self, s, tokens, residual_in):
        """tokens: int32 [N] (always known for prompt items; for decode items only rank 0 gets the real id).
        Returns the residual [N, n_embd] (non-last stage) or the next token id (last stage)."""
        raise NotImplementedError


class GpuStage(Stage):
    """A real stage: llm_amd.llama.Llama restricted to a layer range, one InferenceSession per in-flight sequence."""

    def __init__(self, hp, weights, layer_range, context_size, n_batch=8):
        from . import ggml, llama
        self.G = ggml
        self.model = llama.Llama(hp, weights, context_size=context_size, layer_range=layer_range)
        self.n_embd = hp["n_embd"]
        self.is_first, self.is_last = self.model.is_first, self.model.is_last
        self.n_batch = n_batch
        self.sessions = {}

    def new_sequence(self, s):
        self.sessions[s] = self.model.start_session(n_batch=self.n_batch)

    def evaluate(self, s, tokens, residual_in):
        sess = self.sessions[s]
        n = len(tokens)
        in_dev, out_dev, _ = sess.stage_buffers()
        if not self.is_first:
            r = np.ascontiguousarray(residual_in, dtype=np.float32)
            self.G.lib().ggml_hip_memcpy(in_dev, r.ctypes.data, r.nbytes, 0)
        logits = sess.evaluate(tokens, want_all_logits=self.is_last)
        if self.is_last:
            return int(np.argmax(logits[-1]))
        out = np.empty((n, self.n_embd), np.float32)
        self.G.lib().ggml_hip_memcpy(out.ctypes.data, out_dev, out.nbytes, 1)
        return out

    # the hop inside the library (RCCL on the backend stream): ggml_hip_comm_init must have been called
    comm_ready = False

    def evaluate_comm(self, s, tokens, prev, nxt):
        sess = self.sessions[s]
        n = len(tokens)
        assert n <= self.n_batch, "the hand-off buffers hold n_batch tokens"
        in_dev, out_dev, _ = sess.stage_buffers()
        nbytes = n * self.n_embd * 4
        L = self.G.lib()
        if not self.is_first:
            L.ggml_hip_comm_recv(in_dev, nbytes, prev)  # enqueued: the stage's first kernel reads it in stream order
        logits = sess.evaluate(tokens, want_all_logits=self.is_last)
        if self.is_last:
            return int(np.argmax(logits[-1]))
        L.ggml_hip_comm_send(out_dev, nbytes, nxt)  # the session's next evaluate is ordered behind it on the stream
        return None

    def free(self):
        for s in self.sessions.values():
            s.free()
        self.model.free()


def layer_range(n_layer, rank, world):
    return rank * n_layer // world, (rank + 1) * n_layer // world


def schedule(world, n_seq, items_per_seq):
    """Micro-step table: for every micro-step t the (sequence, item index) each rank works on, or None."""
    total = items_per_seq * n_seq
    steps = []
    for t in range(total + world - 1):
        row = []
        for r in range(world):
            u = t - r
            row.append((u % n_seq, u // n_seq) if 0 <= u < total else None)
        steps.append(row)
    return steps


class Pipeline:
    """Runs `items[s] = [tokens_0, tokens_1, ...]` for every sequence s through the stages.  An item whose tokens are
    None is a decode step: its single token is the argmax of the previous item of the same sequence (sent by the
    last rank to rank 0).  Returns, on the last rank, the token produced after every item."""

    def __init__(self, stage, dist, rank, world, device=None):
        self.stage, self.dist, self.rank, self.world, self.device = stage, dist, rank, world, device
        self.torch = __import__("torch")

    def _buf(self, n):
        t = self.torch
        return t.empty(n, dtype=t.float32, device=self.device) if self.device is not None else t.empty(n, dtype=t.float32)

    def run(self, items, n_batch=8, on_timed_region=None):
        t, dist, rank, world, stage = self.torch, self.dist, self.rank, self.world, self.stage
        n_seq = len(items)
        n_items = len(items[0])
        assert all(len(it) == n_items for it in items)
        E = stage.n_embd
        prev, nxt = (rank - 1) % world, (rank + 1) % world
        recv_res = self._buf(n_batch * E)
        send_res = [self._buf(n_batch * E) for _ in range(2)]  # double buffer: a send may still drain while we compute
        recv_tok = t.zeros(1, dtype=t.int32, device=self.device) if self.device is not None else t.zeros(1, dtype=t.int32)
        send_tok = [t.zeros_like(recv_tok) for _ in range(2)]
        pending = None
        produced = [[None] * n_items for _ in range(n_seq)]
        use_lib = bool(getattr(stage, "comm_ready", False))
        assert n_batch <= getattr(stage, "n_batch", n_batch), "stage hand-off buffers are smaller than the pipeline's batch"
        for step, row in enumerate(schedule(world, n_seq, n_items)):
            if on_timed_region is not None:
                on_timed_region(step)
            work = row[rank]
            if work is None:
                if pending is not None:
                    for p in pending:
                        p.wait()
                    pending = None
                continue
            s, j = work
            toks = items[s][j]
            n = 1 if toks is None else len(toks)
            need_tok = rank == 0 and toks is None and world > 1
            reqs = []
            # (1) post this micro-step's receives first (with the library transport the residual's receive is enqueued
            #     on the backend stream by evaluate_comm; only the token id travels through torch.distributed)
            if rank > 0 and not use_lib:
                reqs.append(dist.irecv(recv_res[: n * E], src=prev))
            if need_tok:
                reqs.append(dist.irecv(recv_tok, src=world - 1))
            # (2) last micro-step's sends must have drained before their buffers are reused two steps later
            if pending is not None:
                for p in pending:
                    p.wait()
                pending = None
            # (3) wait for the input
            for q in reqs:
                q.wait()
            if self.device is not None:
                t.cuda.current_stream().synchronize()
            if toks is None:
                if rank == 0:
                    tok = int(recv_tok.item()) if world > 1 else produced[s][j - 1]
                    toks = np.array([tok], np.int32)
                else:
                    toks = np.zeros(1, np.int32)  # later stages only need N
            # (4) evaluate
            sb = send_res[step & 1]
            if use_lib:
                out = stage.evaluate_comm(s, toks, prev, nxt)
            else:
                rin = recv_res[: n * E].cpu().numpy().reshape(n, E) if rank > 0 else None
                out = stage.evaluate(s, toks, rin)
                if not stage.is_last:
                    sb[: n * E].copy_(t.from_numpy(np.ascontiguousarray(out).reshape(-1)))
            # (5) post the sends
            pending = []
            if not stage.is_last:
                if not use_lib:
                    pending.append(dist.isend(sb[: n * E], dst=nxt))
            else:
                produced[s][j] = out
                follow = j + 1 < n_items and items[s][j + 1] is None
                if follow and world > 1:
                    st = send_tok[step & 1]
                    st.fill_(out)
                    pending.append(dist.isend(st, dst=0))
        if pending is not None:
            for p in pending:
                p.wait()
        if stage.is_last:
            self._last_tokens = [produced[s][-1] for s in range(n_seq)]
        return produced


# ---------------------------------------------------------------------------------------------------------------
# bench.py --gpus N entry
# ---------------------------------------------------------------------------------------------------------------
def run_bench(args):
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} must be launched with torchrun --nproc-per-node {args.gpus} "
                         f"(WORLD_SIZE={world})")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    n_dev = torch.cuda.device_count()
    os.environ["GGML_HIP_DEVICE"] = str(local_rank % max(n_dev, 1))  # one process drives one GPU (read at backend init)
    # torch.distributed is the launcher's rendezvous only (unique id, token ids, barrier): gloo on host tensors.  The
    # residual goes through RCCL inside the library when every rank of the node has its own GPU (decided per node so
    # that all ranks agree); RCCL refuses two ranks on one device, so a 1-GPU box falls back to host copies over gloo.
    dist.init_process_group("gloo")
    device = None
    use_rccl = os.environ.get("LLM_PIPELINE_BACKEND", "rccl") == "rccl" and n_dev >= local_world
    from . import ggml, synth
    hp0 = {"7b": synth.LLAMA_7B, "13b": synth.LLAMA_13B, "65b": synth.LLAMA_65B, "tiny": synth.TINY}[args.model]
    wtype = {"q4_0": ggml.TYPE_Q4_0, "q4_1": ggml.TYPE_Q4_1, "q5_0": ggml.TYPE_Q5_0, "q5_1": ggml.TYPE_Q5_1,
             "q8_0": ggml.TYPE_Q8_0}[args.wtype]
    lb, le = layer_range(hp0["n_layer"], rank, world)
    names = synth.stage_tensor_names(hp0, lb, le)
    make = synth.make_llama_gaussian if getattr(args, "weights", "gaussian") == "gaussian" else synth.make_llama_fast
    hp, w = make(hp0, wtype, only=names)
    ctx = 2048 if args.model != "tiny" else 256
    stage = GpuStage(hp, w, (lb, le), ctx, n_batch=8)
    comm_ranks, rccl_failed = 0, False
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
        # every rank must agree on the transport: one failed communicator sends all of them to host copies over gloo
        ok = torch.tensor([1 if comm_ranks == world else 0], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            stage.comm_ready = True
        else:
            if comm_ranks > 0:
                ggml.lib().ggml_hip_comm_destroy()
            comm_ranks, use_rccl, rccl_failed = 0, False, True
    backend = f"rccl inside libggml_hip.so ({comm_ranks} ranks) for the residual; gloo for token ids and the barrier" \
        if use_rccl else ("gloo (host copies: the RCCL communicator could not be formed)" if rccl_failed else
                          "gloo (host copies: fewer GPUs than ranks on this node)")
    n_seq = world
    for s in range(n_seq):
        stage.new_sequence(s)
    pipe = Pipeline(stage, dist, rank, world, device)
    rng = np.random.default_rng(42)
    prompts = [rng.integers(0, hp["n_vocab"], args.prompt).astype(np.int32) for _ in range(n_seq)]
    chunks = [[p[i:i + 8] for i in range(0, len(p), 8)] for p in prompts]
    # untimed: prompt + warmup decode steps
    pipe.run([c + [None] * args.warmup for c in chunks])

    def barrier():
        dist.barrier()
        if device is not None:
            torch.cuda.synchronize()
        ggml.lib().ggml_hip_synchronize()

    barrier()
    t0 = time.perf_counter()
    # the timed decode continues the same sequences: items are pure decode steps; the first token of each
    # sequence comes from the last warmup item, which the last rank re-sends as item 0's input
    produced = _continue_decode(pipe, n_seq, args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device) if device is not None else torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    ranges = [None] * world  # what every rank holds: [layer_begin, layer_end, local device]
    dist.all_gather_object(ranges, [lb, le, int(os.environ["GGML_HIP_DEVICE"])])
    if rank == 0:
        total_tokens = n_seq * args.steps
        out = {"metric": f"decode tokens/s LLaMA-{args.model.upper()} {args.wtype.upper()}",
               "value": round(total_tokens / elapsed, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None,
               "dtype": "i8*i4->i32 block dots, f32 accumulate (W4A8 = ggml's Q4_0·Q8_0)", "data": "synthetic",
               "config": {"workload": f"LLaMA-{args.model.upper()} {args.wtype.upper()} greedy decode, layer split over "
                                      f"{world} GPUs ({le - lb} layers/GPU), {n_seq} sequences in flight (one per stage), "
                                      f"{args.prompt}-token prompts, ctx {ctx}, f16 KV",
                          "parallelism": f"pp{world} layer split, residual hop: " + ("RCCL send/recv inside the library" if use_rccl else
                                                                                      "host copies over gloo (RCCL init failed)" if rccl_failed else
                                                                                      "host copies over gloo (fewer GPUs than ranks)"),
                          "layer_ranges_by_rank": ranges,
                          "sequences_in_flight": n_seq,
                          "single_stream_tokens_per_s": round(args.steps / elapsed, 2),
                          "comm_backend": backend, "comm_ranks_seen_by_rccl": comm_ranks},
               "roofline": None, "cpu_baseline": None}
        print(json.dumps(out), flush=True)
    stage.free()
    if use_rccl:
        ggml.lib().ggml_hip_comm_destroy()
    dist.destroy_process_group()


def _continue_decode(pipe, n_seq, steps):
    """Decode `steps` more tokens for every sequence.  The token that starts item 0 is the one the last rank produced
    at the end of the previous run(); it is re-sent to rank 0 through the same ring."""
    t, dist, rank, world = pipe.torch, pipe.dist, pipe.rank, pipe.world
    last = getattr(pipe, "_last_tokens", None)
    # hand the carried-over tokens to rank 0
    carry = t.zeros(n_seq, dtype=t.int32, device=pipe.device) if pipe.device is not None else t.zeros(n_seq, dtype=t.int32)
    if world > 1:
        if rank == world - 1:
            carry.copy_(t.tensor(last, dtype=t.int32))
            dist.send(carry, dst=0)
        elif rank == 0:
            dist.recv(carry, src=world - 1)
        first = [np.array([int(x)], np.int32) for x in carry.cpu().tolist()] if rank == 0 else [np.zeros(1, np.int32)] * n_seq
    else:
        first = [np.array([int(x)], np.int32) for x in last]
    items = [[first[s]] + [None] * (steps - 1) for s in range(n_seq)]
    out = pipe.run(items)
    if rank == world - 1:
        pipe._last_tokens = [out[s][-1] for s in range(n_seq)]
    return out


def selftest():
    """`python -m llm_amd.pipeline --selftest` (under torchrun for more than one rank): first contact with the node's GPUs made
    boring — before anything is timed, every rank prints what it drives, forms the RCCL communicator INSIDE the library exactly
    as the bench does (unique id over gloo, ggml_hip_comm_init), reports `comm_ranks_seen_by_rccl`, and moves a known pattern
    once around the ring with ggml_hip_comm_sendrecv on the backend stream (rank r -> r + 1, 32 KiB = the 65B residual), checking
    every byte.  Exit code 0 only if every rank saw world ranks and its payload arrived intact."""
    import ctypes
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    n_dev = torch.cuda.device_count()
    os.environ["GGML_HIP_DEVICE"] = str(local_rank % max(n_dev, 1))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from . import ggml
    L = ggml.lib()
    info = {"rank": rank, "world": world, "visible_gpus": n_dev, "device": int(os.environ["GGML_HIP_DEVICE"]),
            "library": L.ggml_hip_version().decode()}
    idb = torch.zeros(ggml.COMM_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        buf = (ctypes.c_ubyte * ggml.COMM_ID_BYTES)()
        L.ggml_hip_comm_unique_id(buf)
        idb = torch.tensor(list(buf), dtype=torch.uint8)
    dist.broadcast(idb, src=0)
    raw = (ctypes.c_ubyte * ggml.COMM_ID_BYTES)(*idb.tolist())
    seen = L.ggml_hip_comm_init(rank, world, raw)
    info["comm_ranks_seen_by_rccl"] = int(seen)
    ok = seen == world
    if ok:
        n = 32 * 1024
        src = np.frombuffer(np.random.default_rng(1000 + rank).bytes(n), dtype=np.uint8).copy()
        want = np.frombuffer(np.random.default_rng(1000 + (rank - 1) % world).bytes(n), dtype=np.uint8)
        # device buffers from torch (same device as the library's backend); the hop itself runs on the library's stream
        dev = torch.device("cuda", int(os.environ["GGML_HIP_DEVICE"]))
        t_src = torch.from_numpy(src).to(dev)
        t_dst = torch.zeros(n, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        d_src, d_dst = ctypes.c_void_p(t_src.data_ptr()), ctypes.c_void_p(t_dst.data_ptr())
        t0 = time.perf_counter()
        L.ggml_hip_comm_sendrecv(d_src, (rank + 1) % world, d_dst, (rank - 1) % world, n)
        L.ggml_hip_synchronize()
        info["ring_hop_first_call_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        t0 = time.perf_counter()
        for _ in range(20):
            L.ggml_hip_comm_sendrecv(d_src, (rank + 1) % world, d_dst, (rank - 1) % world, n)
        L.ggml_hip_synchronize()
        info["ring_hop_us"] = round((time.perf_counter() - t0) / 20 * 1e6, 1)
        got = t_dst.cpu().numpy()
        ok = bool(np.array_equal(got, want))
        info["payload_intact"] = ok
        L.ggml_hip_comm_destroy()
    allok = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(allok, op=dist.ReduceOp.MIN)
    infos = [None] * world
    dist.all_gather_object(infos, info)
    if rank == 0:
        print(json.dumps({"selftest": "passed" if int(allok.item()) else "FAILED", "ranks": infos}), flush=True)
    dist.destroy_process_group()
    return 0 if int(allok.item()) else 1


if __name__ == "__main__":
    import sys
    if "--selftest" in sys.argv:
        sys.exit(selftest())
    print("usage: [torchrun --nproc-per-node G] python -m llm_amd.pipeline --selftest", file=sys.stderr)
    sys.exit(2)
