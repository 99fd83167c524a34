"""CPU coverage of the multi-GPU path (world size 2, gloo): the micro-step schedule and the message protocol of
llm_amd/pipeline.py (residual hand-off rank r → r+1, token ring last rank → rank 0, G sequences in flight) with a
deterministic stub stage, checked against a sequential single-process evaluation of the same stub model."""
import os
import socket

import numpy as np
import pytest

E, V = 16, 97


class StubStage:
    """Stage r of a toy 'model': stateful per sequence (a counter standing in for n_past / the KV cache)."""

    def __init__(self, rank, world):
        self.n_embd = E
        self.is_first, self.is_last = rank == 0, rank == world - 1
        self.rank = rank
        self.n_past = {}

    def new_sequence(self, s):
        self.n_past[s] = 0

    def evaluate(self, s, tokens, residual_in):
        n = len(tokens)
        pos = self.n_past[s] + np.arange(n)
        self.n_past[s] += n
        if self.is_first:
            x = np.outer(np.asarray(tokens, np.float32) + 1.0, np.arange(1, E + 1, dtype=np.float32)) / 7.0
        else:
            x = np.asarray(residual_in, np.float32).reshape(n, E)
        x = x * (1.0 + 0.25 * self.rank) + pos[:, None].astype(np.float32) * (0.5 + self.rank)
        if self.is_last:
            return int(np.floor(np.abs(x[-1]).sum())) % V
        return x


def _reference(world, prompts, n_decode):
    out = []
    for s, p in enumerate(prompts):
        stages = [StubStage(r, world) for r in range(world)]
        for st in stages:
            st.new_sequence(0)

        def fwd(toks):
            x = None
            for st in stages:
                x = st.evaluate(0, toks, x)
            return x
        toks_out = []
        tok = None
        for i in range(0, len(p), 4):
            tok = fwd(p[i:i + 4])
            toks_out.append(tok)
        for _ in range(n_decode):
            tok = fwd(np.array([tok], np.int32))
            toks_out.append(tok)
        out.append(toks_out)
    return out


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from llm_amd.pipeline import Pipeline, _continue_decode
    stage = StubStage(rank, world)
    n_seq = world
    for s in range(n_seq):
        stage.new_sequence(s)
    rng = np.random.default_rng(1)
    prompts = [rng.integers(0, V, 8).astype(np.int32) for _ in range(n_seq)]
    pipe = Pipeline(stage, dist, rank, world, device=None)
    first = pipe.run([[p[0:4], p[4:8], None, None] for p in prompts], n_batch=4)
    second = _continue_decode(pipe, n_seq, 3)
    if rank == world - 1:
        q.put(([list(x) for x in first], [list(x) for x in second]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipeline_schedule_and_protocol_gloo(world):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    first, second = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(1)
    prompts = [rng.integers(0, V, 8).astype(np.int32) for _ in range(world)]
    ref = _reference(world, prompts, 2 + 3)
    for s in range(world):
        assert first[s] == ref[s][:4], (s, first[s], ref[s])
        assert second[s] == ref[s][4:7], (s, second[s], ref[s])


def test_schedule_table():
    from llm_amd.pipeline import layer_range, schedule
    tab = schedule(4, 4, 2)
    assert len(tab) == 8 + 3
    assert tab[0] == [(0, 0), None, None, None]
    assert tab[3] == [(3, 0), (2, 0), (1, 0), (0, 0)]
    assert tab[4][0] == (0, 1) and tab[4][3] == (1, 0)
    # every (sequence, item) visits every rank exactly once, one micro-step after the previous rank
    seen = {}
    for t, row in enumerate(tab):
        for r, w in enumerate(row):
            if w is not None:
                seen.setdefault(w, []).append((r, t))
    assert all([r for r, _ in v] == [0, 1, 2, 3] and [t for _, t in v] == list(range(v[0][1], v[0][1] + 4))
               for v in seen.values()) and len(seen) == 8
    assert [layer_range(32, r, 8) for r in (0, 7)] == [(0, 4), (28, 32)]
    assert [layer_range(40, r, 4) for r in range(4)] == [(0, 10), (10, 20), (20, 30), (30, 40)]
