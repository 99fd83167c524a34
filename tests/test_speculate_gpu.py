"""Option speculate_next (GGML_HIP_SPECULATE_NEXT): behind every single-token plan run the device samples the greedy token
itself and runs the next token's plan at once, so that the reference's UNCHANGED call sequence — sample on the host, build the
graph, ggml_graph_compute (crates/llm-base/src/inference_session.rs:220-295, 381-424) — finds its results already on their way
when the caller did take the first maximum.  Whatever the caller does instead must give exactly what it gives without the
option: another token (a miss: the real evaluation runs behind the speculation), a rewind, a prompt chunk, another session on
the same slot, a snapshot of the K/V in between."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HP = dict(n_vocab=256, n_embd=512, n_head=8, n_head_kv=8, n_layer=3, n_rot=64, n_ff=704, n_mult=32)


def _stat(G, key):
    return int(G.lib().ggml_hip_get_stat(key.encode()))


def _walk(model, toks, script):
    """script: list of ('greedy', n) | ('token', id) | ('rewind', n) | ('chunk', ids); returns every logits row seen + final K/V"""
    s = model.start_session(n_batch=8)
    s.feed_prompt(toks)
    out = []
    for op, arg in script:
        if op == "greedy":
            for _ in range(arg):
                out.append((s.infer_next_token(), s.last_logits()))
        elif op == "token":
            out.append((arg, s.evaluate(np.array([arg], np.int32))[-1].copy()))
        elif op == "rewind":
            assert s.rewind(arg) == 0
        elif op == "chunk":
            out.append((-1, s.evaluate(np.array(arg, np.int32))[-1].copy()))
    n_past = s.n_past
    k, v = s.get_kv()
    s.free()
    return out, k, v, n_past


@pytest.mark.parametrize("speculate_host", [True, False])
def test_speculation_changes_nothing_but_the_waiting(G, speculate_host):
    from llm_amd import llama, synth
    hp, w = synth.make_llama(HP, 2, seed=41)
    model = llama.Llama(hp, w, context_size=128)
    toks = np.random.default_rng(6).integers(0, hp["n_vocab"], 16).astype(np.int32)
    script = [("greedy", 12), ("token", 7), ("greedy", 5), ("token", 9), ("token", 11), ("rewind", 3), ("greedy", 6),
              ("chunk", [3, 1, 4, 1, 5]), ("greedy", 10)]

    def run(spec):
        G.set_option("speculate_next", spec)
        h0, m0 = _stat(G, "spec_hits"), _stat(G, "spec_misses")
        s = model.start_session(n_batch=8)
        s.set_speculate(speculate_host)  # the reference's own sequence (False) and the begin / build-next / end one (True)
        s.free()
        out, k, v, n_past = _walk(model, toks, script)
        return out, k, v, n_past, _stat(G, "spec_hits") - h0, _stat(G, "spec_misses") - m0

    try:
        ref, k0, v0, np0, h_ref, m_ref = run(0)
        got, k1, v1, np1, hits, misses = run(1)
    finally:
        G.set_option("speculate_next", 0)
        model.free()
    assert h_ref == 0 and m_ref == 0
    assert hits >= 20 and misses >= 2  # the greedy stretches hit, the forced tokens miss
    assert np0 == np1
    for (ta, la), (tb, lb) in zip(ref, got):
        assert ta == tb and np.array_equal(la, lb)
    # rows of positions the session has reached are identical; a speculation nobody took may have left its K/V row at the NEXT position
    Eg, C, L = hp["n_embd"], 128, hp["n_layer"]
    for il in range(L):
        ka, kb = k0[il * C * Eg:(il + 1) * C * Eg].reshape(C, Eg), k1[il * C * Eg:(il + 1) * C * Eg].reshape(C, Eg)
        va, vb = v0[il * C * Eg:(il + 1) * C * Eg].reshape(Eg, C), v1[il * C * Eg:(il + 1) * C * Eg].reshape(Eg, C)
        assert np.array_equal(ka[:np0], kb[:np0]) and np.array_equal(va[:, :np0], vb[:, :np0])


def test_two_sessions_alternating_on_one_slot(G):
    """a speculation of session A is simply overtaken when session B (same model, same slot) comes next"""
    from llm_amd import llama, synth
    hp, w = synth.make_llama(HP, 2, seed=43)
    model = llama.Llama(hp, w, context_size=96)
    toks = [np.random.default_rng(s).integers(0, hp["n_vocab"], 9).astype(np.int32) for s in (1, 2)]

    def run(spec):
        G.set_option("speculate_next", spec)
        ss = [model.start_session(n_batch=8) for _ in range(2)]
        for s, t in zip(ss, toks):
            s.feed_prompt(t)
        out = []
        for i in range(14):
            s = ss[i % 2] if i < 10 else ss[0]
            out.append((s.infer_next_token(), s.last_logits()))
        for s in ss:
            s.free()
        return out

    try:
        ref = run(0)
        got = run(1)
    finally:
        G.set_option("speculate_next", 0)
        model.free()
    for (ta, la), (tb, lb) in zip(ref, got):
        assert ta == tb and np.array_equal(la, lb)


def test_speculation_on_the_k_plan(G):
    """K-quant models run the same speculation (k_argmax_next feeds the K plan's replay exactly as it feeds the device-sampled chain)"""
    from llm_amd import ggml, llama, synth
    hp0 = dict(n_vocab=512, n_embd=512, n_head=8, n_head_kv=2, n_layer=3, n_rot=64, n_ff=768, n_mult=32)
    hp, w = synth.make_llama_fast(hp0, ggml.TYPE_Q4_K, seed=5)
    model = llama.Llama(hp, w, context_size=128)
    toks = np.random.default_rng(8).integers(0, hp["n_vocab"], 16).astype(np.int32)
    script = [("greedy", 10), ("token", 7), ("greedy", 6), ("rewind", 2), ("greedy", 5), ("chunk", [3, 1, 4]), ("greedy", 6)]

    def run(spec):
        G.set_option("speculate_next", spec)
        h0, k0 = _stat(G, "spec_hits"), _stat(G, "kplan_tokens")
        s = model.start_session(n_batch=8)
        s.set_speculate(False)
        s.free()
        out, k, v, n_past = _walk(model, toks, script)
        return out, k, v, n_past, _stat(G, "spec_hits") - h0, _stat(G, "kplan_tokens") - k0

    try:
        ref, k0, v0, np0, h_ref, kp_ref = run(0)
        got, k1, v1, np1, hits, kp = run(1)
    finally:
        G.set_option("speculate_next", 0)
        model.free()
    assert h_ref == 0 and hits >= 10 and kp_ref > 0 and kp > 0  # (a miss pauses the guessing for eight tokens)
    assert np0 == np1
    for (ta, la), (tb, lb) in zip(ref, got):
        assert ta == tb and np.array_equal(la, lb)


def test_device_side_readers_see_the_evaluated_token_not_the_speculated_one(G):
    """the speculative run used to write its logits / embedding row into the graph's device mirrors, so a top-k
    prefilter (llm_session_topk) or a node read on the device, queued behind it, returned the NEXT token's values.  It writes
    plan-private alternates now (DecodePlan::logits_alt): after every token — hit or miss — the device-side top-k and the logits
    node read on the device equal the host logits the evaluation returned."""
    from llm_amd import llama, synth
    hp, w = synth.make_llama(HP, 2, seed=43)
    model = llama.Llama(hp, w, context_size=128)
    toks = np.random.default_rng(7).integers(0, hp["n_vocab"], 12).astype(np.int32)
    try:
        G.set_option("speculate_next", 1)
        h0, m0 = _stat(G, "spec_hits"), _stat(G, "spec_misses")
        s = model.start_session(n_batch=8)
        s.set_speculate(False)
        s.feed_prompt(toks)
        forced = {4: 17, 9: 3}  # two misses among the greedy tokens
        for i in range(14):
            if i in forced:
                lg = s.evaluate(np.array([forced[i]], np.int32))[-1].copy()
            else:
                s.infer_next_token()
                lg = s.last_logits()
            vals, ids = s.top_k(5)
            order = np.lexsort((np.arange(lg.size), -lg))[:5]  # best first, lower id first on ties
            assert np.array_equal(ids, order.astype(np.int32)) and np.array_equal(vals, lg[order]), i
            dev = s.read_node(index=s.graph_stats()[0] - 1)  # the logits node (the graph's last), read on the device
            assert np.array_equal(dev[-lg.size:], lg), i
        assert _stat(G, "spec_hits") - h0 >= 3 and _stat(G, "spec_misses") - m0 >= 1  # (a miss stops the guessing for 8 tokens)
        s.free()
    finally:
        G.set_option("speculate_next", 0)
        model.free()
