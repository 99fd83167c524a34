"""wq|wk|wv and the decode attention as ONE launch (k_qkv_attn, llm_amd/csrc/kernels/decode_fused.h; the nodes of
crates/models/llama/src/lib.rs:191-307 for one token) against the two-launch pair it replaces.

The fused launch performs the pair's operations in the pair's order — Q reaches the attention as the f16 the pair rounds it
to, this token's K / V rows as the very halves the cache receives — so logits, K/V cache and greedy ids must be BIT-IDENTICAL
to option fuse_attn = 0, in every mode the plan runs in (hipGraph replay, eager, the device-sampled chain, after a rewind to a
position whose granules still sit in memory under an older epoch), for MHA and GQA shapes and head sizes 32 / 64 / 128.  The
hand-off counters must show that the fused launch really ran and that no attention workgroup ever gave up waiting."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = {
    "tiny": dict(n_vocab=256, n_embd=128, n_head=4, n_head_kv=4, n_layer=2, n_rot=32, n_ff=352, n_mult=32),
    "gqa": dict(n_vocab=256, n_embd=256, n_head=8, n_head_kv=2, n_layer=2, n_rot=32, n_ff=352, n_mult=32),
    "d64": dict(n_vocab=256, n_embd=512, n_head=8, n_head_kv=8, n_layer=3, n_rot=64, n_ff=704, n_mult=32),
    "d128": dict(n_vocab=512, n_embd=1024, n_head=8, n_head_kv=8, n_layer=2, n_rot=128, n_ff=1408, n_mult=32),
    "d128gqa": dict(n_vocab=512, n_embd=1024, n_head=8, n_head_kv=4, n_layer=2, n_rot=128, n_ff=1408, n_mult=32),
}


def _stat(G, key):
    return int(G.lib().ggml_hip_get_stat(key.encode()))


def _decode(G, model, toks, n_dec, fuse, rewind=False, device_chain=0):
    G.set_option("fuse_attn", fuse)
    sess = model.start_session(n_batch=8)
    f0, p0 = _stat(G, "fused_attn_tokens"), _stat(G, "plan_tokens")
    sess.feed_prompt(toks)
    out = []
    for _ in range(n_dec):
        tok = sess.infer_next_token()
        out.append((tok, sess.last_logits()))
    if rewind:  # back two positions, decode again: the same positions under a new epoch
        assert sess.rewind(2) == 0
        for _ in range(3):
            tok = sess.infer_next_token()
            out.append((tok, sess.last_logits()))
    ids = None
    if device_chain:
        ids = sess.infer_tokens_device(device_chain)
        out.append((int(ids[-1]), sess.last_logits()))
    k, v = sess.get_kv()
    n_fused = _stat(G, "fused_attn_tokens") - f0
    n_plan = _stat(G, "plan_tokens") - p0
    assert _stat(G, "fused_attn_timeouts") == 0
    sess.free()
    return out, k, v, n_fused, n_plan, ids


@pytest.mark.parametrize("shape", list(SHAPES))
@pytest.mark.parametrize("wtype", [2, 3, 6, 7, 8])
def test_fused_launch_equals_the_two_launch_pair(G, wtype, shape):
    from llm_amd import llama, synth
    hp, w = synth.make_llama(SHAPES[shape], wtype, seed=31)
    model = llama.Llama(hp, w, context_size=96)
    toks = np.random.default_rng(8).integers(0, hp["n_vocab"], 13).astype(np.int32)
    try:
        ref, k0, v0, nf0, np0, _ = _decode(G, model, toks, 9, 0, rewind=True)
        got, k1, v1, nf1, np1, _ = _decode(G, model, toks, 9, 2, rewind=True)
    finally:
        G.set_option("fuse_attn", 1)
        model.free()
    assert nf0 == 0 and nf1 == 12 and np0 == np1  # 9 + 3 decode tokens took the fused launch, none without the option
    for (ta, la), (tb, lb) in zip(ref, got):
        assert ta == tb and np.array_equal(la, lb)
    assert np.array_equal(k0, k1) and np.array_equal(v0, v1)


@pytest.mark.parametrize("graph", [0, 1])
def test_fused_launch_eager_graph_and_device_chain(G, graph):
    """eager launches (option graph = 0: what rocprofv3 traces), hipGraph replay, and the device-sampled greedy chain (the
    epoch advances on the device there, not on the host): the same tokens and logits as the two-launch pair."""
    from llm_amd import llama, synth
    hp, w = synth.make_llama(SHAPES["d64"], 2, seed=5)
    model = llama.Llama(hp, w, context_size=96)
    toks = np.random.default_rng(3).integers(0, hp["n_vocab"], 10).astype(np.int32)
    try:
        G.set_option("graph", graph)
        ref, k0, v0, _, _, ids0 = _decode(G, model, toks, 3, 0, device_chain=9)
        got, k1, v1, nf, _, ids1 = _decode(G, model, toks, 3, 2, device_chain=9)
    finally:
        G.set_option("graph", 1)
        G.set_option("fuse_attn", 1)
        model.free()
    assert nf >= 3 + 9
    assert np.array_equal(ids0, ids1)
    for (ta, la), (tb, lb) in zip(ref, got):
        assert ta == tb and np.array_equal(la, lb)
    assert np.array_equal(k0, k1) and np.array_equal(v0, v1)


def test_fused_launch_is_the_default_at_7b_width_and_matches_the_pair(G, O):
    """The shape the option exists for: 4096-wide layers with 32 heads on 256 CUs (224 mat-vec workgroups x 14 waves deal the
    6144 row pairs as evenly as 256 x 12).  Default option value (1) must pick the fused launch there, up to the split-attention
    threshold, and leave the logits bit-identical to the pair; beyond 255 positions the attention workgroups fetch the second
    256-position pass inside the launch."""
    from llm_amd import llama, synth
    hp0 = dict(synth.LLAMA_7B)
    hp0["n_layer"], hp0["n_vocab"] = 2, 512
    hp, w = synth.make_llama_gaussian(hp0, 2)
    model = llama.Llama(hp, w, context_size=1024)
    toks = np.random.default_rng(1).integers(0, hp["n_vocab"], 300).astype(np.int32)
    res = {}
    try:
        for fuse in (0, 1):
            G.set_option("fuse_attn", fuse)
            sess = model.start_session(n_batch=64)
            f0 = _stat(G, "fused_attn_tokens")
            outs = []
            sess.feed_prompt(toks[:40])
            outs += [(sess.infer_next_token(), sess.last_logits()) for _ in range(4)]       # short context
            sess.feed_prompt(toks[40:292])                                                   # -> n_past 296
            outs += [(sess.infer_next_token(), sess.last_logits()) for _ in range(4)]       # second 256-position pass
            res[fuse] = (outs, sess.get_kv(), _stat(G, "fused_attn_tokens") - f0)
            assert _stat(G, "fused_attn_timeouts") == 0
            sess.free()
    finally:
        G.set_option("fuse_attn", 1)
        model.free()
    assert res[0][2] == 0 and res[1][2] == 8
    for (ta, la), (tb, lb) in zip(res[0][0], res[1][0]):
        assert ta == tb and np.array_equal(la, lb)
    assert np.array_equal(res[0][1][0], res[1][1][0]) and np.array_equal(res[0][1][1], res[1][1][1])


def test_chunk_plan_norm_launches_that_warm_the_next_launch_leave_every_bit_alone(G):
    """k_rmsnorm_quant_warm (kernels/decode.h ColsWarm): in a 2..8-token chunk the norm launches carry one more workgroup for every
    other CU, which pulls the leading row groups of the next k_mmq_cols launch into the L2 that will read them — loads into a junk
    LDS word, nothing else.  At the width it is built for (one k_mmq_cols workgroup per CU: 7B-wide rows) the logits of every chunk
    and the K/V cache with warm_mb = 0 and with the default must be the same bits, and the counter must show that the warming
    launches ran exactly when asked for."""
    from llm_amd import llama, synth
    hp0 = dict(synth.LLAMA_7B)
    hp0["n_layer"], hp0["n_vocab"] = 2, 4096
    hp, w = synth.make_llama_gaussian(hp0, 2)
    model = llama.Llama(hp, w, context_size=256)
    toks = np.random.default_rng(9).integers(0, hp["n_vocab"], 45).astype(np.int32)
    chunks = [toks[0:8], toks[8:16], toks[16:21], toks[21:23], toks[23:31], toks[31:39], toks[39:45]]
    res = {}
    try:
        for warm in (0, 24):
            G.set_option("warm_mb", warm)
            sess = model.start_session(n_batch=8)
            c0 = _stat(G, "cols_warm_launches")
            outs = [sess.evaluate(c) for c in chunks]
            res[warm] = (outs, sess.get_kv(), _stat(G, "cols_warm_launches") - c0)
            sess.free()
    finally:
        G.set_option("warm_mb", 24)
        model.free()
    assert res[0][2] == 0 and res[24][2] > 0, (res[0][2], res[24][2])
    for a, b in zip(res[0][0], res[24][0]):
        assert np.isfinite(a).all() and np.array_equal(a, b)
    assert np.array_equal(res[0][1][0], res[24][1][0]) and np.array_equal(res[0][1][1], res[24][1][1])


def test_wo_tail_warm_up_and_affine_dealing_leave_every_bit_alone(G):
    """The WO form (option fuse_wo: wo + residual as the mat-vec workgroups' second phase, kernels/decode_fused.h wo_tail), the L2
    warm-up of the next launch's w1|w3 rows from its idle window (option warm_mb, NextWarm) and the XCD-affine dealing of
    wq|wk|wv with its hand-off through the XCD's own L2 (option affine, BigArgs::aff_hpl) at the width they are built for:
    logits, ids and K/V with each of them off and with the defaults must be the same bits, and the counters must show that
    every form ran exactly when asked for."""
    from llm_amd import llama, synth
    hp0 = dict(synth.LLAMA_7B)
    hp0["n_layer"], hp0["n_vocab"] = 3, 512
    hp, w = synth.make_llama_gaussian(hp0, 2)
    model = llama.Llama(hp, w, context_size=512)
    toks = np.random.default_rng(4).integers(0, hp["n_vocab"], 150).astype(np.int32)
    res = {}
    try:
        for wo, warm, aff in ((0, 0, 0), (0, 0, 1), (1, 0, 0), (1, 24, 1), (1, 64, 1), (1, 24, 0)):
            G.set_option("fuse_wo", wo)
            G.set_option("warm_mb", warm)
            G.set_option("affine", aff)
            sess = model.start_session(n_batch=64)
            w0, a0 = _stat(G, "fused_wo_tokens"), _stat(G, "fused_affine_tokens")
            sess.feed_prompt(toks)
            outs = [(sess.infer_next_token(), sess.last_logits()) for _ in range(6)]
            res[(wo, warm, aff)] = (outs, sess.get_kv(), _stat(G, "fused_wo_tokens") - w0, _stat(G, "fused_affine_tokens") - a0)
            assert _stat(G, "fused_attn_timeouts") == 0
            sess.free()
    finally:
        G.set_option("fuse_wo", 1)
        G.set_option("warm_mb", 24)
        G.set_option("affine", 1)
        model.free()
    for (wo, warm, aff), got in res.items():
        assert got[2] == (6 if wo else 0) and got[3] == (6 if aff else 0), (wo, warm, aff, got[2], got[3])
    base = res[(0, 0, 0)]
    for key, got in res.items():
        for (ta, la), (tb, lb) in zip(base[0], got[0]):
            assert ta == tb and np.array_equal(la, lb), key
        assert np.array_equal(base[1][0], got[1][0]) and np.array_equal(base[1][1], got[1][1]), key


LONG_SHAPES = {
    "tiny": (SHAPES["tiny"], 1024),
    "gqa": (SHAPES["gqa"], 1024),
    "d128gqa": (SHAPES["d128gqa"], 2048),
}


@pytest.mark.parametrize("shape", ["tiny", "gqa", "d128gqa"])
@pytest.mark.parametrize("wtype", [2, 7])
def test_split_attention_as_one_launch_is_bit_identical_to_its_three_launches(G, shape, wtype):
    """Long contexts (from 512 positions on): scores / softmax + V.P / combine of the position-split decode attention as ONE
    launch (k_attn_split_one, kernels/decode_attn_split.h: scores and partial outputs handed over as tagged granules between the
    workgroups of a head, the last workgroup to arrive combines) against the three launches it replaces (option attn_one = 0):
    same float operations in the same order, so logits and K/V must be BIT-IDENTICAL — at context lengths on and off the 64-position
    range boundaries, right below the context size, and through a hipGraph replay, an eager run and the device-sampled chain."""
    from llm_amd import llama, synth
    hp0, ctx = LONG_SHAPES[shape]
    hp, w = synth.make_llama(hp0, wtype, seed=21)
    model = llama.Llama(hp, w, context_size=ctx)
    rng = np.random.default_rng([wtype, len(shape)])
    toks = rng.integers(0, hp["n_vocab"], ctx).astype(np.int32)
    starts = [513, 577, 700, ctx - 70, ctx - 9]  # decode 8 tokens from each (the last run ends on the final position)

    def run(one, graph=1):
        G.set_option("attn_one", one)
        G.set_option("graph", graph)
        G.set_option("attn_split", 512)  # the split path from 512 positions on (default: 768), so that the range edges below are met
        outs = []
        try:
            s = model.start_session(n_batch=512)
            pos = 0
            for st in starts:
                s.feed_prompt(toks[pos:st])
                b0 = _stat(G, "attn_split_tokens")
                for i in range(8):
                    outs.append(s.evaluate(toks[st + i:st + i + 1])[-1].copy())
                assert _stat(G, "attn_split_tokens") - b0 == 8
                pos = st + 8
            k, v = s.get_kv()
            s.free()
        finally:
            G.set_option("attn_one", 1)
            G.set_option("graph", 1)
            G.set_option("attn_split", 1)
        return outs, k, v

    a, ka, va = run(1)
    b, kb, vb = run(0)
    c, kc, vc = run(1, graph=0)
    assert _stat(G, "fused_attn_timeouts") == 0
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    assert np.array_equal(ka, kb) and np.array_equal(va, vb) and np.array_equal(ka, kc) and np.array_equal(va, vc)
    # the device-sampled greedy chain crosses the same launches
    ids = {}
    G.set_option("attn_split", 512)
    for one in (1, 0):
        G.set_option("attn_one", one)
        s = model.start_session(n_batch=512)
        s.feed_prompt(toks[:600])
        ids[one] = s.infer_tokens_device(12)
        s.free()
    G.set_option("attn_one", 1)
    G.set_option("attn_split", 1)
    assert list(ids[1]) == list(ids[0])
    model.free()


WIDE_SHAPES = {  # wide enough for the mat-vec to keep a quarter of the chip next to 2 / 4 attention workgroups per head
    "mha": dict(n_vocab=256, n_embd=2048, n_head=16, n_head_kv=16, n_layer=2, n_rot=128, n_ff=512, n_mult=32),
    "gqa": dict(n_vocab=256, n_embd=2048, n_head=16, n_head_kv=4, n_layer=2, n_rot=128, n_ff=512, n_mult=32),
}


@pytest.mark.parametrize("shape", ["mha", "gqa"])
def test_several_attention_workgroups_per_head_inside_the_qkv_launch(G, O, shape):
    """Beyond k_qkv_attn's 512-position register window the launch takes 2 (up to 1024 positions) or 4 (up to 2048) attention
    workgroups per head (one per 512 positions), which hand range maxima, range sums and partial outputs to each other
    (attn_consumer_split, kernels/decode_fused.h; option fuse_heads).  Against the plan with the separate split attention
    (fuse_heads = 0: same rounding points, another f32 association of V.P) and against the oracle on the session's own K/V, at
    context lengths around the range edges (512, 1024, 1536) and right below the context size; graph replay and eager."""
    from llm_amd import llama, synth
    hp, w = synth.make_llama(WIDE_SHAPES[shape], 2, seed=33)
    ctx = 2048
    model = llama.Llama(hp, w, context_size=ctx)
    toks = np.random.default_rng(len(shape)).integers(0, hp["n_vocab"], ctx).astype(np.int32)
    starts = [580, 1020, 1532, ctx - 6]  # 5 tokens each: 2 workgroups per head; 2 -> 3 (T crosses 1024); 3 -> 4 (T crosses 1536); the last positions (4)

    def run(heads, graph=1):
        G.set_option("fuse_heads", heads)
        G.set_option("graph", graph)
        G.set_option("attn_split", 1 if heads else 512)  # the comparison path: the separate split attention from 512 positions on
        outs, kinds = [], []
        try:
            s = model.start_session(n_batch=512)
            pos = 0
            for st in starts:
                s.feed_prompt(toks[pos:st])
                f0, p0 = _stat(G, "fused_heads_tokens"), _stat(G, "attn_split_tokens")
                for i in range(5):
                    outs.append(s.evaluate(toks[st + i:st + i + 1])[-1].copy())
                kinds.append((_stat(G, "fused_heads_tokens") - f0, _stat(G, "attn_split_tokens") - p0))
                pos = st + 5
            k, v = s.get_kv()
            s.free()
        finally:
            G.set_option("fuse_heads", 1)
            G.set_option("graph", 1)
            G.set_option("attn_split", 1)
        return outs, kinds, k, v

    a, kinds_a, ka, va = run(1)
    b, kinds_b, kb, vb = run(0)
    c, kinds_c, kc, vc = run(1, graph=0)
    assert kinds_a == [(5, 0)] * 4 and kinds_b == [(0, 5)] * 4, (kinds_a, kinds_b)
    assert _stat(G, "fused_attn_timeouts") == 0
    for x, z in zip(a, c):
        assert np.array_equal(x, z)  # eager == graph replay
    worst = max(float(np.max(np.abs(x - y))) / float(y.std()) for x, y in zip(a, b))
    print(f"{shape}: 2 / 4 attention workgroups per head vs the separate split attention, worst |dlogit|/std = {worst:.2e}")
    # the two paths hand wo the same head outputs up to the f32 association of V.P; after the Q8 re-quantization that is either
    # NOTHING (most tokens: 0.0) or one int8 flip of a 2048-wide row (1e-2 ... 4.4e-2 here, tests/tools/heads_debug.py) that the
    # session then carries in its K/V
    assert worst <= 8e-2
    # the oracle on the device's own K/V (teacher-forced), at the first token of every start — for both paths: 2048-wide rows put
    # more activations next to an int8 rounding edge than the 128-wide test models (one flip: ~3e-2 here), so the yardstick for
    # the new path is what the established one (separate split attention) does on the same inputs
    orc = O.Llama(hp, w, ctx)
    worst_o = {}
    for heads in (1, 0):
        G.set_option("fuse_heads", heads)
        G.set_option("attn_split", 1 if heads else 512)
        try:
            s = model.start_session(n_batch=512)
            pos, wo = 0, 0.0
            for st in starts:
                s.feed_prompt(toks[pos:st])
                k, v = s.get_kv()
                orc.memory_k[:] = k[:orc.memory_k.size]
                orc.memory_v[:] = v[:orc.memory_v.size]
                orc.n_past = st
                got = s.evaluate(toks[st:st + 1])[-1]
                ref = orc.evaluate(toks[st:st + 1], mode=O.ref_mode())[-1]
                wo = max(wo, float(np.max(np.abs(got - ref))) / float(ref.std()))
                pos = st + 1
            s.free()
        finally:
            G.set_option("fuse_heads", 1)
            G.set_option("attn_split", 1)
        worst_o[heads] = wo
    print(f"{shape}: vs oracle worst |dlogit|/std = {worst_o[1]:.2e} (separate split attention: {worst_o[0]:.2e})")
    assert worst_o[1] <= max(4e-2, 2.0 * worst_o[0]) and worst_o[1] <= 1e-1
    model.free()


def test_hand_off_tags_stay_unique_beyond_64_layers(G):
    """The granules of the attention hand-offs carry tag = f(token epoch, layer) and share their buffers between layers; a model
    with more than 64 layers (LLaMA-65B: 80) must not see an earlier layer's granules as current.  66 layers, 2048 wide:
    (a) 2 attention workgroups per head inside the wq|wk|wv launch against the separate split attention, (b) k_attn_split_one
    against its three launches (bit-identical), at ~600 positions."""
    from llm_amd import llama, synth
    hp0 = dict(n_vocab=64, n_embd=2048, n_head=16, n_head_kv=16, n_layer=66, n_rot=128, n_ff=256, n_mult=32)
    hp, w = synth.make_llama(hp0, 2, seed=44)
    model = llama.Llama(hp, w, context_size=1024)
    toks = np.random.default_rng(1).integers(0, hp["n_vocab"], 640).astype(np.int32)

    def run(heads, split, one):
        G.set_option("fuse_heads", heads)
        G.set_option("attn_split", split)
        G.set_option("attn_one", one)
        try:
            s = model.start_session(n_batch=512)
            s.feed_prompt(toks[:600])
            outs = [s.evaluate(toks[600 + i:601 + i])[-1].copy() for i in range(4)]
            s.free()
        finally:
            G.set_option("fuse_heads", 1)
            G.set_option("attn_split", 1)
            G.set_option("attn_one", 1)
        return outs

    fused = run(1, 1, 1)
    one = run(0, 512, 1)
    three = run(0, 512, 0)
    assert _stat(G, "fused_attn_timeouts") == 0
    for x, y in zip(one, three):
        assert np.array_equal(x, y)
    worst = max(float(np.max(np.abs(x - y))) / float(y.std()) for x, y in zip(fused, one))
    print(f"66 layers: fused heads vs split attention worst |dlogit|/std = {worst:.2e}")
    assert worst <= 1e-1  # Q8 flips through 66 layers; stale granules would give O(1)
    model.free()
