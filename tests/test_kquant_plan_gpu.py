"""The K plan (llama_plan.inc plan_launch_k, kernels/kquant_plan.h): single-token decode of a LLaMA whose matrices are
K-quants (block structs crates/ggml/sys/src/lib.rs:2977-3303, file types crates/llm-base/src/loader.rs:80-93) as 10-13 launches
per layer from a captured hipGraph instead of the node-by-node executor.

* against the oracle (its restatement of k_quants.c in the mode the reference's build runs) on the session's own K/V state,
  at the bound of the other model-level tests (EDGE: one rounding-edge flip of a downstream activation quant);
* against the node-by-node executor (option plan_k = 0), which test_kquant_gpu.py holds to the oracle op by op: every launch
  of the plan repeats the executor's float operations in its order except the attention (k_attn_decode sums a head's scores
  and V.P in a different order than the three generic launches), so the two agree to a few f32 ulps of the logits' scale;
* a mixed-type model (wv / w2 / output Q6_K, the rest Q4_K — the *_K_M recipe): the plan takes each tensor's own type."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KTYPES = [10, 11, 12, 13, 14]  # q2_K q3_K q4_K q5_K q6_K
TINY_K = dict(n_vocab=256, n_embd=256, n_head=4, n_head_kv=4, n_layer=2, n_rot=64, n_ff=512, n_mult=32)
GQA_K = dict(n_vocab=512, n_embd=512, n_head=8, n_head_kv=2, n_layer=3, n_rot=64, n_ff=768, n_mult=32)


def _stat(G, key):
    return int(G.lib().ggml_hip_get_stat(key.encode()))


def _model(O, hp0, wtype, seed, wtypes=None):
    from llm_amd import synth
    rng = np.random.default_rng([wtype, seed])
    hp = dict(hp0)
    w = {}
    for name, (ne0, ne1) in synth.tensor_shapes(hp).items():
        if ne1 is None:
            w[name] = (1.0 + 0.01 * rng.standard_normal(ne0)).astype(np.float32)
        else:
            t = (wtypes or {}).get(name, wtype)
            w[name] = O.quantize(t, (0.02 * rng.standard_normal((ne1, ne0))).astype(np.float32))
    hp["wtype"] = wtype
    if wtypes:
        hp["wtypes"] = dict(wtypes)
    return hp, w


def _decode(G, model, toks, n_prompt, plan_k):
    G.set_option("plan_k", plan_k)
    try:
        sess = model.start_session(n_batch=8)
        sess.evaluate(toks[:n_prompt])
        k0 = _stat(G, "kplan_tokens")
        outs = [sess.evaluate(toks[i:i + 1])[-1].copy() for i in range(n_prompt, len(toks))]
        ran = _stat(G, "kplan_tokens") - k0
        k, v = sess.get_kv()
        sess.free()
    finally:
        G.set_option("plan_k", 1)
    return outs, k, v, ran


@pytest.mark.parametrize("wtype", KTYPES)
@pytest.mark.parametrize("cfg", ["tiny", "gqa"])
def test_k_plan_matches_the_oracle_and_the_executor(G, O, wtype, cfg):
    from llm_amd import llama
    hp0 = {"tiny": TINY_K, "gqa": GQA_K}[cfg]
    hp, w = _model(O, hp0, wtype, 77)
    ctx = 96
    model = llama.Llama(hp, w, context_size=ctx)
    toks = np.random.default_rng([wtype, 5]).integers(0, hp["n_vocab"], 24).astype(np.int32)
    n_prompt = 8
    a, ka, va, ran_a = _decode(G, model, toks, n_prompt, 1)
    b, kb, vb, ran_b = _decode(G, model, toks, n_prompt, 0)
    c, kc, vc, ran_c = _decode(G, model, toks, n_prompt, 2)  # one launch per matrix instead of wq|wk|wv and w1|w3 together
    assert ran_a == len(toks) - n_prompt and ran_b == 0 and ran_c == ran_a  # the plan ran every decode token / none with the option off
    for x, y in zip(a, c):
        assert np.array_equal(x, y)  # a row's sum does not depend on how the rows are dealt
    assert np.array_equal(ka, kc) and np.array_equal(va, vc)
    # the oracle, token by token on the device's own K/V state (teacher-forced: chaos cannot accumulate)
    sess = model.start_session(n_batch=8)
    orc = O.Llama(hp, w, ctx)
    got = sess.evaluate(toks[:n_prompt])
    ref = orc.evaluate(toks[:n_prompt], mode=O.ref_mode())
    worst = 0.0
    for i in range(n_prompt, len(toks)):
        k, v = sess.get_kv()
        orc.memory_k[:] = k[:orc.memory_k.size]
        orc.memory_v[:] = v[:orc.memory_v.size]
        got = sess.evaluate(toks[i:i + 1])[-1]
        ref = orc.evaluate(toks[i:i + 1], mode=O.ref_mode())[-1]
        worst = max(worst, float(np.max(np.abs(got - ref))) / float(ref.std()))
        assert np.array_equal(got, a[i - n_prompt])  # the same session replayed: deterministic
    sess.free()
    print(f"type {wtype} {cfg}: K plan vs oracle worst |dlogit|/std = {worst:.2e}")
    # EDGE of test_llama_gpu.py (one rounding-edge flip of a downstream Q8_K quant in a two-layer model); the three-layer,
    # 512-wide model amplifies a layer-0 flip through two more layers (measured 1.8e-2 … 4.1e-2 where the executor — 5e-7
    # from the plan — lands on the same values): twice that.  Anything structural is >> 1e-1.
    assert worst <= (4e-2 if cfg == "tiny" else 8e-2)
    exec_worst = max(float(np.max(np.abs(x - y))) / float(y.std()) for x, y in zip(a, b))
    print(f"type {wtype} {cfg}: K plan vs node-by-node executor worst |dlogit|/std = {exec_worst:.2e}")
    # same float operations in the same order except inside the attention launch: ~1e-6, or one rounding-edge flip behind it
    assert exec_worst <= (4e-2 if cfg == "tiny" else 8e-2)
    # K/V rows: the prompt rows were written by the same executor launches on both sides
    Eg = hp["n_embd"] // (hp["n_head"] // hp["n_head_kv"])
    assert np.array_equal(ka[:n_prompt * Eg], kb[:n_prompt * Eg])
    # layer 0's rows of the decoded tokens depend on the embedding, the norm and wk / wv only: same launches, same bits
    assert np.array_equal(ka[n_prompt * Eg:len(toks) * Eg], kb[n_prompt * Eg:len(toks) * Eg])
    va0, vb0 = va[:ctx * Eg].reshape(Eg, ctx), vb[:ctx * Eg].reshape(Eg, ctx)
    assert np.array_equal(va0[:, :len(toks)], vb0[:, :len(toks)])
    model.free()


def test_k_plan_on_a_mixed_type_model(G, O):
    """The *_K_M recipe: attention.wv, feed_forward.w2 and output as Q6_K, everything else Q4_K.  The plan reads each
    tensor's own type; checked against the node-by-node executor (the oracle's model takes one type)."""
    from llm_amd import llama
    wt = {"output.weight": 14}
    for i in range(GQA_K["n_layer"]):
        wt[f"layers.{i}.attention.wv.weight"] = 14
        wt[f"layers.{i}.feed_forward.w2.weight"] = 14
    hp, w = _model(O, GQA_K, 12, 91, wtypes=wt)
    model = llama.Llama(hp, w, context_size=64)
    toks = np.random.default_rng(17).integers(0, hp["n_vocab"], 20).astype(np.int32)
    a, ka, va, ran_a = _decode(G, model, toks, 7, 1)
    b, kb, vb, ran_b = _decode(G, model, toks, 7, 0)
    assert ran_a == 13 and ran_b == 0
    worst = max(float(np.max(np.abs(x - y))) / float(y.std()) for x, y in zip(a, b))
    print(f"mixed Q4_K/Q6_K: K plan vs executor worst |dlogit|/std = {worst:.2e}")
    assert worst <= 8e-2
    assert all(int(np.argmax(x)) == int(np.argmax(y)) for x, y in zip(a, b))
    model.free()


@pytest.mark.parametrize("case", ["q4_k", "q6_k", "k_m mix", "gqa q4_k", "q5_k", "q3_k", "q2_k", "q5_k_m mix"])
def test_big_workgroup_k_mat_vecs_equal_the_helper_launch_form(G, O, case):
    """The K plan's decode mat-vecs as one wave of 1024-thread workgroups that stage the activation themselves (k_mmvq_kbig,
    kernels/kquant_big.h: norm / silu·mul + Q8_K inside the mat-vec's staging, 6 launches per layer) against option kbig = 0
    (k_mmvq_k behind helper launches, 10 per layer): the row dots and the quantizer are the same expressions in the same order,
    so logits, K/V and greedy ids are BIT-IDENTICAL; whole model and a stage of a layer split."""
    from llm_amd import llama
    wt = {"q4_k": 12, "q6_k": 14, "k_m mix": 12, "gqa q4_k": 12, "q5_k": 13, "q3_k": 11, "q2_k": 10, "q5_k_m mix": 13}[case]
    hp0 = GQA_K if case == "gqa q4_k" else TINY_K
    wtypes = None
    if case in ("k_m mix", "q5_k_m mix"):
        wtypes = {"output.weight": 14}
        for il in range(hp0["n_layer"]):
            wtypes[f"layers.{il}.attention.wv.weight"] = 14
            wtypes[f"layers.{il}.feed_forward.w2.weight"] = 14
    hp, w = _model(O, hp0, wt, 31, wtypes)
    model = llama.Llama(hp, w, context_size=96)
    toks = np.random.default_rng(12).integers(0, hp["n_vocab"], 30).astype(np.int32)
    res = {}
    try:
        for kbig in (1, 0):
            G.set_option("kbig", kbig)
            outs, k, v, ran = _decode(G, model, toks, 11, 1)
            sess = model.start_session(n_batch=8)
            sess.feed_prompt(toks[:9])
            ids = [sess.infer_next_token() for _ in range(12)]
            sess.free()
            res[kbig] = (outs, k, v, ran, ids)
    finally:
        G.set_option("kbig", 1)
        model.free()
    assert res[1][3] == 19 and res[0][3] == 19  # both forms are the K plan
    for a, b in zip(res[1][0], res[0][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(res[1][1], res[0][1]) and np.array_equal(res[1][2], res[0][2])
    assert res[1][4] == res[0][4]


def test_k_plan_greedy_tokens_equal_the_executor(G, O):
    """infer_next_token (greedy) over 24 tokens: the K plan and the executor pick the same ids."""
    from llm_amd import llama
    hp, w = _model(O, TINY_K, 12, 3)
    model = llama.Llama(hp, w, context_size=64)
    prompt = np.random.default_rng(8).integers(0, hp["n_vocab"], 9).astype(np.int32)
    ids = {}
    for plan_k in (1, 0):
        G.set_option("plan_k", plan_k)
        sess = model.start_session(n_batch=8)
        sess.feed_prompt(prompt)
        ids[plan_k] = [sess.infer_next_token() for _ in range(24)]
        sess.free()
    G.set_option("plan_k", 1)
    assert ids[1] == ids[0]
    model.free()


def test_k_plan_stages_of_a_layer_split_reproduce_the_unsplit_session(G, O):
    """A K-quant model layer-split over three (virtual) device slots of one process: every stage runs the K plan on its own
    layers (the first starts from get_rows, the later ones from the hand-off buffer, only the last holds the lm_head), the f32
    residual crosses unchanged — bit-identical to the unsplit session."""
    import os
    from llm_amd import llama
    assert G.lib().ggml_hip_get_main_device() == 0  # every test (and every entry point) leaves the main device as it found it
    hp0 = dict(n_vocab=256, n_embd=256, n_head=4, n_head_kv=4, n_layer=5, n_rot=64, n_ff=512, n_mult=32)
    hp, w = _model(O, hp0, 12, 41)
    toks = np.random.default_rng(4).integers(0, hp["n_vocab"], 20).astype(np.int32)

    def run(model):
        sess = model.start_session(n_batch=8)
        sess.feed_prompt(toks[:8])
        k0 = _stat(G, "kplan_tokens")
        outs = [sess.evaluate(toks[i:i + 1])[-1].copy() for i in range(8, 20)]
        ran = _stat(G, "kplan_tokens") - k0
        k, v = sess.get_kv()
        sess.free()
        return outs, k, v, ran

    whole = llama.Llama(hp, w, context_size=64)
    ref = run(whole)
    whole.free()
    assert ref[3] == 12
    os.environ["GGML_HIP_VIRTUAL_DEVICES"] = "3"
    os.environ["GGML_HIP_LAYER_SPLIT"] = "3"
    try:
        split = llama.Llama(hp, w, context_size=64)
        assert split.stages() == [(0, 2, 0), (2, 3, 1), (3, 5, 2)]
        got = run(split)
        split.free()
    finally:
        os.environ.pop("GGML_HIP_LAYER_SPLIT", None)
        G.lib().ggml_hip_set_layer_split(None, 0)
        G.lib().ggml_hip_set_main_device(0)
        os.environ.pop("GGML_HIP_VIRTUAL_DEVICES", None)
    assert got[3] == 12  # statistics are per device slot: slot 0's stage ran every token as a K plan (the others do on their slots)
    for a, b in zip(ref[0], got[0]):
        assert np.array_equal(a, b)
    assert np.array_equal(ref[1], got[1]) and np.array_equal(ref[2], got[2])


@pytest.mark.parametrize("wtype", [12, 14, 10])
def test_k_plan_takes_prompt_chunks_of_up_to_31_tokens(G, O, wtype):
    """InferenceSession::feed_prompt at the reference's default n_batch = 8 (crates/llm-base/src/inference_session.rs:315-316, :837)
    and at the batch sizes its GPU users set: chunks of 8, 5, 3, 2, 11 and 31 tokens of a K-quant model run on the K plan (columns of
    the mat-vecs in passes of 8 / 4 / 2 / 1 exactly as the node-by-node executor takes them; 32 and more go to the executor and
    its f16 GEMMs); logits of every token, the embeddings and the K/V cache against the executor and the oracle."""
    from llm_amd import llama
    hp, w = _model(O, GQA_K, wtype, 23)
    ctx = 96
    model = llama.Llama(hp, w, context_size=ctx)
    toks = np.random.default_rng([wtype, 9]).integers(0, hp["n_vocab"], 72).astype(np.int32)
    cuts = [(0, 8), (8, 13), (13, 16), (16, 18), (18, 19), (19, 30), (30, 38), (38, 69)]  # 8, 5, 3, 2, 1, 11, 8, 31

    def run(plan_k):
        G.set_option("plan_k", plan_k)
        G.set_option("k_prompt_min", 32)  # (by default chunks of 12 and more tokens take the prompt plan on f16 copies: tests below)
        try:
            sess = model.start_session(n_batch=32)
            outs, ran = [], []
            for lo, hi in cuts:
                k0 = _stat(G, "kplan_tokens")
                outs.append(sess.evaluate(toks[lo:hi]).copy())
                ran.append(_stat(G, "kplan_tokens") - k0)
            k, v = sess.get_kv()
            sess.free()
        finally:
            G.set_option("plan_k", 1)
            G.set_option("k_prompt_min", 12)
        return outs, ran, k, v

    a, ran_a, ka, va = run(1)
    b, ran_b, kb, vb = run(0)
    assert ran_a == [8, 5, 3, 2, 1, 11, 8, 31] and ran_b == [0] * 8
    worst = 0.0
    for x, y in zip(a, b):
        assert x.shape == y.shape
        worst = max(worst, float(np.max(np.abs(x - y))) / float(y.std()))
    print(f"type {wtype}: chunks, K plan vs executor worst |dlogit|/std = {worst:.2e}")
    assert worst <= 8e-2
    Eg = hp["n_embd"] // (hp["n_head"] // hp["n_head_kv"])
    assert np.array_equal(ka[:69 * Eg], kb[:69 * Eg])  # layer 0's K rows: same launches, same bits
    # the oracle, chunk by chunk on the device's own K/V state
    sess = model.start_session(n_batch=32)
    orc = O.Llama(hp, w, ctx)
    worst = 0.0
    for lo, hi in cuts:
        k, v = sess.get_kv()
        orc.memory_k[:] = k[:orc.memory_k.size]
        orc.memory_v[:] = v[:orc.memory_v.size]
        orc.n_past = lo
        got = sess.evaluate(toks[lo:hi])
        ref = orc.evaluate(toks[lo:hi], mode=O.ref_mode())
        worst = max(worst, float(np.max(np.abs(got - ref))) / float(ref.std()))
    sess.free()
    print(f"type {wtype}: chunks, K plan vs oracle worst |dlogit|/std = {worst:.2e}")
    assert worst <= 8e-2
    model.free()


def test_k_plan_at_long_context_uses_the_split_attention(G, O):
    """On long contexts (option attn_split: from 768 positions by default, 512 here) the K plan's attention is the position-split one-launch kernel (k_attn_split_one): against the
    node-by-node executor and the oracle at ~600 and ~1000 positions of a 1024-position context."""
    from llm_amd import llama
    hp, w = _model(O, TINY_K, 12, 57)
    ctx = 1024
    model = llama.Llama(hp, w, context_size=ctx)
    toks = np.random.default_rng(31).integers(0, hp["n_vocab"], ctx).astype(np.int32)
    res = {}
    for plan_k in (1, 0):
        G.set_option("plan_k", plan_k)
        G.set_option("attn_split", 512)  # the split path from 512 positions on (default: 768)
        G.set_option("attn_fused", 0)    # the prompts' attention as the executor's three launches: the same K/V on both sides (the
                                         # prompt plan of a K-quant model is then bit-identical to the executor, see below)
        try:
            s = model.start_session(n_batch=512)
            outs = []
            s.feed_prompt(toks[:600])
            b0 = _stat(G, "attn_split_tokens")
            outs += [s.evaluate(toks[600 + i:601 + i])[-1].copy() for i in range(4)]
            s.feed_prompt(toks[604:1010])
            outs += [s.evaluate(toks[1010 + i:1011 + i])[-1].copy() for i in range(6)]
            split = _stat(G, "attn_split_tokens") - b0
            s.free()
        finally:
            G.set_option("plan_k", 1)
            G.set_option("attn_split", 1)
            G.set_option("attn_fused", 1)
        res[plan_k] = (outs, split)
    assert res[1][1] == 10 and res[0][1] == 0
    assert _stat(G, "fused_attn_timeouts") == 0
    worst = max(float(np.max(np.abs(x - y))) / float(y.std()) for x, y in zip(res[1][0], res[0][0]))
    print(f"long context: K plan vs executor worst |dlogit|/std = {worst:.2e}")
    assert worst <= 4e-2
    model.free()


def test_mixed_k_quant_file_loads_and_decodes_on_the_k_plan(G, O, tmp_path):
    """llm::load of a *_K_M-style GGJT v3 file (wv / w2 / output Q6_K, the rest Q4_K; written the way
    crates/ggml/src/format/saver.rs does, every tensor record with its own type) through the C++ mmap loader: prompt feed at
    n_batch = 8 and greedy decode on the K plan, bit-identical to the same weights handed over in memory."""
    from llm_amd import llama, synth
    hp0 = dict(n_vocab=256, n_embd=512, n_head=4, n_head_kv=4, n_layer=3, n_rot=128, n_ff=768, n_mult=32)  # the file format has no n_head_kv
    wt = {"output.weight": 14}
    for i in range(hp0["n_layer"]):
        wt[f"layers.{i}.attention.wv.weight"] = 14
        wt[f"layers.{i}.feed_forward.w2.weight"] = 14
    hp, w = _model(O, hp0, 12, 19, wtypes=wt)
    path = tmp_path / "q4_k_m.bin"
    synth.write_ggjt(str(path), hp, w)
    toks = np.random.default_rng(3).integers(0, hp["n_vocab"], 21).astype(np.int32)

    def run(model):
        s = model.start_session(n_batch=8)
        k0 = _stat(G, "kplan_tokens")
        s.feed_prompt(toks)
        ids = [s.infer_next_token() for _ in range(10)]
        ran = _stat(G, "kplan_tokens") - k0
        last = s.last_logits().copy()
        s.free()
        return ids, last, ran

    mem = llama.Llama(hp, w, context_size=64)
    a = run(mem)
    mem.free()
    fil = llama.Llama.load(str(path), context_size=64)
    b = run(fil)
    fil.free()
    assert a[2] == 21 + 10 and b[2] == 21 + 10  # chunks of 8, 8, 5 and ten single tokens: all on the K plan
    assert a[0] == b[0] and np.array_equal(a[1], b[1])


# ---- prompt batches of a K-quant model on the prompt plan (llama_plan.inc plan_launch_prompt with k_prompt_weights) -----------------
KP_GQA = dict(n_vocab=512, n_embd=512, n_head=8, n_head_kv=2, n_layer=2, n_rot=64, n_ff=768, n_mult=32)
KP_SPLITK = dict(n_vocab=256, n_embd=1024, n_head=8, n_head_kv=4, n_layer=2, n_rot=128, n_ff=2048, n_mult=32)  # every GEMM splits K in two


def _k_m_types(hp):
    from llm_amd import synth
    t = {}
    for name in synth.tensor_shapes(hp):
        if name == "output.weight" or name.endswith("attention.wv.weight") or name.endswith("feed_forward.w2.weight"):
            t[name] = 14
    return t


def _run_prompt(G, model, chunks, plan):
    G.set_option("plan_prompt", plan)
    G.set_option("mmq_min", 12)  # the executor takes the f16 GEMM from 12 tokens on too (the plan does by itself: option k_prompt_min)
    try:
        sess = model.start_session(n_batch=192)
        outs = []
        for c in chunks:
            p0, g0, k0 = _stat(G, "prompt_plan_tokens"), _stat(G, "generic_graphs"), _stat(G, "kplan_tokens")
            r = sess.evaluate(c, want_embeddings=True)
            dp, dg, dk = _stat(G, "prompt_plan_tokens") - p0, _stat(G, "generic_graphs") - g0, _stat(G, "kplan_tokens") - k0
            if len(c) >= 12:
                assert (dp, dg) == ((len(c), 0) if plan else (0, 1)), (len(c), plan, dp, dg)
            else:
                assert dk == len(c)  # chunks of up to 11 tokens: the K plan's multi-token form, whatever plan_prompt says
            outs.append(r)
        k, v = sess.get_kv()
        sess.free()
    finally:
        G.set_option("plan_prompt", 1)
        G.set_option("mmq_min", 32)
    return outs, k, v


@pytest.mark.parametrize("wtype", KTYPES + ["k_m"])
@pytest.mark.parametrize("cfg", ["gqa", "splitk"])
def test_k_prompt_plan_is_bit_identical_to_the_node_by_node_executor(G, O, wtype, cfg):
    """Batches of 12 and more tokens of a K-quant model: the prompt plan (13 launches per layer; token operand = the Q8_K round trip
    of k_quant_act_f16_k inside k_p_norm_quant / k_p_silu_mul_quant / k_p_quant4, weights = their resident f16 copies) against the
    node-by-node executor (mul_mat_k_gemm per matrix): logits of every token, the final-norm rows and the K/V, bit for bit (the
    three-launch attention on both sides, as in test_prompt_plan_gpu.py)."""
    from llm_amd import llama
    if cfg == "splitk" and wtype not in (12, "k_m"):
        pytest.skip("the K-split paths do not depend on the block format")
    hp0 = {"gqa": KP_GQA, "splitk": KP_SPLITK}[cfg]
    base = 12 if wtype == "k_m" else wtype
    hp, w = _model(O, hp0, base, 91, wtypes=_k_m_types(hp0) if wtype == "k_m" else None)
    model = llama.Llama(hp, w, context_size=512)
    toks = np.random.default_rng([base, len(cfg)]).integers(0, hp["n_vocab"], 400).astype(np.int32)
    # 64 at n_past 0 (makes the f16 copies); 33 (ragged); 3 (the K plan's multi-token form in between); 20 (the prompt plan from 12
    # tokens on for a K-quant model); 110 at n_past 120; 128 at n_past 230
    chunks = [toks[0:64], toks[64:97], toks[97:100], toks[100:120], toks[120:230], toks[230:358]]
    G.set_option("attn_fused", 0)
    try:
        a, ka, va = _run_prompt(G, model, chunks, 1)
        b, kb, vb = _run_prompt(G, model, chunks, 0)
    finally:
        G.set_option("attn_fused", 1)
        model.free()
    for i, ((la, ea), (lb, eb)) in enumerate(zip(a, b)):
        assert la.shape == (len(chunks[i]), hp["n_vocab"])
        assert np.array_equal(la, lb), (cfg, wtype, i, float(np.max(np.abs(la - lb))))
        assert np.array_equal(ea, eb), (cfg, wtype, i)
    assert np.array_equal(ka, kb) and np.array_equal(va, vb)


@pytest.mark.parametrize("wtype", [12, 14])
def test_k_prompt_plan_matches_the_oracle(G, O, wtype):
    """... and against the CPU oracle with the fused attention kernel (the default): the f16 GEMM's bound of the other formats
    (tests/test_prompt_plan_gpu.py)."""
    from llm_amd import llama
    hp, w = _model(O, KP_GQA, wtype, 17)
    model = llama.Llama(hp, w, context_size=160)
    orc = O.Llama(hp, w, 160)
    toks = np.random.default_rng(3).integers(0, hp["n_vocab"], 144).astype(np.int32)
    sess = model.start_session(n_batch=96)
    try:
        for c in (toks[:96], toks[96:128], toks[128:144]):  # (16 tokens: the prompt plan from 12 on for a K-quant model)
            p0 = _stat(G, "prompt_plan_tokens")
            got = sess.evaluate(c)
            assert _stat(G, "prompt_plan_tokens") - p0 == len(c)
            ref = orc.evaluate(c, mode=O.ref_mode())
            std = float(ref.std())
            rms = float(np.sqrt(np.mean((got - ref) ** 2))) / std
            mx = float(np.max(np.abs(got - ref))) / std
            print(f"K prompt plan type {wtype}, {len(c)} tokens: rms {rms:.2e} max {mx:.2e} of std(logits)")
            assert rms <= 2e-2, rms
            assert mx <= 1e-1
            k, v = sess.get_kv()
            orc.memory_k[:] = k[:orc.memory_k.size]
            orc.memory_v[:] = v[:orc.memory_v.size]
    finally:
        sess.free()
        model.free()
