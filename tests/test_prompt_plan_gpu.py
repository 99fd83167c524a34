"""The fused prompt plan (llama_plan.inc plan_launch_prompt, kernels/prompt.h): a prompt batch of >= 32 tokens
(crates/llm-base/src/inference_session.rs:315-316 feeds n_batch tokens per Model::evaluate; graph of
crates/models/llama/src/lib.rs:166-362) as 13 launches per layer instead of 24.

The plan performs the node-by-node executor's floating-point operations in the same order on the same values (same GEMM
kernels and split-K choice, same rms_norm / RoPE / softmax / SiLU / re-quantization arithmetic), so the comparison is
BIT-EXACT: logits of every token, the final-norm embedding, and the K/V cache — for all five block formats, grouped-query
attention, ragged batch sizes, and batches that start at n_past > 0.  Parity of the generic executor with the oracle
is established in test_llama_gpu.py; one case here checks the plan against the oracle directly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _three_launch_attention(request, G):
    """The bit-identity tests of this file hold the prompt plan to the node-by-node executor, which runs K.Q / softmax /
    V.P as three kernels: the plan does the same here.  The fused attention kernel (kernels/prompt_attn.h, the default)
    equals that path up to one f16 rounding of about one probability per hundred rows; it has its own test below and is
    what every oracle comparison elsewhere runs on."""
    fused = "fused_prompt_attention" in request.node.name
    G.set_option("attn_fused", 1 if fused else 0)
    yield
    G.set_option("attn_fused", 1)

GQA = dict(n_vocab=256, n_embd=128, n_head=4, n_head_kv=2, n_layer=2, n_rot=32, n_ff=352, n_mult=32)
WIDE = dict(n_vocab=320, n_embd=256, n_head=4, n_head_kv=4, n_layer=3, n_rot=64, n_ff=512, n_mult=32)  # K/32 even everywhere: DMA GEMM
# K >= 1024 and few tiles: every GEMM splits K in two — atomics into a zeroed dst in the node-by-node executor, partial
# tiles added by the consuming kernel in the plan; grouped-query attention on top (wk / wv narrower than wq)
SPLITK = dict(n_vocab=256, n_embd=1024, n_head=8, n_head_kv=4, n_layer=2, n_rot=128, n_ff=2048, n_mult=32)


def _stat(G, key):
    return int(G.lib().ggml_hip_get_stat(key.encode()))


def _run(G, model, chunks, plan, want_emb=False):
    G.set_option("plan_prompt", plan)
    sess = model.start_session(n_batch=192)
    outs = []
    for c in chunks:
        p0, g0 = _stat(G, "prompt_plan_tokens"), _stat(G, "generic_graphs")
        r = sess.evaluate(c, want_embeddings=want_emb)
        dp, dg = _stat(G, "prompt_plan_tokens") - p0, _stat(G, "generic_graphs") - g0
        if len(c) >= 32:
            assert (dp, dg) == ((len(c), 0) if plan else (0, 1)), (len(c), plan, dp, dg)
        outs.append(r)
    k, v = sess.get_kv()
    sess.free()
    G.set_option("plan_prompt", 1)
    return outs, k, v


@pytest.mark.parametrize("wtype", [2, 3, 6, 7, 8])
@pytest.mark.parametrize("cfg", ["tiny", "gqa", "wide", "splitk", "splitk_unfused"])
def test_prompt_plan_is_bit_identical_to_the_node_by_node_executor(G, wtype, cfg):
    from llm_amd import llama, synth
    if cfg.startswith("splitk") and wtype not in (2, 7):
        pytest.skip("the K-split paths do not depend on the block format: two formats")
    hp0 = {"tiny": synth.TINY, "gqa": GQA, "wide": WIDE, "splitk": SPLITK, "splitk_unfused": SPLITK}[cfg]
    G.set_option("mmq_fuse", 0 if cfg == "splitk_unfused" else 3)
    hp, w = synth.make_llama(hp0, wtype, seed=5)
    model = llama.Llama(hp, w, context_size=512)
    toks = np.random.default_rng([wtype, len(cfg)]).integers(0, hp["n_vocab"], 400).astype(np.int32)
    # N = 64 at n_past 0; 33 (ragged, one partial tile); 3 (multi-token plan in between); 20 (below mmq_min: node by node on
    # both sides); 110 (n_past 120); 128 at n_past 230; one token; 41 at the ODD n_past 359 (k_p_qkv_post stores V one f16 at a time
    # there: its two-token stores need the tile's first cache position even)
    chunks = [toks[0:64], toks[64:97], toks[97:100], toks[100:120], toks[120:230], toks[230:358], toks[358:359], toks[359:400]]
    a, ka, va = _run(G, model, chunks, 1, want_emb=True)
    b, kb, vb = _run(G, model, chunks, 0, want_emb=True)
    for i, ((la, ea), (lb, eb)) in enumerate(zip(a, b)):
        assert la.shape == (len(chunks[i]), hp["n_vocab"])
        assert np.array_equal(la, lb), (cfg, wtype, i, float(np.max(np.abs(la - lb))))
        assert np.array_equal(ea, eb), (cfg, wtype, i)
    assert np.array_equal(ka, kb) and np.array_equal(va, vb)
    G.set_option("mmq_fuse", 3)
    model.free()


def test_prompt_plan_matches_the_oracle(G, O):
    """Direct check against the CPU oracle (mode 0 = ggml's exact integer block dots): the prompt GEMM rounds both
    operands to f16, so the bound is the f16 GEMM's (test_llama_gpu.py: RMS 2e-2 of std(logits), EDGE on the maximum)."""
    from llm_amd import llama, synth
    hp, w = synth.make_llama(synth.TINY, 2, seed=1234)
    model = llama.Llama(hp, w, context_size=128)
    orc = O.Llama(hp, w, 128)
    toks = np.random.default_rng(3).integers(0, hp["n_vocab"], 96).astype(np.int32)
    sess = model.start_session(n_batch=64)
    for c in (toks[:64], toks[64:96]):
        p0 = _stat(G, "prompt_plan_tokens")
        got = sess.evaluate(c)
        assert _stat(G, "prompt_plan_tokens") - p0 == len(c)
        ref = orc.evaluate(c, mode=O.ref_mode())
        std = float(ref.std())
        rms = float(np.sqrt(np.mean((got - ref) ** 2))) / std
        assert rms <= 2e-2, rms
        assert float(np.max(np.abs(got - ref))) / std <= 1e-1
        k, v = sess.get_kv()
        orc.memory_k[:] = k
        orc.memory_v[:] = v
    sess.free()
    model.free()


def test_prompt_plan_stage_of_a_layer_split(G):
    """A first stage of a layer split (no final norm: the residual is handed on) and a later stage (residual received in
    the hand-off buffer): with the plan and with the node-by-node executor the stages hand on the same residual, produce
    the same logits and leave the same K/V."""
    from llm_amd import synth
    from llm_amd.pipeline import GpuStage
    hp, w = synth.make_llama(synth.TINY, 2, seed=9)
    toks = np.random.default_rng(1).integers(0, hp["n_vocab"], 48).astype(np.int32)

    def run(plan):
        G.set_option("plan_prompt", plan)
        st0 = GpuStage(hp, {k: v for k, v in w.items() if k in synth.stage_tensor_names(hp, 0, 1)}, (0, 1), 64, n_batch=64)
        st1 = GpuStage(hp, {k: v for k, v in w.items() if k in synth.stage_tensor_names(hp, 1, 2)}, (1, 2), 64, n_batch=64)
        st0.new_sequence(0)
        st1.new_sequence(0)
        p0, g0 = _stat(G, "prompt_plan_tokens"), _stat(G, "generic_graphs")
        res = st0.evaluate(0, toks, None)
        st1.evaluate(0, toks, res)
        dp, dg = _stat(G, "prompt_plan_tokens") - p0, _stat(G, "generic_graphs") - g0
        assert (dp, dg) == ((2 * toks.size, 0) if plan else (0, 2)), (plan, dp, dg)
        out = [res.copy(), st1.sessions[0].last_logits()] + st0.sessions[0].get_kv() + st1.sessions[0].get_kv()
        st0.free()
        st1.free()
        G.set_option("plan_prompt", 1)
        return out

    for x, y in zip(run(1), run(0)):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("wtype", [2, 7])
def test_prompt_plan_rows_longer_than_512_positions(G, wtype):
    """Score rows of more than 512 positions take the softmax kernel's multi-pass branch (row not held in registers), the
    attention GEMMs see several key tiles with the causal skip active at n_past > 0, and a 500-token batch has a ragged
    last token tile: still bit-identical to the node-by-node executor."""
    from llm_amd import llama, synth
    hp, w = synth.make_llama(GQA, wtype, seed=13)
    model = llama.Llama(hp, w, context_size=1024)
    toks = np.random.default_rng([wtype, 99]).integers(0, hp["n_vocab"], 900).astype(np.int32)
    chunks = [toks[0:64], toks[64:564], toks[564:764], toks[764:900]]  # T = 64, 564, 764, 900

    def run(plan):
        G.set_option("plan_prompt", plan)
        sess = model.start_session(n_batch=512)
        outs = [sess.evaluate(c) for c in chunks]
        k, v = sess.get_kv()
        sess.free()
        G.set_option("plan_prompt", 1)
        return outs, k, v

    a, ka, va = run(1)
    b, kb, vb = run(0)
    for i, (la, lb) in enumerate(zip(a, b)):
        assert np.array_equal(la, lb), (wtype, i, float(np.max(np.abs(la - lb))))
    assert np.array_equal(ka, kb) and np.array_equal(va, vb)
    model.free()


@pytest.mark.parametrize("wtype", [2, 3, 6, 7, 8])
@pytest.mark.parametrize("cfg", ["wide", "splitk", "tiny"])
def test_gemm_on_resident_f16_weight_copies_is_bit_identical(G, wtype, cfg):
    """k_mmq_w16_p8 (both operands by LDS-DMA from an f16 copy of the weights made once, kernels/mmq_w16.h) against the
    kernels that dequantize the blocks in LDS: the same f16 values in the same MFMA sequence, so the same bits.  The
    copies exist only where K / 32 is even (tiny: w2 keeps dequantizing) and are released with the model."""
    from llm_amd import llama, synth
    if cfg == "splitk" and wtype not in (2, 7):
        pytest.skip("two formats for the K-split shapes")
    hp, w = synth.make_llama({"wide": WIDE, "splitk": SPLITK, "tiny": synth.TINY}[cfg], wtype, seed=8)
    toks = np.random.default_rng([wtype, 17]).integers(0, hp["n_vocab"], 300).astype(np.int32)
    chunks = [toks[0:48], toks[48:200], toks[200:300]]
    res = {}
    for w16 in (1, 0):
        G.set_option("mmq_w16", w16)
        b0 = _stat(G, "w16_bytes")
        model = llama.Llama(hp, w, context_size=512)
        res[w16] = _run(G, model, chunks, 1)
        held = _stat(G, "w16_bytes") - b0
        assert (held > 0) == bool(w16), (w16, held)
        model.free()
        assert _stat(G, "w16_bytes") == b0  # released with the weights
    G.set_option("mmq_w16", 1)
    (a, ka, va), (b, kb, vb) = res[1], res[0]
    for la, lb in zip(a, b):
        assert np.array_equal(la, lb), (cfg, wtype, float(np.max(np.abs(la - lb))))
    assert np.array_equal(ka, kb) and np.array_equal(va, vb)


def test_prompt_plan_with_a_k_split_on_wk_wv_only(G):
    """LLaMA-2-70B-like attention shape (8192-wide, 64 heads, 8 K/V heads): at 512 tokens wq has 256 tiles and runs
    unsplit while wk / wv have 32 each and split K — the three cannot share a launch, and the kernel that consumes q | k | v
    adds two partials for all of them or for none: wq's second partial is zero-filled.  Bit-identical to the node-by-node
    executor (which splits exactly the same matrices, through atomics)."""
    from llm_amd import llama, synth
    hp0 = dict(n_vocab=256, n_embd=8192, n_head=64, n_head_kv=8, n_layer=1, n_rot=128, n_ff=1024, n_mult=32)
    hp, w = synth.make_llama_fast(hp0, G.TYPE_Q4_0, seed=21)
    model = llama.Llama(hp, w, context_size=1024)
    toks = np.random.default_rng(77).integers(0, hp["n_vocab"], 600).astype(np.int32)
    chunks = [toks[0:512], toks[512:600]]

    def run(plan):
        G.set_option("plan_prompt", plan)
        sess = model.start_session(n_batch=512)
        outs = []
        for c in chunks:
            p0 = _stat(G, "prompt_plan_tokens")
            outs.append(sess.evaluate(c))
            assert _stat(G, "prompt_plan_tokens") - p0 == (len(c) if plan else 0)
        k, v = sess.get_kv()
        sess.free()
        G.set_option("plan_prompt", 1)
        return outs, k, v

    a, ka, va = run(1)
    b, kb, vb = run(0)
    for la, lb in zip(a, b):
        assert np.isfinite(la).all() and np.array_equal(la, lb), float(np.max(np.abs(la - lb)))
    assert np.array_equal(ka, kb) and np.array_equal(va, vb)
    model.free()


def test_exp_le0_is_expf_for_every_f16_argument_the_softmax_can_pass(G):
    """kernels/prompt_attn.h exp_le0 (expf's own operation sequence with one clamp in place of its two range checks) against the
    device library's expf, both rounded to f16 as ggml's soft_max stores them, for ALL 65536 f16 bit patterns: identical for every
    x <= 0 (both zeros, -inf included) and NaN for every NaN — the only arguments f16(score - row maximum) can be.  Against
    numpy's exp in f64 the f16 results sit within one f16 ulp (correctly rounded or its neighbour)."""
    fast = np.zeros(65536, np.uint16)
    ref = np.zeros(65536, np.uint16)
    assert G.lib().ggml_hip_debug_exp_le0(fast.ctypes.data, ref.ctypes.data) == 0
    x = np.arange(65536, dtype=np.uint32).astype(np.uint16).view(np.float16)
    le0 = (x <= 0)  # -0, +0, negative normals / subnormals, -inf
    assert le0.sum() == 2 ** 15 - 1024 + 1 + 1  # every pattern with the sign bit set that is not a NaN, and +0
    assert np.array_equal(fast[le0], ref[le0]), np.flatnonzero(le0 & (fast != ref))[:8]
    nan = np.isnan(x)
    assert np.isnan(fast[nan].view(np.float16)).all() and np.isnan(ref[nan].view(np.float16)).all()
    assert fast[x == 0].view(np.float16).tolist() == [1.0, 1.0] and float(fast.view(np.float16)[0xFC00]) == 0.0  # exp(+-0) = 1, exp(-inf) = 0
    fin = le0 & np.isfinite(x)
    want = np.exp(x[fin].astype(np.float64))
    got = fast.view(np.float16)[fin].astype(np.float64)
    ulp = np.maximum(np.spacing(want.astype(np.float16)).astype(np.float64), 2.0 ** -24)
    assert np.all(np.abs(got - want) <= ulp)


def test_fused_prompt_attention_kernel_matches_the_three_launch_path(G):
    """kernels/prompt_attn.h (K.Q, scale + mask + softmax and V.P in one launch, scores in LDS) against k_gemm_f16 ->
    k_p_soft_max -> k_gemm_f16_b16 on random Q / K / V through ggml_hip_debug_prompt_attention: head sizes 32 / 64 / 128,
    grouped-query attention, ragged batches, rows up to ~1100 keys with 32 queries per workgroup (LDS limit 1184) and up to
    2237 keys with 16.  Both paths
    perform the same operations; their scores can differ in the last f32 bit, which moves about one probability in a
    hundred rows across an f16 rounding boundary (measured: 1..8 % of rows differ, by <= 5e-5): bound 2e-4 of max|out|."""
    import ctypes as C
    f = G.lib().ggml_hip_debug_prompt_attention
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_int]
    cases = [(64, 4, 4, 32, 0), (128, 4, 4, 32, 230), (33, 2, 1, 64, 500), (512, 8, 8, 128, 1), (200, 4, 2, 128, 700), (96, 4, 4, 32, 1000),
             (70, 4, 4, 128, 1050), (1, 4, 4, 64, 77),
             # rows longer than 1184 keys: 16 queries per workgroup (the last chunks of a 2048-token context)
             (64, 4, 4, 128, 1500), (100, 4, 2, 64, 1948), (37, 2, 2, 32, 2200), (512, 4, 4, 128, 1536)]
    for N, H, Hkv, D, n_past in cases:
        rng = np.random.default_rng([N, H, D, n_past])
        E, Eg, T = H * D, Hkv * D, n_past + N
        Cc = 1184 if T <= 1184 else 2304
        q = rng.standard_normal((N, E)).astype(np.float32)
        k = np.full((Cc, Eg), np.nan, np.float16)  # unwritten cache rows must not matter
        v = np.full((Eg, Cc), np.nan, np.float16)
        k[:T] = rng.standard_normal((T, Eg)).astype(np.float16)
        v[:, :T] = rng.standard_normal((Eg, T)).astype(np.float16)
        outs = []
        for fused in (1, 0):
            out = np.zeros((N, E), np.float32)
            assert f(q.ctypes.data, k.ctypes.data, v.ctypes.data, out.ctypes.data, N, E, Eg, H, n_past, Cc, 1.0 / np.sqrt(D), fused) == 0
            outs.append(out)
        assert np.isfinite(outs[0]).all() and np.isfinite(outs[1]).all(), (N, H, D, n_past)
        d = float(np.max(np.abs(outs[0] - outs[1]))) / float(np.max(np.abs(outs[1])))
        assert d <= 2e-4, (N, H, Hkv, D, n_past, d)
        # numpy reference of the same semantics (f64 accumulation): both within the f16-product noise
        kf, vf = k[:T].astype(np.float64), v[:, :T].astype(np.float64)
        qh = q.astype(np.float16).astype(np.float64)
        for h in (0, H - 1):
            hk = h // (H // Hkv)
            s = (qh[:, h * D:(h + 1) * D] @ kf[:, hk * D:(hk + 1) * D].T) / np.sqrt(D)
            for n in (0, N - 1):
                row = s[n, :n_past + n + 1]
                e = np.exp(row - row.max())
                pr = (e / e.sum()).astype(np.float16).astype(np.float64)
                ref = vf[hk * D:(hk + 1) * D, :n_past + n + 1] @ pr
                assert np.allclose(outs[0][n, h * D:(h + 1) * D], ref, rtol=0, atol=4e-3 * max(1.0, float(np.abs(ref).max()))), (N, D, n_past, h, n)


@pytest.mark.parametrize("wtype", [2, 3, 8])
@pytest.mark.parametrize("cfg", ["tiny", "gqa", "wide", "splitk"])
def test_fused_prompt_attention_with_its_quantizing_epilogue_in_the_plan(G, wtype, cfg):
    """In the prompt plan the fused attention kernel writes wo's GEMM operand itself (every 32-channel block of the merged
    row re-quantized to Q8 and stored as f16(d * q): k_p_quant4's arithmetic in the V.P epilogue; both block-scale kinds:
    f16-rounded for Q4_0 / Q8_0, f32 for Q4_1).  Against the plan with the three-launch attention + k_p_quant4: identical
    except where the attention outputs differ by their one-f16-rounding noise, which moves an int8 code now and then:
    every chunk within 4e-2 * std (the EDGE bound of the other tests), and chunks without such a flip agree to 1e-4 or
    exactly (seen: 0.0 beside 2e-2 in the same session).  The kernel also rotates Q itself (RoPE while loading the raw wq
    product, k_p_qkv_post then only handles K and V); "splitk": wq arrives as two K-split partials that it adds first."""
    from llm_amd import llama, synth
    hp0 = {"tiny": synth.TINY, "gqa": GQA, "wide": WIDE, "splitk": SPLITK}[cfg]
    hp, w = synth.make_llama(hp0, wtype, seed=13)
    model = llama.Llama(hp, w, context_size=512)
    toks = np.random.default_rng([wtype, 3]).integers(0, hp["n_vocab"], 330).astype(np.int32)
    chunks = [toks[0:64], toks[64:97], toks[97:225], toks[225:330]]
    outs = {}
    for fused in (1, 0):
        G.set_option("attn_fused", fused)
        sess = model.start_session(n_batch=192)
        p0 = _stat(G, "prompt_plan_tokens")
        outs[fused] = [sess.evaluate(c) for c in chunks]
        assert _stat(G, "prompt_plan_tokens") - p0 == sum(len(c) for c in chunks)
        sess.free()
    G.set_option("attn_fused", 1)
    ds = []
    for a, b in zip(outs[1], outs[0]):
        assert np.isfinite(a).all()
        ds.append(float(np.max(np.abs(a - b))) / float(b.std()))
    print(cfg, wtype, ["%.1e" % d for d in ds])
    # (seen: [0, 0, 0, 4.3e-2] for one model, all four chunks at 2..5e-2 for a 1024-wide one: more edges per chunk)
    assert max(ds) <= 6e-2, ds
    if cfg == "tiny":
        assert min(ds) <= 1e-4, ds
    model.free()
