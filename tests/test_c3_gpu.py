"""BASELINE configs[2] ("C3": LLaMA-7B Q4_0, 512-token prefill on the matrix cores) and configs[1] at the bench's own
operating point, against the CPU oracle AT THE REAL SHAPES — what tests/test_prompt_plan_gpu.py (128-wide toy, GPU
variants held to each other) and tests/test_fullsize_gpu.py (random blocks, ctx 32) do not cover:

* the prompt GEMM at K = 4096 / 11008, M = 11008 / 4096 / 32000, N = 512 (crates/models/llama/src/lib.rs:194-352 at
  n_batch = 512, crates/llm-base/src/inference_session.rs:315), every kernel the default build can pick, against oracle
  mode 0 on sampled weight rows (the oracle needs ~1 s per 256 rows x 512 tokens; rows are independent);
* two full-size 7B layers + the 32000-row lm_head through the fused prompt plan at N = 512, with the bench's gaussian
  weights, K/V synchronised, against oracle mode 0 — with the resident-f16-copy GEMM and with the
  in-LDS-dequant GEMM;
* full-size 7B decode with gaussian weights at ctx 2048, n_past >= 133 (the bench's timed region), oracle on the
  session's own K/V.

Tolerances.  Op level: the f16 GEMM's (both operands rounded to f16, unit roundoff 2^-11, f32 accumulate):
|err| <= 1.1e-3 * sum_k |w||x| and RMS <= 1e-4 of that scale.  Logits: the yardstick is the reference's OWN ambiguity on the
same model and tokens, measured in the test — the oracle against itself with its f32 block sums added in reverse order (a
second legal order of ggml's vec_dot; upstream's scalar and AVX2 branches are as far apart, oracle mode 2) = `band`, and
against its math mode (no activation quantization) = `floor`.  With the bench's gaussian weights a 7B-wide model is ~15x more
sensitive than with random blocks (2 layers, 8 tokens, on the CPU: band 7.8e-2 std / floor 1.1e-1 vs 5e-3 / 1.1e-2; the toy's
6e-2 is therefore not an artefact of its width).  The device must stay within 2 x band of the forward-order oracle (max and
RMS) and below the floor; measured values are printed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Q4_0 = 2
EDGE = 4e-2  # tests/test_llama_gpu.py: one int8 activation quant on a rounding edge


def _stat(G, key):
    return int(G.lib().ggml_hip_get_stat(key.encode()))


def _mul_mat_gpu(G, wtype, W_raw, M, K, X):
    N = X.shape[0]
    mem = W_raw.nbytes + X.nbytes + M * N * 4 + (1 << 20)
    with G.Context(mem) as ctx:
        w = ctx.tensor_from(W_raw, wtype, (K, M)).set_name("w")
        w.transfer_to_gpu()
        x = ctx.tensor_from(X, G.TYPE_F32, (K, N)).set_name("x")
        y = ctx.op_mul_mat(w, x)
        g = ctx.graph().build_forward_expand(y)
        g.compute()
        return y.read_data().reshape(N, M)


@pytest.mark.parametrize("shape", [(11008, 4096), (4096, 11008), (32000, 4096), (12288, 4096)])
def test_c3_gemm_shapes_match_the_oracle_on_sampled_rows(G, O, shape):
    M, K = shape
    N = 512
    rng = np.random.default_rng([M, K, N, 3])
    W = (0.02 * rng.standard_normal((M, K), dtype=np.float32))
    X = rng.standard_normal((N, K), dtype=np.float32)
    X[:, ::97] *= 6.0  # a few heavy channels, as post-norm activations have
    W_raw = G.quantize(Q4_0, W)
    del W
    rows = np.sort(rng.choice(M, 192, replace=False))
    rows[:4] = [0, 1, M - 2, M - 1]  # first / last tile rows
    rows = np.unique(rows)
    rb = O.row_bytes(Q4_0, K)
    sub = np.concatenate([W_raw[r * rb:(r + 1) * rb] for r in rows])
    exact = O.mul_mat(Q4_0, sub, len(rows), K, X, mode=O.ref_mode())
    Wd = np.stack([O.dequantize(Q4_0, sub[i * rb:(i + 1) * rb], K) for i in range(len(rows))])
    scale = np.abs(X) @ np.abs(Wd).T
    outs = {}
    for name, opts in (("default", {}), ("t256", {"mmq_t256": 2}), ("w16_128", {"mmq_t256": 0}), ("dma_p8", {"mmq_w16": 0})):
        for k, v in opts.items():
            G.set_option(k, v)
        try:
            c0 = {k: _stat(G, "mmq_launches_" + k) for k in ("w16_256", "w16_p8", "dma_p8")}
            got = _mul_mat_gpu(G, Q4_0, W_raw, M, K, X)
            ran = [k for k in c0 if _stat(G, "mmq_launches_" + k) > c0[k]]
        finally:
            for k in opts:
                G.set_option(k, 1)
        err = np.abs(got[:, rows] - exact)
        rel = err / (scale + 1e-12)
        print(f"{M}x{K} N={N} {name} ({'+'.join(ran)}): max {rel.max():.2e}  rms {np.sqrt(np.mean(rel ** 2)):.2e} of sum|w||x|")
        assert np.all(err <= 1.1e-3 * scale + 1e-7), (name, float(rel.max()))
        assert float(np.sqrt(np.mean(rel ** 2))) <= 1e-4, name
        assert np.isfinite(got).all()
        outs[name] = (got, ran)
    assert outs["dma_p8"][1] == ["dma_p8"] and outs["w16_128"][1] == ["w16_p8"] and outs["t256"][1] == ["w16_256"]
    # all three kernels perform the same f16 products in the same order (the K split does not depend on the kernel)
    assert np.array_equal(outs["w16_128"][0], outs["dma_p8"][0])
    assert np.array_equal(outs["t256"][0], outs["w16_128"][0])
    assert np.array_equal(outs["default"][0], outs["w16_128"][0])


def _two_layer_7b(synth, wtype):
    hp0 = dict(synth.LLAMA_7B)
    hp0["n_layer"] = 2
    return synth.make_llama_gaussian(hp0, wtype)


def test_c3_prefill_512_tokens_two_full_size_layers_match_the_oracle(G, O):
    from llm_amd import llama, synth
    hp, w = _two_layer_7b(synth, Q4_0)
    N, ctx = 512, 1024
    toks = np.random.default_rng(42).integers(0, hp["n_vocab"], N).astype(np.int32)
    orc = O.Llama(hp, w, ctx)
    ref = orc.evaluate(toks, mode=O.ref_mode())
    rev = O.Llama(hp, w, ctx).evaluate(toks, mode=O.ref_mode(), reverse_blocks=True)
    mth = O.Llama(hp, w, ctx).evaluate(toks, mode=1)
    std = float(mth.std())
    band, band_rms = float(np.max(np.abs(ref - rev))) / std, float(np.sqrt(np.mean((ref - rev) ** 2))) / std
    floor, floor_rms = float(np.max(np.abs(ref - mth))) / std, float(np.sqrt(np.mean((ref - mth) ** 2))) / std
    print(f"7B x2 layers, N=512: oracle fwd-vs-rev band max {band:.2e} rms {band_rms:.2e}; exact-vs-math floor max {floor:.2e} rms {floor_rms:.2e}")
    model = llama.Llama(hp, w, context_size=ctx)
    res = {}
    for name, opts in (("default", {}), ("t256", {"mmq_t256": 2}), ("w16_128", {"mmq_t256": 0}), ("dma_p8", {"mmq_w16": 0})):
        for k, v in opts.items():
            G.set_option(k, v)
        try:
            sess = model.start_session(n_batch=N)
            p0 = _stat(G, "prompt_plan_tokens")
            got = sess.evaluate(toks)
            assert _stat(G, "prompt_plan_tokens") - p0 == N
            k_, v_ = sess.get_kv()
            sess.free()
        finally:
            for k in opts:
                G.set_option(k, 1)
        d = float(np.max(np.abs(got - ref))) / std
        rms = float(np.sqrt(np.mean((got - ref) ** 2))) / std
        agree = float(np.mean(np.argmax(got, -1) == np.argmax(ref, -1)))
        kd = float(np.max(np.abs(k_.view(np.float16).astype(np.float32) - orc.memory_k.view(np.float16).astype(np.float32))))
        print(f"7B x2 layers, N=512, {name}: max {d:.2e} std, rms {rms:.2e} std, argmax agreement {agree:.3f}, max |dK| {kd:.2e}")
        assert d <= 2 * band + EDGE and rms <= 2 * band_rms + 2e-3, (name, d, rms, band, band_rms)
        assert d <= max(floor, EDGE) * 1.5 and rms <= floor_rms, (name, d, rms, floor, floor_rms)
        res[name] = got
    assert np.array_equal(res["w16_128"], res["dma_p8"])
    assert np.array_equal(res["t256"], res["w16_128"]) and np.array_equal(res["default"], res["w16_128"])
    model.free()


def test_c3_decode_full_size_7b_gaussian_at_the_bench_operating_point(G, O):
    """configs[1] as bench.py times it: gaussian weights, ctx 2048, n_past >= 133.  The device feeds the 133-token prompt;
    the oracle takes over the session's K/V and evaluates the same next tokens."""
    from llm_amd import llama, synth
    hp, w = synth.make_llama_gaussian(synth.LLAMA_7B, Q4_0)
    ctx = 2048
    model = llama.Llama(hp, w, context_size=ctx)
    sess = model.start_session(n_batch=8)
    prompt = np.random.default_rng(42).integers(0, hp["n_vocab"], 133).astype(np.int32)
    sess.feed_prompt(prompt)
    orc, orc_r, orc_m = (O.Llama(hp, w, ctx) for _ in range(3))
    p0 = _stat(G, "plan_tokens")
    for step in range(3):
        k, v = sess.get_kv()
        for o in (orc, orc_r, orc_m):
            o.memory_k[:] = k
            o.memory_v[:] = v
            o.n_past = sess.n_past
        tok = np.array([int(np.argmax(sess.last_logits()))], np.int32)
        assert sess.infer_next_token() == int(tok[0])
        got = sess.last_logits()
        ref = orc.evaluate(tok, mode=O.ref_mode())[-1]
        rev = orc_r.evaluate(tok, mode=O.ref_mode(), reverse_blocks=True)[-1]
        mth = orc_m.evaluate(tok, mode=1)[-1]
        std = float(mth.std())
        d = float(np.max(np.abs(got - ref))) / std
        band, floor = float(np.max(np.abs(ref - rev))) / std, float(np.max(np.abs(ref - mth))) / std
        print(f"7B gaussian decode at n_past {orc.n_past - 1}: gpu-vs-exact {d:.2e} std, oracle fwd-vs-rev band {band:.2e}, "
              f"exact-vs-math floor {floor:.2e}, argmax {int(np.argmax(got))} vs {int(np.argmax(ref))}")
        assert d <= 2 * band + 1e-5 and d <= max(floor, 1e-5), (d, band, floor)
        if d <= 1e-3:
            assert int(np.argmax(got)) == int(np.argmax(ref))
        # the new token's K/V rows of LAYER 0 depend on the embedding and wk / wv only: f16 roundings of mat-vec sums that differ
        # in f32 summation order at most.  (Deeper layers of this random-init model amplify a last-bit difference chaotically —
        # that is what `band` measures — so their rows differ in most halves for ANY two legal implementations.)
        k2, v2 = sess.get_kv()
        C, Eg = ctx, hp["n_embd"]
        pos = orc.n_past - 1
        krow_g, krow_o = k2[pos * Eg:(pos + 1) * Eg], orc.memory_k[pos * Eg:(pos + 1) * Eg]
        vcol_g, vcol_o = v2[:C * Eg].reshape(Eg, C)[:, pos], orc.memory_v[:C * Eg].reshape(Eg, C)[:, pos]
        nk = int(np.count_nonzero(krow_g != krow_o)) + int(np.count_nonzero(vcol_g != vcol_o))
        print(f"   layer-0 K/V halves of the new token that differ from the oracle's: {nk} of {2 * Eg}")
        assert nk <= 0.01 * 2 * Eg
    assert _stat(G, "plan_tokens") - p0 == 3
    sess.free()
    model.free()
