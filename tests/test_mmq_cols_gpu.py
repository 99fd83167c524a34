"""k_mmq_cols (kernels/mmq_cols.h): the mat-muls of a 2..8-token prompt chunk on the integer matrix cores
(v_mfma_i32_16x16x64_i8 block dots, ggml's per-block scale formula in f32) against k_mmvq_big8 (the same contract on the
VALU) and against the oracle.  The two kernels add a row's f32 block terms in different orders, so they agree to f32
summation noise (STRICT) unless that noise moves a downstream activation across an int8 rounding edge (EDGE, see
tests/test_llama_gpu.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STRICT, EDGE = 1e-5, 4e-2
WIDE = dict(n_vocab=320, n_embd=1024, n_head=8, n_head_kv=8, n_layer=1, n_rot=128, n_ff=1024, n_mult=32)  # one layer: few rounding edges
GQA2 = dict(n_vocab=512, n_embd=1024, n_head=8, n_head_kv=4, n_layer=2, n_rot=128, n_ff=2816, n_mult=32)  # K chunks, 3-matrix QKV with narrow K/V


def _cols(G, wtype, W_raw, M, K, X):
    """W (quantized rows) x X [N][K] through k_mmq_cols as the multi-token plan launches it (ggml_hip_debug_mul_mat_cols)."""
    N = X.shape[0]
    out = np.zeros((N, M), np.float32)
    with G.Context(W_raw.nbytes + (1 << 20)) as ctx:
        w = ctx.tensor_from(W_raw, wtype, (K, M)).set_name("w")
        w.transfer_to_gpu()
        X = np.ascontiguousarray(X, np.float32)
        rc = G.lib().ggml_hip_debug_mul_mat_cols(w.ptr, X.ctypes.data, out.ctypes.data, N)
    return rc, out


@pytest.mark.parametrize("wtype", [2, 3, 6, 7, 8])
@pytest.mark.parametrize("shape", [(64, 1024), (48, 1280), (4096, 4096), (1000 * 16, 1024), (256, 11008), (272, 2816)])
@pytest.mark.parametrize("N", [2, 3, 4, 5, 8])
def test_cols_mat_mul_matches_the_oracle_at_op_level(G, O, wtype, shape, N):
    """One launch, no re-quantization downstream: the exact integer block dots with ggml's per-block f32 formula, the f32 sum
    over blocks in this kernel's order (eight K ranges, four block groups) -> the mat-vec bound 2e-5 * sum|w||x|.  Shapes: one
    group per workgroup and many, K ranges of uneven length (K = 1280: 10 steps over 8 waves; 11008: 86), more groups than
    12 per workgroup (16000 rows: the grid grows past the CU count), K = 2816."""
    M, K = shape
    rng = np.random.default_rng([wtype, M, K, N])
    W = (0.05 * rng.standard_normal((M, K))).astype(np.float32)
    W_raw = O.quantize(wtype, W)
    X = rng.standard_normal((N, K)).astype(np.float32)
    X[:, ::7] *= 4.0
    rc, got = _cols(G, wtype, W_raw, M, K, X)
    assert rc == 0
    rows = np.arange(M) if M <= 512 else np.sort(rng.choice(M, 256, replace=False))
    rb = O.row_bytes(wtype, K)
    sub = np.concatenate([W_raw[m * rb:(m + 1) * rb] for m in rows])
    exact = O.mul_mat(wtype, sub, len(rows), K, X, mode=O.ref_mode())
    D = np.stack([O.dequantize(wtype, sub[i * rb:(i + 1) * rb], K) for i in range(len(rows))])
    scale = np.abs(X) @ np.abs(D).T
    err = np.abs(got[:, rows] - exact)
    assert np.all(err <= 2e-5 * scale + 1e-7), float(np.max(err / (scale + 1e-12)))


def test_cols_hook_refuses_shapes_the_plan_would_not_run(G, O):
    W_raw = O.quantize(2, np.zeros((64, 256), np.float32))  # K = 256: fewer than 8 steps of 4 blocks
    rc, _ = _cols(G, 2, W_raw, 64, 256, np.zeros((4, 256), np.float32))
    assert rc == -1


def _stat(G, key):
    return int(G.lib().ggml_hip_get_stat(key.encode()))


@pytest.mark.parametrize("wtype", [2, 3, 6, 7, 8])
@pytest.mark.parametrize("cfg", ["wide", "gqa2"])
def test_cols_kernel_matches_big8_and_the_oracle(G, O, wtype, cfg):
    from llm_amd import llama, synth
    hp0 = WIDE if cfg == "wide" else GQA2
    n_strict = n_all = n_same = 0
    for seed in (3, 4):
        hp, w = synth.make_llama(hp0, wtype, seed=seed)
        model = llama.Llama(hp, w, context_size=64)
        toks = np.random.default_rng(seed).integers(0, hp["n_vocab"], 40).astype(np.int32)
        chunks = [toks[0:8], toks[8:10], toks[10:13], toks[13:18], toks[18:25], toks[25:33]]
        outs = {}
        for cols in (1, 0):
            G.set_option("mmq_cols", cols)
            try:
                sess = model.start_session(n_batch=8)
                p0 = _stat(G, "plan_tokens")
                outs[cols] = [sess.evaluate(c) for c in chunks]
                assert _stat(G, "plan_tokens") - p0 == sum(len(c) for c in chunks)
                sess.free()
            finally:
                G.set_option("mmq_cols", 1)
        orc = O.Llama(hp, w, 64)
        for c, a, b in zip(chunks, outs[1], outs[0]):
            ref = orc.evaluate(c, mode=O.ref_mode())
            std = float(ref.std())
            d_ab = float(np.max(np.abs(a - b))) / std
            d_ref = float(np.max(np.abs(a - ref))) / std
            # against big8: f32 summation order only, unless it moves a downstream activation across an int8 rounding edge
            # on either side (seen: big8 4.6e-3 off where this kernel matches the oracle to 8e-7); against the oracle: the
            # 1024-wide model's own band is ~4e-2 (tests/test_llama_gpu.py TOL_MATH = 6e-2)
            assert d_ab <= 1e-1 and d_ref <= 1e-1, (cfg, wtype, seed, len(c), d_ab, d_ref)
            n_same += d_ab <= 1e-5
            n_all += 1
            n_strict += d_ref <= STRICT
        model.free()
    # how often a chunk stays within f32 summation noise of the oracle is reported, not asserted: with 1024-wide rows nearly
    # every chunk moves some activation across an int8 rounding edge; the kernel's own arithmetic is pinned at op level above
    print(f"{cfg} type {wtype}: {n_strict} of {n_all} chunks within {STRICT} of the oracle, {n_same} equal to big8 within 1e-5")


@pytest.mark.parametrize("wtype", [2, 7])
def test_chunks_of_9_to_31_tokens_run_in_passes_and_equal_the_8_token_chunks(G, wtype):
    """A chunk of 9..31 tokens (below the prompt plan's threshold of 32) is the multi-token plan in passes of 8 columns: every
    weight matrix streamed once per pass.  A token's arithmetic does not depend on which other tokens share its chunk (same
    kernels, columns independent), so feeding [31, 9, 16] must reproduce [8, 8, 8, ...] BIT FOR BIT: logits of every token
    and the K/V cache; the counters say the fused plan ran every token."""
    from llm_amd import llama, synth
    hp, w = synth.make_llama(GQA2, wtype, seed=11)
    model = llama.Llama(hp, w, context_size=96)
    toks = np.random.default_rng(9).integers(0, hp["n_vocab"], 56).astype(np.int32)

    def run(n_batch, sizes):
        sess = model.start_session(n_batch=n_batch)
        p0, g0 = _stat(G, "plan_tokens"), _stat(G, "generic_graphs")
        outs, at = [], 0
        for n in sizes:
            outs.append(sess.evaluate(toks[at:at + n]))
            at += n
        assert at == len(toks)
        assert _stat(G, "plan_tokens") - p0 == len(toks) and _stat(G, "generic_graphs") == g0
        k, v = sess.get_kv()
        sess.free()
        return np.concatenate(outs), k, v

    a, ka, va = run(31, [31, 9, 16])
    b, kb, vb = run(8, [8] * 7)  # (a last chunk of ONE token would be the decode plan: another kernel, another sum order)
    assert a.shape == b.shape == (56, hp["n_vocab"])
    assert np.array_equal(a, b)
    assert np.array_equal(ka, kb) and np.array_equal(va, vb)
    model.free()
