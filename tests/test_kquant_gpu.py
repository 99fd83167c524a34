"""GPU parity of the K-quant path: Q2_K, Q3_K, Q4_K, Q5_K and Q6_K weights — the super-block formats
the reference's bindings name (crates/ggml/sys/src/lib.rs:2977, 3040, 3103-3108, 3166, 3240-3245; activation side
block_q8_K :3303-3307; file types crates/llm-base/src/loader.rs:80-143) — through the C ABI against the oracle's
restatement of k_quants.c (vec_dot_q*_K_q8_K, quantize_row_q8_K, dequantize_row_q*_K).

The library has no K-quant ENCODER (files arrive pre-quantized), so the test weights are encoded by the oracle; what is
under test is everything after that: upload re-layout, the Q8_K activation quantizer, the mat-vec and get_rows.

Tolerances: the integer block sums are exact on both sides; the f32 scaling is summed in a different order
(DPP tree over lanes vs ggml's running sum over super-blocks): |got - exact| <= 2e-5 * sum_k |w_k||x_k|, the same bound as
the other mat-vecs.  get_rows repeats dequantize_row's float operations in the same order: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KTYPES = [10, 11, 12, 13, 14]  # q2_K q3_K q4_K q5_K q6_K


def _mul_mat_gpu(G, wtype, W_raw, M, K, X, transform=True):
    N = X.shape[0]
    mem = W_raw.nbytes + X.nbytes + M * N * 4 + (1 << 20)
    with G.Context(mem) as ctx:
        w = ctx.tensor_from(W_raw, wtype, (K, M)).set_name("w")
        if transform:
            w.transfer_to_gpu()
        x = ctx.tensor_from(X, G.TYPE_F32, (K, N)).set_name("x")
        y = ctx.op_mul_mat(w, x)
        g = ctx.graph().build_forward_expand(y)
        g.compute()
        return y.read_data().reshape(N, M)


def _dequant(O, wtype, W_raw, M, K):
    rb = O.row_bytes(wtype, K)
    return np.stack([O.dequantize(wtype, W_raw[m * rb:(m + 1) * rb], K) for m in range(M)])


def _weights(O, wtype, M, K, rng):
    W = (0.02 * rng.standard_normal((M, K))).astype(np.float32)
    W[:, ::5] *= 3.0  # uneven sub-block ranges: exercises the 6-bit scales / mins and the int8 sub-block scales
    return O.quantize(wtype, W)


@pytest.mark.parametrize("wtype", KTYPES)
@pytest.mark.parametrize("shape", [(64, 256), (33, 512), (257, 1024), (130, 2816), (96, 4096)])
@pytest.mark.parametrize("N", [1, 2, 3, 5, 8, 13])
def test_mul_mat_k_matches_oracle(G, O, wtype, shape, N):
    M, K = shape
    rng = np.random.default_rng([wtype, M, K, N])
    W_raw = _weights(O, wtype, M, K, rng)
    X = rng.standard_normal((N, K)).astype(np.float32)
    X[:, ::7] *= 4.0
    got = _mul_mat_gpu(G, wtype, W_raw, M, K, X)
    exact = O.mul_mat(wtype, W_raw, M, K, X, mode=O.ref_mode())
    scale = np.abs(X) @ np.abs(_dequant(O, wtype, W_raw, M, K)).T
    err = np.abs(got - exact)
    assert np.all(err <= 2e-5 * scale + 1e-7), float(np.max(err / (scale + 1e-12)))
    # and the result is a real dot product, not noise: close to the f64 math of the dequantized weights
    math = X.astype(np.float64) @ _dequant(O, wtype, W_raw, M, K).astype(np.float64).T
    assert np.all(np.abs(got - math) <= 1.2e-2 * scale + 1e-6)


@pytest.mark.parametrize("wtype", KTYPES)
def test_mul_mat_k_on_the_7b_shapes_and_a_prompt_batch(G, O, wtype):
    """One row block of each 7B shape (K = 4096 and K = 11008 = 43 super-blocks, the ragged last step of a wave), N = 1
    and a 40-token batch (five 8-column passes); rows sampled because the oracle is a scalar loop."""
    for M, K in ((512, 4096), (384, 11008)):
        rng = np.random.default_rng([wtype, M, K])
        W_raw = _weights(O, wtype, M, K, rng)
        for N in (1, 40):
            X = rng.standard_normal((N, K)).astype(np.float32)
            got = _mul_mat_gpu(G, wtype, W_raw, M, K, X)
            exact = O.mul_mat(wtype, W_raw, M, K, X, mode=O.ref_mode())
            scale = np.abs(X) @ np.abs(_dequant(O, wtype, W_raw, M, K)).T
            err = np.abs(got - exact)
            assert np.all(err <= 2e-5 * scale + 1e-7), (M, K, N, float(np.max(err / (scale + 1e-12))))


@pytest.mark.parametrize("wtype", KTYPES)
def test_mul_mat_k_edge_blocks(G, O, wtype):
    """All-zero activation super-blocks (d8 = 0), an all-zero weight row, and ties in the Q8_K extreme (+a before -a:
    the first one in index order decides the sign of the scale)."""
    M, K, N = 64, 768, 3
    rng = np.random.default_rng([wtype, 99])
    W = (0.02 * rng.standard_normal((M, K))).astype(np.float32)
    W[7] = 0.0
    W_raw = O.quantize(wtype, W)
    X = rng.standard_normal((N, K)).astype(np.float32)
    X[0, 256:512] = 0.0
    X[1, :256] = np.clip(X[1, :256], -1.0, 1.0)
    X[1, 17], X[1, 200] = 1.5, -1.5
    X[2, 300], X[2, 290] = -2.5, 2.5
    got = _mul_mat_gpu(G, wtype, W_raw, M, K, X)
    exact = O.mul_mat(wtype, W_raw, M, K, X, mode=O.ref_mode())
    scale = np.abs(X) @ np.abs(_dequant(O, wtype, W_raw, M, K)).T
    assert np.all(np.abs(got - exact) <= 2e-5 * scale + 1e-7)
    assert np.all(got[:, 7] == 0.0)


@pytest.mark.parametrize("wtype", KTYPES)
def test_raw_layout_operand_and_get_rows(G, O, wtype):
    """A K-quant weight that was never handed to transform_tensor is re-laid-out on the fly: same numbers; get_rows
    (the embedding lookup of a K-quant file) is dequantize_row bit for bit."""
    M, K = 96, 512
    rng = np.random.default_rng([wtype, 5])
    W_raw = _weights(O, wtype, M, K, rng)
    X = rng.standard_normal((2, K)).astype(np.float32)
    a = _mul_mat_gpu(G, wtype, W_raw, M, K, X, transform=True)
    b = _mul_mat_gpu(G, wtype, W_raw, M, K, X, transform=False)
    assert np.array_equal(a, b)
    ids = np.array([0, 95, 17, 17, 3], dtype=np.int32)
    with G.Context(W_raw.nbytes + (1 << 20)) as ctx:
        w = ctx.tensor_from(W_raw, wtype, (K, M)).set_name("tab")
        w.transfer_to_gpu()
        i = ctx.tensor_from(ids, G.TYPE_I32, (len(ids),))
        y = ctx.op_get_rows(w, i)
        g = ctx.graph().build_forward_expand(y)
        g.compute()
        got = y.read_data().reshape(len(ids), K)
    ref = _dequant(O, wtype, W_raw, M, K)[ids]
    assert np.array_equal(got, ref)


TINY_K = dict(n_vocab=256, n_embd=256, n_head=4, n_head_kv=4, n_layer=2, n_rot=64, n_ff=512, n_mult=32)


@pytest.mark.parametrize("wtype", KTYPES)
def test_llama_with_k_quant_weights_matches_oracle(G, O, wtype):
    """A two-layer LLaMA whose 2-D weights are all of one K type through the session API: prompt chunks and decode
    steps against the oracle on the same K/V state.  Chunks of up to 8 tokens and the single-token steps run on the K plan (the
    statistic says so; tests/test_kquant_plan_gpu.py holds the plan to the node-by-node executor)."""
    from llm_amd import llama, synth
    rng = np.random.default_rng([wtype, 321])
    hp = dict(TINY_K)
    w = {}
    for name, (ne0, ne1) in synth.tensor_shapes(hp).items():
        if ne1 is None:
            w[name] = (1.0 + 0.01 * rng.standard_normal(ne0)).astype(np.float32)
        else:
            w[name] = O.quantize(wtype, (0.02 * rng.standard_normal((ne1, ne0))).astype(np.float32))
    hp["wtype"] = wtype
    model = llama.Llama(hp, w, context_size=64)
    sess = model.start_session(n_batch=8)
    orc = O.Llama(hp, w, 64)
    toks = rng.integers(0, 256, 20).astype(np.int32)
    g0 = int(G.lib().ggml_hip_get_stat(b"kplan_tokens"))
    worst = 0.0
    for lo, hi in ((0, 8), (8, 13), (13, 14), (14, 15), (15, 16), (16, 20)):
        got = sess.evaluate(toks[lo:hi])
        ref = orc.evaluate(toks[lo:hi], mode=O.ref_mode())
        std = float(ref.std())
        worst = max(worst, float(np.max(np.abs(got - ref))) / std)
        k, v = sess.get_kv()
        orc.memory_k[:] = k[:orc.memory_k.size]
        orc.memory_v[:] = v[:orc.memory_v.size]
    print(f"type {wtype}: worst |dlogit|/std = {worst:.2e}")
    assert worst <= 4e-2  # EDGE of test_llama_gpu.py: one rounding-edge flip of a downstream activation quant
    assert int(G.lib().ggml_hip_get_stat(b"kplan_tokens")) - g0 == 20  # every chunk here has <= 8 tokens: all on the K plan
    sess.free()
    model.free()


@pytest.mark.parametrize("wtype", KTYPES)
@pytest.mark.parametrize("shape", [(256, 512, 64), (384, 1024, 200), (1000, 2048, 96), (512, 4096, 512)])
def test_prompt_batch_of_a_k_quant_weight_runs_on_the_f16_gemm(G, O, wtype, shape):
    """64 tokens and more: the weight's resident f16 copy (its dequantized values in f16) times the activations after
    their Q8_K round trip, on the MFMA GEMM of the other formats (mul_mat_k_gemm).  Against the oracle's exact dots the
    difference is the f16 rounding of both operands (2^-11 relative each, random sign) plus f32 accumulation:
    |got - exact| <= 1e-3 * sum|w||x| and 20x below that in the rms; the launch counters say a GEMM kernel ran."""
    M, K, N = shape
    rng = np.random.default_rng([wtype, M, K, N])
    W_raw = _weights(O, wtype, M, K, rng)
    X = rng.standard_normal((N, K)).astype(np.float32)
    X[:, ::7] *= 4.0
    def gemms():
        return sum(int(G.lib().ggml_hip_get_stat(k)) for k in (b"mmq_launches_w16_p8", b"mmq_launches_w16_256"))
    n0, b0 = gemms(), int(G.lib().ggml_hip_get_stat(b"w16_bytes"))
    got = _mul_mat_gpu(G, wtype, W_raw, M, K, X)
    assert gemms() > n0
    rows = rng.choice(M, 48, replace=False)  # the oracle is a scalar loop
    rb = O.row_bytes(wtype, K)
    sub = np.concatenate([W_raw[m * rb:(m + 1) * rb] for m in rows])
    exact = O.mul_mat(wtype, sub, len(rows), K, X, mode=O.ref_mode())
    scale = np.abs(X) @ np.abs(_dequant(O, wtype, sub, len(rows), K)).T
    err = np.abs(got[:, rows] - exact)
    assert np.all(err <= 1e-3 * scale + 1e-6), float(np.max(err / (scale + 1e-12)))
    assert float(np.sqrt(np.mean((err / (scale + 1e-12)) ** 2))) <= 1e-4
    # the copy is a cache entry like the other formats': accounted, and released with the weight
    assert int(G.lib().ggml_hip_get_stat(b"w16_bytes")) == b0
