"""GPU parity: every op of the LLaMA hot path, HIP kernel (through the C ABI) vs the CPU oracle on the
same seeded inputs.  Method follows the reference's own integration tests (binaries/llm-test:
deterministic inputs, compare outputs), applied op by op because no model file is obtainable offline.

Tolerances (stated per test): integer block dots are exact, so quantized mat-vec differs from the
oracle's "exact" (ggml scalar semantics) mode only by f32 summation order across blocks:
|Δ| <= 2e-5 * Σ_blocks |term|  (we bound with rtol on the row's absolute-sum scale).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

QTYPES = [2, 3, 6, 7, 8]  # q4_0 q4_1 q5_0 q5_1 q8_0


def _mul_mat_gpu(G, wtype, W_raw, M, K, X, transform=True):
    """dst = mul_mat(W [K,M] wtype, X [K,N] f32) through ggml_graph_compute."""
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


def _abs_scale(O, wtype, W_raw, M, K, X):
    Wd = np.stack([O.dequantize(wtype, W_raw[m * O.row_bytes(wtype, K):(m + 1) * O.row_bytes(wtype, K)], K)
                   for m in range(M)])
    return np.abs(X) @ np.abs(Wd).T  # [N, M]


@pytest.mark.parametrize("wtype", QTYPES)
@pytest.mark.parametrize("shape", [(64, 64), (96, 256), (257, 1024), (33, 4096), (352, 128), (128, 352), (5, 32)])
@pytest.mark.parametrize("N", [1, 2, 3, 5, 8, 13])
def test_mul_mat_q_matches_oracle_exact(G, O, wtype, shape, N):
    M, K = shape
    rng = np.random.default_rng([wtype, M, K, N])
    W = (0.02 * rng.standard_normal((M, K))).astype(np.float32)
    X = rng.standard_normal((N, K)).astype(np.float32)
    W_raw = G.quantize(wtype, W)
    assert (W_raw == O.quantize(wtype, W)).all()
    got = _mul_mat_gpu(G, wtype, W_raw, M, K, X)
    exact = O.mul_mat(wtype, W_raw, M, K, X, mode=O.ref_mode())
    math = O.mul_mat(wtype, W_raw, M, K, X, mode=1)
    scale = _abs_scale(O, wtype, W_raw, M, K, X)
    # vs ggml-exact semantics: f32 summation-order noise only
    assert np.all(np.abs(got - exact) <= 2e-5 * scale + 1e-7), float(np.max(np.abs(got - exact) / (scale + 1e-12)))
    # vs the math yardstick: activation-quantization noise (Q8: <= 1/254 of |x|max per element)
    assert np.all(np.abs(got - math) <= 1.2e-2 * scale + 1e-6)


@pytest.mark.parametrize("wtype", QTYPES)
@pytest.mark.parametrize("shape", [(128, 64), (200, 96), (256, 4096), (384, 352), (130, 1024)])
@pytest.mark.parametrize("N", [32, 33, 100, 128, 300])
@pytest.mark.parametrize("i8", [0, 1])
def test_mul_mat_q_prefill_mfma_matches_oracle(G, O, wtype, shape, N, i8):
    """N >= 32 tokens on the matrix cores, both prompt GEMMs.
    mmq_i8 = 1 and K/32 even: the INTEGER GEMM (kernels/mmq_i8.h, v_mfma_i32_32x32x32_i8): ggml's exact block dots,
    scaled and accumulated in f32 — the mat-vec bound, 2e-5 * scale.
    mmq_i8 = 0 (the default: it is the faster one) or K/32 odd: the f16 GEMM (kernels/mmq_w16*.h, mmq_dmap8.h; mmq_plain.h for an odd K/32),
    which rounds each dequantized weight and activation to f16 (unit roundoff 2^-11 each):
        |got - exact| <= 2 * 2^-11 * sum_k |w||x|   (worst case; stated bound 1.1e-3 * scale), RMS <= 1e-4 * scale."""
    M, K = shape
    rng = np.random.default_rng([wtype, M, K, N, 1])
    W = (0.02 * rng.standard_normal((M, K))).astype(np.float32)
    X = rng.standard_normal((N, K)).astype(np.float32)
    X[:, ::7] *= 4.0
    W_raw = G.quantize(wtype, W)
    G.set_option("mmq_i8", i8)
    try:
        got = _mul_mat_gpu(G, wtype, W_raw, M, K, X)
    finally:
        G.set_option("mmq_i8", 0)
    exact = O.mul_mat(wtype, W_raw, M, K, X, mode=O.ref_mode())
    scale = _abs_scale(O, wtype, W_raw, M, K, X)
    err = np.abs(got - exact)
    if i8 and (K // 32) % 2 == 0:
        assert np.all(err <= 2e-5 * scale + 1e-7), float(np.max(err / (scale + 1e-12)))
    else:
        assert np.all(err <= 1.1e-3 * scale + 1e-7), float(np.max(err / (scale + 1e-12)))
        assert float(np.sqrt(np.mean((err / (scale + 1e-12)) ** 2))) <= 1e-4
    # the mat-vec path on the same inputs (mmq_min = 0 disables the GEMM) agrees to the tight bound
    G.set_option("mmq_min", 0)
    try:
        mv = _mul_mat_gpu(G, wtype, W_raw, M, K, X)
    finally:
        G.set_option("mmq_min", 32)
    assert np.all(np.abs(mv - exact) <= 2e-5 * scale + 1e-7)
    assert not np.array_equal(mv, got)  # the two paths really are different kernels


def test_mul_mat_q_prefill_mfma_is_used_and_handles_extremes(G, O):
    """timing class MMQ_MFMA records the launch; zero rows/columns stay exactly zero; a huge activation stays finite
    (the integer GEMM, option mmq_i8 = 1, scales in f32; the default f16 GEMM clamps to the f16 range) — both checked."""
    M, K, N = 256, 512, 64
    rng = np.random.default_rng(11)
    W = (0.02 * rng.standard_normal((M, K))).astype(np.float32)
    W[5] = 0.0
    X = rng.standard_normal((N, K)).astype(np.float32)
    X[3] = 0.0
    X[4, 9] = 1e6
    W_raw = G.quantize(2, W)
    G.set_option("mmq_i8", 1)
    try:
        G.lib().ggml_hip_timing_begin()
        got = _mul_mat_gpu(G, 2, W_raw, M, K, X)
        G.lib().ggml_hip_timing_end()
    finally:
        G.set_option("mmq_i8", 0)
    ms, launches, flops = G.timing_query(G.KCLASS_MMQ_MFMA)
    assert launches == 1 and flops == 2.0 * M * N * K and ms > 0
    assert np.all(got[3] == 0.0) and np.all(got[:, 5] == 0.0)
    assert np.isfinite(got).all()
    exact = O.mul_mat(2, W_raw, M, K, X, mode=O.ref_mode())
    assert np.allclose(got, exact, rtol=2e-5, atol=2e-5 * float(np.abs(exact).max()))
    G.lib().ggml_hip_timing_begin()
    got16 = _mul_mat_gpu(G, 2, W_raw, M, K, X)
    G.lib().ggml_hip_timing_end()
    ms, launches, flops = G.timing_query(G.KCLASS_MMQ_MFMA)
    assert launches == 1 and flops == 2.0 * M * N * K and ms > 0
    assert np.isfinite(got16).all() and np.all(got16[3] == 0.0) and np.all(got16[:, 5] == 0.0)
    ok = np.ones(N, bool)
    ok[4] = False  # the f16 kernel's clamped row differs from the reference by construction
    assert np.allclose(got16[ok], exact[ok], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("wtype", QTYPES)
def test_mul_mat_q_raw_layout_operand(G, O, wtype):
    """A quantized weight that was never handed to transform_tensor (lives in the compute context) is
    re-laid-out on the fly: same numbers."""
    M, K, N = 48, 512, 2
    rng = np.random.default_rng([wtype, 77])
    W = (0.02 * rng.standard_normal((M, K))).astype(np.float32)
    X = rng.standard_normal((N, K)).astype(np.float32)
    W_raw = G.quantize(wtype, W)
    a = _mul_mat_gpu(G, wtype, W_raw, M, K, X, transform=True)
    b = _mul_mat_gpu(G, wtype, W_raw, M, K, X, transform=False)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("wtype", [2, 8])
def test_mul_mat_q_edge_blocks(G, O, wtype):
    """all-zero activations (d == 0 → id = 0), all-zero weights, and a huge-magnitude column."""
    M, K = 16, 256
    rng = np.random.default_rng(5)
    W = (0.02 * rng.standard_normal((M, K))).astype(np.float32)
    W[3] = 0.0
    X = rng.standard_normal((3, K)).astype(np.float32)
    X[0] = 0.0
    X[1, 17] = 1e4
    W_raw = G.quantize(wtype, W)
    got = _mul_mat_gpu(G, wtype, W_raw, M, K, X)
    exact = O.mul_mat(wtype, W_raw, M, K, X, mode=O.ref_mode())
    assert np.all(got[0] == 0.0) and np.all(got[:, 3] == 0.0)
    assert np.allclose(got, exact, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(128, 4), (4096, 1), (11008, 3), (100, 7)])
def test_rms_norm_and_weight(G, O, shape):
    E, N = shape
    rng = np.random.default_rng([E, N])
    x = rng.standard_normal((N, E)).astype(np.float32) * 3
    w = (1 + 0.01 * rng.standard_normal(E)).astype(np.float32)
    for fuse in (0, 1):
        G.lib().ggml_hip_set_option(b"fuse", fuse)
        with G.Context(x.nbytes * 4 + (1 << 20)) as ctx:
            tx, tw = ctx.tensor_from(x), ctx.tensor_from(w)
            y = ctx.op_mul(ctx.op_rms_norm(tx, 5e-6), tw)
            ctx.graph().build_forward_expand(y).compute()
            got = y.read_data().reshape(N, E)
        ref = O.rms_norm(x, 5e-6) * w
        assert np.allclose(got, ref, rtol=3e-7, atol=1e-7), np.max(np.abs(got - ref))
    G.lib().ggml_hip_set_option(b"fuse", 1)


@pytest.mark.parametrize("n_past", [0, 1, 37, 2047])
@pytest.mark.parametrize("N", [1, 5])
def test_rope(G, O, n_past, N):
    H, D = 4, 128
    rng = np.random.default_rng([n_past, N])
    x = rng.standard_normal((N, H, D)).astype(np.float32)
    with G.Context(x.nbytes * 4 + (1 << 20)) as ctx:
        tx = ctx.tensor_from(x, G.TYPE_F32, (D, H, N))
        y = ctx.op_rope_inplace(tx, n_past, D, 0, 0)
        # rope_inplace returns a view of tx: route it through a cpy so a real node output exists on host
        out = ctx.op_cont(y)
        ctx.graph().build_forward_expand(out).compute()
        got = out.read_data().reshape(N, H, D)
    ref = O.rope(x, n_past, D)
    # sinf/cosf of the device libm vs glibc: a few ulp at |theta| up to 2e3
    assert np.allclose(got, ref, rtol=0, atol=2e-5 * np.abs(x).max()), np.max(np.abs(got - ref))


def test_rope_custom_freq(G, O):
    H, D, N, n_past = 2, 64, 3, 100
    x = np.random.default_rng(3).standard_normal((N, H, D)).astype(np.float32)
    with G.Context(1 << 20) as ctx:
        tx = ctx.tensor_from(x, G.TYPE_F32, (D, H, N))
        out = ctx.op_cont(ctx.op_rope_custom_inplace(tx, n_past, D, 0, 1, 26000.0, 0.5))
        ctx.graph().build_forward_expand(out).compute()
        got = out.read_data().reshape(N, H, D)
    ref = O.rope(x, n_past, D, 26000.0, 0.5)
    assert np.allclose(got, ref, atol=2e-5 * np.abs(x).max())


@pytest.mark.parametrize("fuse", [0, 1])
@pytest.mark.parametrize("n_past,N", [(0, 1), (0, 7), (128, 1), (300, 4), (2047, 1)])
def test_scale_mask_softmax(G, O, fuse, n_past, N):
    H = 4
    T = n_past + N
    rng = np.random.default_rng([n_past, N])
    x = (rng.standard_normal((H, N, T)) * 4).astype(np.float32)
    scale = 1.0 / np.sqrt(np.float32(128.0))
    G.lib().ggml_hip_set_option(b"fuse", fuse)
    with G.Context(x.nbytes * 4 + (1 << 20)) as ctx:
        tx = ctx.tensor_from(x, G.TYPE_F32, (T, N, H))
        src = ctx.op_cont(tx)  # a node (not a leaf) so the in-place chain has a producer, as KQ does
        s = ctx.new_f32(float(scale))
        y = ctx.op_soft_max_inplace(ctx.op_diag_mask_inf_inplace(ctx.op_scale_inplace(src, s), n_past))
        ctx.graph().build_forward_expand(y).compute()
        got = y.read_data().reshape(H, N, T)
    G.lib().ggml_hip_set_option(b"fuse", 1)
    ref = O.scale_mask_softmax(x, float(scale), n_past, mode=O.ref_mode())
    # exp goes through f16 on both sides; device expf vs glibc expf may land on the other side of an f16
    # rounding boundary for a few elements: one f16 ulp (2^-11 relative) on those, renormalised.
    assert np.allclose(got, ref, rtol=1.5e-3, atol=1e-7), np.max(np.abs(got - ref))
    assert np.allclose(got.sum(-1), 1.0, atol=1e-5)
    mism = np.mean(np.abs(got - ref) > 1e-6 * np.abs(ref) + 1e-9)
    assert mism < 0.5
    ref_math = O.scale_mask_softmax(x, float(scale), n_past, mode=1)
    assert np.allclose(got, ref_math, rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize("fuse", [0, 1])
def test_silu_mul(G, O, fuse):
    n = 11008 * 3
    rng = np.random.default_rng(11)
    a = (rng.standard_normal(n) * 3).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    G.lib().ggml_hip_set_option(b"fuse", fuse)
    with G.Context(n * 4 * 6 + (1 << 20)) as ctx:
        ta, tb = ctx.tensor_from(a), ctx.tensor_from(b)
        src = ctx.op_cont(ta)
        y = ctx.op_mul(ctx.op_silu(src), tb)
        ctx.graph().build_forward_expand(y).compute()
        got = y.read_data()
    G.lib().ggml_hip_set_option(b"fuse", 1)
    ref = O.silu(a, mode=O.ref_mode()) * b
    bad = np.abs(got - ref) > 1e-6 * np.abs(ref) + 1e-9
    # identical up to f16-boundary flips of the device expf (<0.1% of elements, each one f16 ulp)
    assert bad.mean() < 2e-3, bad.mean()
    assert np.allclose(got, ref, rtol=1.1e-3, atol=1e-6)


def test_add_mul_broadcast_and_repeat(G, O):
    rng = np.random.default_rng(2)
    a = rng.standard_normal((5, 96)).astype(np.float32)
    b = rng.standard_normal((5, 96)).astype(np.float32)
    w = rng.standard_normal(96).astype(np.float32)
    with G.Context(1 << 20) as ctx:
        ta, tb, tw = ctx.tensor_from(a), ctx.tensor_from(b), ctx.tensor_from(w)
        y1 = ctx.op_add(ta, tb)
        y2 = ctx.op_mul(ta, tw)
        y3 = ctx.op_add(ta, tw)
        y4 = ctx.op_repeat(tw, ta)
        g = ctx.graph()
        for y in (y1, y2, y3, y4):
            g.build_forward_expand(y)
        g.compute()
        assert np.array_equal(y1.read_data().reshape(5, 96), a + b)
        assert np.array_equal(y2.read_data().reshape(5, 96), a * w)
        assert np.array_equal(y3.read_data().reshape(5, 96), a + w)
        assert np.array_equal(y4.read_data().reshape(5, 96), np.broadcast_to(w, (5, 96)))


@pytest.mark.parametrize("wtype", [0, 1] + QTYPES)
def test_get_rows(G, O, wtype):
    V, E = 64, 256
    rng = np.random.default_rng([wtype, 9])
    W = (0.02 * rng.standard_normal((V, E))).astype(np.float32)
    raw = G.quantize(wtype, W)
    ids = np.array([3, 0, 63, 3, 17], dtype=np.int32)
    with G.Context(raw.nbytes + (1 << 20)) as ctx:
        tw = ctx.tensor_from(raw, wtype, (E, V))
        if wtype in QTYPES:
            tw.transfer_to_gpu()
        ti = ctx.tensor_from(ids)
        y = ctx.op_get_rows(tw, ti)
        ctx.graph().build_forward_expand(y).compute()
        got = y.read_data().reshape(len(ids), E)
    rb = O.row_bytes(wtype, E)
    ref = np.stack([O.dequantize(wtype, raw[i * rb:(i + 1) * rb], E) for i in ids])
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("dims", [(4, 32, 64, 3, 5), (4, 32, 256, 40, 5), (3, 128, 512, 130, 71)])
def test_kv_store_and_attention_matmuls(G, O, dims):
    """The KV-cache copy (K contiguous f16 run, V scatter-transposed), then K·Q and V·P over the cache,
    wired exactly as crates/models/llama/src/lib.rs:228-307 for one layer.  N = 3: one-wave-per-output kernels;
    N >= 32 (prompt batch): the batched f16 MFMA GEMM of kernels/gemm_f16.h, incl. ragged M/N tiles and a k tail
    (T = 45, 201 is not a multiple of 8), with the cache beyond T holding NaN to prove the tail is masked."""
    H, D, C, N, P = dims
    E = H * D
    T = P + N
    rng = np.random.default_rng(21)
    kc = rng.standard_normal((N, E)).astype(np.float32)
    vc = rng.standard_normal((N, E)).astype(np.float32)
    q = rng.standard_normal((N, H, D)).astype(np.float32)
    k_past = rng.standard_normal((P, E)).astype(np.float16)
    v_past = rng.standard_normal((P, E)).astype(np.float16)
    memk = np.full((C, E), np.nan, np.float16)
    memk[:P] = k_past
    memv = np.full((E, C), np.nan, np.float16)
    memv[:, :P] = v_past.T
    with G.Context(1 << 22) as sctx, G.Context(1 << 26) as ctx:
        mk = sctx.tensor_from(memk.reshape(-1), G.TYPE_F16).set_name("memory_k")
        mv = sctx.tensor_from(memv.reshape(-1), G.TYPE_F16).set_name("memory_v")
        mk.transfer_to_gpu()
        mv.transfer_to_gpu()
        tk = ctx.op_cont(ctx.tensor_from(kc, G.TYPE_F32, (E, N)))
        tv = ctx.op_cont(ctx.tensor_from(vc, G.TYPE_F32, (E, N)))
        tq = ctx.op_cont(ctx.tensor_from(q, G.TYPE_F32, (D, H, N)))
        g = ctx.graph()
        kview = ctx.op_view_1d(mk, N * E, 2 * E * P)
        vview = ctx.op_view_2d(mv, N, E, C * 2, P * 2)
        g.build_forward_expand(ctx.op_cpy(tk, kview))
        g.build_forward_expand(ctx.op_cpy(ctx.op_transpose(tv), vview))
        Q = ctx.op_permute(tq, 0, 2, 1, 3)
        K = ctx.op_permute(ctx.op_reshape_3d(ctx.op_view_1d(mk, T * E, 0), D, H, T), 0, 2, 1, 3)
        KQ = ctx.op_mul_mat(K, Q)
        V = ctx.op_view_3d(mv, T, D, H, C * 2, C * 2 * D, 0)
        probs = ctx.op_soft_max(KQ)
        KQV = ctx.op_mul_mat(V, probs)
        merged = ctx.op_cpy(ctx.op_permute(KQV, 0, 2, 1, 3), ctx.new_tensor(G.TYPE_F32, E, N))
        g.build_forward_expand(merged)
        g.compute()
        got_kq = KQ.read_data().reshape(H, N, T)
        got_merged = merged.device_get().reshape(N, E)
        got_memk = mk.device_get(np.float16).reshape(C, E)
        got_memv = mv.device_get(np.float16).reshape(E, C)
    # cache contents: exact f16 RNE conversion at the right places
    memk[P:T] = kc.astype(np.float16)
    memv[:, P:T] = vc.astype(np.float16).T
    assert np.array_equal(got_memk, memk, equal_nan=True)
    assert np.array_equal(got_memv, memv, equal_nan=True)
    # K·Q with f16-rounded Q (ggml converts src1 to f16), f32 accumulate
    Kf = memk[:T].astype(np.float32).reshape(T, H, D)
    Qf = q.astype(np.float16).astype(np.float32)
    ref_kq = np.einsum("thd,nhd->hnt", Kf.astype(np.float64), Qf.astype(np.float64))
    assert np.allclose(got_kq, ref_kq, rtol=1e-5, atol=1e-4 if D > 32 else 1e-5)
    pr = O.soft_max(got_kq, mode=O.ref_mode())
    Vf = memv[:, :T].astype(np.float64).reshape(H, D, T)
    ref = np.einsum("hdt,hnt->nhd", Vf, pr.astype(np.float16).astype(np.float64)).reshape(N, E)
    assert np.allclose(got_merged, ref, rtol=2e-3, atol=2e-4)


def test_cpu_backend_nodes_are_mirrored_to_host_and_gpu_nodes_are_not(G, O):
    x = np.arange(64, dtype=np.float32)
    with G.Context(1 << 20) as ctx:
        tx = ctx.tensor_from(x)
        a = ctx.op_add(tx, tx)
        a.offload()  # GPU-backend node: result stays on the device
        b = ctx.op_add(a, tx)
        ctx.graph().build_forward_expand(b).compute()
        assert np.array_equal(b.read_data(), 3 * x)
        assert np.array_equal(a.device_get(), 2 * x)
        assert not np.array_equal(a.read_data(), 2 * x)  # host copy untouched (uninitialised arena bytes)


def test_graph_compute_bumps_ggml_perf_counters(G):
    """ggml's tracing hook (crates/ggml/sys/src/lib.rs:253-255 on the tensor, :542-544 on the graph): every
    ggml_graph_compute bumps perf_runs of the graph and of each node; the graph's perf_time_us accumulates the call's
    wall time (upstream moves the time fields only in a GGML_PERF build, the counters always)."""
    x = np.arange(64, dtype=np.float32).reshape(2, 32)
    with G.Context(1 << 20) as ctx:
        tx = ctx.tensor_from(x)
        y = ctx.op_mul(ctx.op_rms_norm(tx, 1e-5), tx)
        gr = ctx.graph().build_forward_expand(y)
        for run in (1, 2, 3):
            gr.compute()
            c = gr.ptr.contents
            assert c.perf_runs == run and c.perf_time_us >= 0
            assert all(c.nodes[i].contents.perf_runs == run for i in range(c.n_nodes))
            assert all(c.leafs[i].contents.perf_runs == 0 for i in range(c.n_leafs))
        assert gr.ptr.contents.perf_time_us > 0
