"""k_mmq_w16_256 (kernels/mmq_w16_256.h): the 256 x 256 x 64 prompt GEMM, forced (option mmq_t256 = 2) onto shapes that
exercise its edges — partial tiles in both dimensions, one / two / many k-stages per item (the DMA cursor crossing item
boundaries, the re-requested last stage), K splits with partial tiles and with atomics, several items per workgroup —
held BIT-EXACT to k_mmq_w16_p8 (same f16 values, same k order, same MFMA per k step, same K split) and to the oracle
within the f16 GEMM's bound."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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


SHAPES = [  # (M, K, N)
    (256, 64, 256),      # one tile, ONE k-stage
    (256, 128, 256),     # two stages
    (256, 192, 64),      # three stages, a quarter token tile
    (300, 256, 300),     # partial tiles both ways: 2 x 2 tiles, 3 of them ragged
    (130, 4096, 70),     # 64 stages (K split in two on its own), one ragged tile
    (1000, 1024, 513),   # 4 x 3 tiles
    (70000, 128, 300),   # 274 x 2 tiles = 548 items: more than two per workgroup, short items (cursor crosses items often)
    (2048, 2048, 512),   # 8 x 2 tiles, K split in two
]


@pytest.mark.parametrize("wtype", [2, 7])
@pytest.mark.parametrize("shape", SHAPES)
def test_mmq256_bit_identical_to_the_128_tile_kernel_and_close_to_the_oracle(G, O, wtype, shape):
    M, K, N = shape
    rng = np.random.default_rng([wtype, M, K, N])
    W = (0.02 * rng.standard_normal((M, K))).astype(np.float32)
    X = rng.standard_normal((N, K)).astype(np.float32)
    X[:, ::5] *= 3.0
    W_raw = G.quantize(wtype, W)
    outs = {}
    for name, v in (("t256", 2), ("t128", 0)):
        G.set_option("mmq_t256", v)
        try:
            c0 = {k: _stat(G, "mmq_launches_" + k) for k in ("w16_256", "w16_p8")}
            outs[name] = _mul_mat_gpu(G, wtype, W_raw, M, K, X)
            ran = [k for k in c0 if _stat(G, "mmq_launches_" + k) > c0[k]]
        finally:
            G.set_option("mmq_t256", 1)
        assert ran == (["w16_256"] if v else ["w16_p8"]), (name, ran)
    assert np.array_equal(outs["t256"], outs["t128"]), float(np.max(np.abs(outs["t256"] - outs["t128"])))
    rows = np.unique(np.concatenate([[0, M - 1], rng.choice(M, min(M, 96), replace=False)]))
    rb = O.row_bytes(wtype, K)
    sub = np.concatenate([W_raw[r * rb:(r + 1) * rb] for r in rows])
    exact = O.mul_mat(wtype, sub, len(rows), K, X, mode=O.ref_mode())
    Wd = np.stack([O.dequantize(wtype, sub[i * rb:(i + 1) * rb], K) for i in range(len(rows))])
    scale = np.abs(X) @ np.abs(Wd).T
    err = np.abs(outs["t256"][:, rows] - exact)
    assert np.all(err <= 1.1e-3 * scale + 1e-7), float(np.max(err / (scale + 1e-12)))


def test_mmq256_prompt_plan_forced_is_bit_identical(G):
    """The fused prompt plan with every GEMM forced onto the 256-tile kernel (fused wq|wk|wv and w1|w3 launches: three /
    two weight matrices per launch, K splits stored as partial tiles) against the same plan on the 128-tile kernel."""
    from llm_amd import llama, synth
    hp0 = dict(n_vocab=512, n_embd=1024, n_head=8, n_head_kv=4, n_layer=2, n_rot=128, n_ff=2816, n_mult=32)
    hp, w = synth.make_llama(hp0, 2, seed=11)
    model = llama.Llama(hp, w, context_size=512)
    toks = np.random.default_rng(5).integers(0, hp["n_vocab"], 400).astype(np.int32)
    res = {}
    for v in (2, 0):
        G.set_option("mmq_t256", v)
        try:
            sess = model.start_session(n_batch=256)
            c0 = _stat(G, "mmq_launches_w16_256")
            a = sess.evaluate(toks[:256])
            b = sess.evaluate(toks[256:400])  # 144 tokens at n_past 256
            used = _stat(G, "mmq_launches_w16_256") - c0
            k, vv = sess.get_kv()
            sess.free()
        finally:
            G.set_option("mmq_t256", 1)
        assert (used > 0) == (v == 2)
        res[v] = (a, b, k, vv)
    for x, y in zip(res[2], res[0]):
        assert np.array_equal(x, y)
    model.free()
