"""GPU parity of the two steps either side of the path that moved onto the device:
* ggml_hip_quantize / ggml_hip_quantize_resident — ggml_quantize_q4_0 .. q8_0 (crates/ggml/src/lib.rs:419-483, called by
  crates/llm-base/src/quantize.rs:363-379): BYTE-identical to the host functions and to the oracle, histogram included;
* ggml_hip_topk / llm_session_topk — the top-k stage of the sampler chain (crates/llm-base/src/samplers.rs:289-306): the
  same (value, id) pairs a stable descending sort of the logits puts first."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

QTYPES = [2, 3, 6, 7, 8]


def _rows(rng, n_rows, k):
    x = (0.02 * rng.standard_normal((n_rows, k))).astype(np.float32)
    x[0, :32] = 0.0  # an all-zero block: d = 0, id = 0
    x[1, :32] = 0.5  # a constant block: Q4_1/Q5_1 have d = 0
    x[2, 3], x[2, 7] = 0.25, -0.25  # a tie in magnitude: the first one decides the sign of d
    x[2, :32] = np.clip(x[2, :32], -0.25, 0.25)
    x[3, :64] *= 1e4
    x[4, :32] = np.float32(6e4) * np.sign(x[4, :32] + 1e-9)  # d near the f16 range
    return x


@pytest.mark.parametrize("wtype", QTYPES)
def test_device_quantizer_is_byte_identical_to_the_host_quantizer(G, O, wtype):
    rng = np.random.default_rng([wtype, 77])
    x = _rows(rng, 37, 1184)
    host = G.quantize(wtype, x)
    dev, hist = G.quantize_on_device(wtype, x)
    assert np.array_equal(dev, host)
    assert np.array_equal(dev, O.quantize(wtype, x))
    ref_hist = np.zeros(16, dtype=np.int64)
    out = np.zeros_like(host)
    getattr(G.lib(), "ggml_quantize_" + G.TYPE_NAMES[wtype])(x.ctypes.data, out.ctypes.data, x.size, x.shape[1], ref_hist.ctypes.data)
    assert np.array_equal(hist, ref_hist) and hist.sum() == x.size


def test_device_quantizer_of_a_large_tensor(G):
    """A 7B-sized matrix (4096 x 11008 f32, 180 MB) in one call: same bytes as the threaded host quantizer."""
    rng = np.random.default_rng(5)
    x = (0.02 * rng.standard_normal((4096, 11008), dtype=np.float32))
    dev, hist = G.quantize_on_device(G.TYPE_Q4_0, x)
    assert np.array_equal(dev, G.quantize(G.TYPE_Q4_0, x))
    assert hist.sum() == x.size


@pytest.mark.parametrize("wtype", QTYPES)
@pytest.mark.parametrize("src_f16", [False, True])
def test_quantize_resident_matrix_gives_the_weight_the_host_path_uploads(G, O, wtype, src_f16):
    """f32 / f16 matrix already in HBM -> quantized weight in HBM, no PCIe traffic: get_rows of every row (the dequantized
    blocks) and a mat-vec are bit-identical to the same operations on host-quantized, uploaded blocks."""
    M, K = 48, 256
    rng = np.random.default_rng([wtype, int(src_f16), 3])
    W = _rows(rng, M, K)
    if src_f16:
        W = np.clip(W, -6e4, 6e4).astype(np.float16)
    Wf = W.astype(np.float32)
    X = rng.standard_normal((2, K)).astype(np.float32)
    ids = np.arange(M, dtype=np.int32)

    def run(make_weight):
        with G.Context(16 << 20) as ctx:
            w = make_weight(ctx)
            x = ctx.tensor_from(X, G.TYPE_F32, (K, 2))
            i = ctx.tensor_from(ids, G.TYPE_I32, (M,))
            y = ctx.op_mul_mat(w, x)
            r = ctx.op_get_rows(w, i)
            g = ctx.graph().build_forward_expand(y).build_forward_expand(r)
            g.compute()
            return y.read_data().copy(), r.read_data().copy()

    def from_host(ctx):
        w = ctx.tensor_from(G.quantize(wtype, Wf), wtype, (K, M))
        w.transfer_to_gpu()
        return w

    def on_device(ctx):
        src = ctx.tensor_from(W, G.TYPE_F16 if src_f16 else G.TYPE_F32, (K, M))
        src.transfer_to_gpu()
        dst = ctx.new_tensor(wtype, K, M)
        hist = np.zeros(16, dtype=np.int64)
        assert G.lib().ggml_hip_quantize_resident(src.ptr, dst.ptr, hist.ctypes.data) == 0
        assert hist.sum() == M * K
        ctx._offloaded.append(dst)
        return dst

    y0, r0 = run(from_host)
    y1, r1 = run(on_device)
    assert np.array_equal(r0, r1) and np.array_equal(y0, y1)
    assert np.array_equal(r0.reshape(M, K), np.stack([O.dequantize(wtype, O.quantize(wtype, Wf[m]), K) for m in range(M)]))


def _ref_topk(x, k):
    order = np.lexsort((np.arange(x.size), -x.astype(np.float64)))  # value descending, id ascending
    return x[order[:k]], order[:k].astype(np.int32)


@pytest.mark.parametrize("n,k", [(32000, 40), (50257, 1), (50257, 1024), (4096, 100), (257, 257), (1000, 7)])
def test_topk_of_a_row_matches_a_stable_sort(G, n, k):
    rng = np.random.default_rng([n, k])
    x = rng.standard_normal((3, n)).astype(np.float32)
    x[1] = np.round(x[1] * 4) / 4  # many equal values: ties resolved towards the lower id
    x[2, ::3] = -0.0
    x[2, 1::3] = 0.0
    x[2, 5] = np.inf
    x[2, 6] = -np.inf
    extra = np.array([0, n - 1, 17, 17], dtype=np.int32)
    with G.Context(x.nbytes + (1 << 20)) as ctx:
        a = ctx.tensor_from(x, G.TYPE_F32, (n, 3))
        y = ctx.op_scale(a, ctx.new_f32(1.0)) if hasattr(G.lib(), "ggml_scale") else a
        g = ctx.graph().build_forward_expand(y)
        g.compute()
        for row in range(3):
            vals, ids = G.topk(y, row, k, extra)
            rv, ri = _ref_topk(x[row], k)
            # -0.0 and 0.0 compare equal (as in the host sort): the lower id comes first
            assert np.array_equal(ids[:k], ri), (row, ids[:8], ri[:8])
            assert np.array_equal(vals[:k], rv)
            assert np.array_equal(ids[k:], extra) and np.array_equal(vals[k:], x[row][extra])
    with pytest.raises(ValueError):
        G.topk(y, 0, 0)


def test_session_topk_gives_the_head_of_the_logits(G, O):
    """llm_session_topk after prompt chunks and decode steps (fused plan and generic graphs) = sorting last_logits."""
    from llm_amd import llama, synth
    hp, w = synth.make_llama(synth.TINY, 2, seed=3)
    model = llama.Llama(hp, w, context_size=64)
    sess = model.start_session(n_batch=8)
    toks = np.random.default_rng(9).integers(0, 256, 20).astype(np.int32)
    for lo, hi in ((0, 8), (8, 11), (11, 12), (12, 13)):
        sess.evaluate(toks[lo:hi])
        logits = sess.last_logits()
        vals, ids = sess.top_k(40, extra_ids=toks[:hi][-4:])
        rv, ri = _ref_topk(logits, 40)
        assert np.array_equal(ids[:40], ri) and np.array_equal(vals[:40], rv)
        assert np.array_equal(vals[40:], logits[toks[:hi][-4:]])
    sess.free()
    model.free()


def test_sampled_token_step_draws_the_same_tokens_from_device_topk_and_from_all_logits(G):
    """llm_infer_next_token_topk: the reference's token step with its default sampler's shape (samplers.rs:97-188) — candidates from
    all n_vocab logits on the host (what model/common.rs:6-19 reads back) or from the device's top-k + the penalty window
    (llm_session_topk, the logits never leave HBM): same candidates, same generator, same tokens."""
    import ctypes
    from llm_amd import llama, synth
    hp, w = synth.make_llama(dict(n_vocab=512, n_embd=256, n_head=4, n_head_kv=4, n_layer=2, n_rot=64, n_ff=704, n_mult=32), 2, seed=77)
    model = llama.Llama(hp, w, context_size=128)
    toks = np.random.default_rng(9).integers(0, hp["n_vocab"], 11).astype(np.int32)
    drawn = []
    try:
        for dev in (False, True):
            s = model.start_session(n_batch=8)
            s.set_speculate(False)
            s.feed_prompt(toks)
            rng = ctypes.c_uint64(0x1234567887654321)
            drawn.append([s.infer_next_token_topk(rng, 40, 0.8, dev) for _ in range(40)])
            assert s.n_past == 11 + 40
            s.free()
    finally:
        model.free()
    assert drawn[0] == drawn[1]
    assert len(set(drawn[0])) > 5  # (it does sample: not one token forever)
