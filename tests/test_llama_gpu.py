"""GPU parity of the whole hot path: the LLaMA graph built by the host mirror of
crates/models/llama/src/lib.rs (llm_amd/csrc/host/llm_host.cpp), executed by ggml_graph_compute on the
MI355X, against the CPU oracle on identical synthetic GGML weights.

Method = the reference's own integration tests (binaries/llm-test/src/{inference,tokens,delete}.rs):
 (c) greedy determinism, (d) argmax-token agreement, (e) rewind consistency — plus direct logits
comparison, which the reference cannot do offline.

Stated tolerance on logits (north_star: "within a stated fp tolerance on logits"):
   vs oracle mode 0 (ggml CPU semantics):  max|Δ| <= 2e-3 * std(logits)   (f32 summation order and
                                            libm-vs-device expf/sinf flips at f16 / int8 rounding edges)
   vs oracle mode 1 (f64 math):            max|Δ| <= 6e-2 * std(logits)   (the reference's own
                                            activation-quantization noise; reported as the noise floor)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_EXACT = 2e-3
TOL_MATH = 6e-2


def _mk(G, wtype, hp=None, ctx=64, seed=1234):
    from llm_amd import llama, synth
    hp, w = synth.make_llama(hp or synth.TINY, wtype, seed=seed)
    return hp, w, llama.Llama(hp, w, context_size=ctx)


@pytest.mark.parametrize("seed", [1234, 7])
@pytest.mark.parametrize("wtype", [2, 3, 6, 7, 8])
def test_logits_match_oracle_prompt_and_decode(G, O, wtype, seed):
    """`band` = distance between two legal orders of ggml's own f32 block sum (ascending vs descending) on this
    model: the reference's rounding sensitivity.  It is ~5e-7 except where an activation sits on an int8 / f16
    rounding edge (Q5_0 with seed 1234: 2.3e-2 — one flipped quant in a 128-wide model), so the GPU must be
    within TOL_EXACT + 1.2*band of the ascending-order oracle."""
    hp, w, model = _mk(G, wtype, seed=seed)
    sess = model.start_session(n_batch=8)
    orc0, orc0r, orc1 = O.Llama(hp, w, 64), O.Llama(hp, w, 64), O.Llama(hp, w, 64)
    toks = np.random.default_rng(42).integers(0, hp["n_vocab"], 20).astype(np.int32)
    strict = 0
    # prompt in two batches (N=8, N=5), then 7 single-token decodes (N=1): both mat-vec column paths
    for chunk in (toks[:8], toks[8:13]) + tuple(toks[13 + i:14 + i] for i in range(7)):
        got = sess.evaluate(chunk)
        e0 = orc0.evaluate(chunk, mode=0)
        e0r = orc0r.evaluate(chunk, mode=0, reverse_blocks=True)
        e1 = orc1.evaluate(chunk, mode=1)
        std = float(e1.std())
        d0 = float(np.max(np.abs(got - e0))) / std
        d1 = float(np.max(np.abs(got - e1))) / std
        band = float(np.max(np.abs(e0 - e0r))) / std
        floor = float(np.max(np.abs(e0 - e1))) / std
        print(f"type {wtype} seed {seed} N={len(chunk)} n_past={sess.n_past}: gpu-vs-exact {d0:.2e}  "
              f"exact re-association band {band:.2e}  gpu-vs-math {d1:.2e}  exact-vs-math (noise floor) {floor:.2e}")
        assert d0 <= TOL_EXACT + 1.2 * band, (d0, band, "vs ggml-exact oracle")
        assert d1 <= TOL_MATH, (d1, "vs math oracle")
        if band < 1e-5:
            strict += 1
            assert d0 <= 1e-5  # no rounding edge in play: agreement to f32 summation noise
            assert (np.argmax(got, -1) == np.argmax(e0, -1)).all()  # llm-test `Tokens` check
    # every (type, seed) except the one known rounding-edge case must have been checked strictly
    assert strict >= 1 or (wtype, seed) == (6, 1234), (wtype, seed)
    sess.free()
    model.free()


def test_interior_taps_layer0(G, O):
    """Tensor-by-tensor: embeddings output (final norm) against the oracle tap."""
    hp, w, model = _mk(G, 2)
    sess = model.start_session()
    orc = O.Llama(hp, w, 64)
    toks = np.array([1, 5, 200, 17], np.int32)
    logits, emb = sess.evaluate(toks, want_embeddings=True)
    ref_logits, taps = orc.evaluate(toks, mode=0, taps=True)
    assert np.allclose(emb, taps["final_norm"][-1], rtol=2e-3, atol=2e-3)
    sess.free()
    model.free()


def test_greedy_is_deterministic_and_matches_oracle_tokens(G, O):
    hp, w, model = _mk(G, 2)
    prompt = np.random.default_rng(7).integers(0, hp["n_vocab"], 8).astype(np.int32)
    runs = []
    for _ in range(2):
        s = model.start_session(n_batch=8)
        s.feed_prompt(prompt)
        runs.append([s.infer_next_token() for _ in range(24)])
        s.free()
    assert runs[0] == runs[1]
    orc = O.Llama(hp, w, 64)
    lg = orc.evaluate(prompt, mode=0)[-1]
    ref = []
    for _ in range(24):
        t = int(np.argmax(lg))
        ref.append(t)
        lg = orc.evaluate(np.array([t], np.int32), mode=0)[-1]
    # greedy chains may legitimately fork at a near-tie; require a long common prefix and report it
    common = next((i for i, (a, b) in enumerate(zip(runs[0], ref)) if a != b), len(ref))
    print("greedy tokens equal to the oracle for", common, "of", len(ref))
    assert common >= 12
    model.free()


def test_rewind_then_refeed_reproduces_logits(G, O):
    """binaries/llm-test/src/delete.rs:48-56: logits after rewind(1)+re-feed equal the originals."""
    hp, w, model = _mk(G, 2)
    s = model.start_session()
    toks = np.array([3, 9, 27, 81, 243 % 256, 11], np.int32)
    s.feed_prompt(toks[:5])
    s.evaluate(toks[5:6])
    a = s.last_logits()
    assert s.rewind(1) == 0
    s.evaluate(toks[5:6])
    b = s.last_logits()
    assert np.array_equal(a, b)  # same kernels, same inputs: bitwise
    s.free()
    model.free()


def test_prompt_chunking_invariance(G, O):
    """n_batch=8 vs n_batch=1 feed the same KV cache: last-token logits agree to fp noise."""
    hp, w, model = _mk(G, 8)
    toks = np.random.default_rng(3).integers(0, hp["n_vocab"], 11).astype(np.int32)
    outs = []
    for nb in (8, 1, 4):
        s = model.start_session(n_batch=nb)
        s.feed_prompt(toks)
        outs.append(s.last_logits())
        s.free()
    std = outs[0].std()
    assert np.max(np.abs(outs[0] - outs[1])) <= 2e-3 * std
    assert np.max(np.abs(outs[0] - outs[2])) <= 2e-3 * std
    model.free()


def test_graph_is_the_reference_graph(G, O):
    """Node count of the graph handed to ggml_graph_compute: per layer the Rust builder creates 37 non-leaf
    tensors + get_rows + final rms_norm, mul, mul_mat."""
    hp, w, model = _mk(G, 2)
    s = model.start_session()
    s.evaluate(np.array([1, 2, 3], np.int32))
    n_nodes, n_leafs = s.graph_stats()
    L = hp["n_layer"]
    assert n_nodes == 37 * L + 4, n_nodes
    # leafs: embd + wte + norm + output + per layer (9 weights + kq_scale + merge dst) + memory_k + memory_v
    assert n_leafs == 4 + 11 * L + 2, n_leafs
    s.free()
    model.free()
