"""GPU parity at the shapes of BASELINE.json's configs: the decode kernels are shape-
specialised in places (blocks per row per lane, staging passes, heads per model), so the 13B- and 65B-shaped
layers and the full-size 7B model are run against the CPU oracle, not only the 128-wide test model.
Weights: llm_synth_blocks (random valid GGML blocks, the bench's generator).  The oracle and the GPU run on the same
bytes and the same K/V state.  Tolerance: STRICT = 1e-5·std when no int8 activation quant sits on a rounding edge;
otherwise the yardstick is the reference's OWN ambiguity — the oracle evaluated with its f32 block sums in forward
and in reverse order (two legal orders of ggml's vec_dot) differ by `band` (3e-3 for Q4_0, 3e-2..6e-2 for the
random Q5_1 / Q8_0 blocks of llm_synth_blocks at these widths, printed) — and the GPU must stay within
2·band + STRICT of the forward-order oracle and below the exact-vs-math noise floor."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STRICT = 1e-5


def _run(G, O, hp0, wtype, n_prompt, n_decode, ctx=64):
    from llm_amd import llama, synth
    hp, w = synth.make_llama_fast(hp0, wtype)
    model = llama.Llama(hp, w, context_size=ctx)
    sess = model.start_session(n_batch=8)
    orc, orc_r, orc_m = (O.Llama(hp, w, ctx) for _ in range(3))
    toks = np.random.default_rng(9).integers(0, hp["n_vocab"], n_prompt + n_decode).astype(np.int32)
    p0 = int(G.get_stat("plan_tokens"))
    worst, n_strict, n = 0.0, 0, 0
    for chunk in (toks[:n_prompt],) + tuple(toks[n_prompt + i:n_prompt + i + 1] for i in range(n_decode)):
        got = sess.evaluate(chunk)
        ref = orc.evaluate(chunk, mode=O.ref_mode())
        rev = orc_r.evaluate(chunk, mode=O.ref_mode(), reverse_blocks=True)
        mth = orc_m.evaluate(chunk, mode=1)
        k, v = sess.get_kv()
        for o in (orc, orc_r, orc_m):
            o.memory_k[:] = k
            o.memory_v[:] = v
        std = float(mth.std())
        d = float(np.max(np.abs(got - ref))) / std
        band = float(np.max(np.abs(ref - rev))) / std
        floor = float(np.max(np.abs(ref - mth))) / std
        print(f"N={len(chunk)}: gpu-vs-exact {d:.2e}  oracle fwd-vs-rev band {band:.2e}  exact-vs-math {floor:.2e}")
        worst = max(worst, d)
        n += 1
        n_strict += d <= STRICT
        assert d <= 2 * band + STRICT and d <= max(floor, STRICT), (len(chunk), d, band, floor)
        if d <= STRICT:
            assert (np.argmax(got, -1) == np.argmax(ref, -1)).all()
    # the decode steps ran on the fused plan; so did the prompt chunk where 8 Q8 columns of the widest row fit LDS
    nbp = (max(hp["n_embd"], hp["n_ff"]) // 32 + 63) // 64 * 64
    multi = 8 * nbp * 40 <= 150 * 1024
    assert int(G.get_stat("plan_tokens")) - p0 == n_decode + (n_prompt if multi else 0)
    sess.free()
    model.free()
    return worst, n_strict, n


def test_13b_shaped_layers_q5_1(G, O):
    """C4: LLaMA-13B dims (E 5120, 40 heads, F 13824), Q5_1, 2 layers, small vocabulary."""
    hp = dict(n_vocab=512, n_embd=5120, n_head=40, n_head_kv=40, n_layer=2, n_rot=128, n_ff=13824, n_mult=256)
    worst, ns, n = _run(G, O, hp, G.TYPE_Q5_1, 6, 3)
    print(f"13B-shaped Q5_1: worst {worst:.2e}, {ns}/{n} strict")


def test_65b_shaped_layer_q8_0(G, O):
    """C5: LLaMA-65B dims (E 8192, 64 heads, F 22016), Q8_0, 1 layer, small vocabulary."""
    hp = dict(n_vocab=512, n_embd=8192, n_head=64, n_head_kv=64, n_layer=1, n_rot=128, n_ff=22016, n_mult=256)
    worst, ns, n = _run(G, O, hp, G.TYPE_Q8_0, 5, 3)
    print(f"65B-shaped Q8_0: worst {worst:.2e}, {ns}/{n} strict")


def test_full_size_7b_q4_0_decode_matches_oracle(G, O):
    """C2 at full size: LLaMA-7B Q4_0 (32 layers, 32000 vocabulary, 3.7 GB of blocks): a 4-token prompt and two decode
    steps against the oracle."""
    from llm_amd import synth
    worst, ns, n = _run(G, O, synth.LLAMA_7B, G.TYPE_Q4_0, 4, 2, ctx=32)
    print(f"7B Q4_0 full size: worst {worst:.2e}, {ns}/{n} strict")

