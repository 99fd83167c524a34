"""k_mmq_cols (kernels/mmq_cols.h): the mat-muls of a 2..8-token prompt chunk on the integer matrix cores
(v_mfma_i32_16x16x64_i8 block dots, ggml's per-block scale formula in f32) against k_mmvq_big8 (the same contract on the
VALU) and against the oracle.  The two kernels add a row's f32 block terms in different orders, so they agree to f32
summation noise (STRICT) unless that noise moves a downstream activation across an int8 rounding edge (EDGE, see
tests/test_llama_gpu.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STRICT, EDGE = 1e-5, 4e-2
WIDE = dict(n_vocab=320, n_embd=256, n_head=4, n_head_kv=4, n_layer=2, n_rot=64, n_ff=512, n_mult=32)
GQA2 = dict(n_vocab=512, n_embd=1024, n_head=8, n_head_kv=4, n_layer=2, n_rot=128, n_ff=2816, n_mult=32)  # K chunks, 3-matrix QKV with narrow K/V


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
            ref = orc.evaluate(c, mode=0)
            std = float(ref.std())
            d_ab = float(np.max(np.abs(a - b))) / std
            d_ref = float(np.max(np.abs(a - ref))) / std
            # against big8: f32 summation order only, unless it moves a downstream activation across an int8 rounding edge
            # on either side (seen: big8 4.6e-3 off where this kernel matches the oracle to 8e-7); against the oracle: the
            # 1024-wide model's own band is ~4e-2 (tests/test_llama_gpu.py TOL_MATH = 6e-2)
            assert d_ab <= 6e-2 and d_ref <= 6e-2, (cfg, wtype, seed, len(c), d_ab, d_ref)
            n_same += d_ab <= 1e-5
            n_all += 1
            n_strict += d_ref <= STRICT
        model.free()
    print(f"{cfg} type {wtype}: {n_strict} of {n_all} chunks within {STRICT} of the oracle")
    if cfg == "wide":  # the 1024-wide model crosses a rounding edge in nearly every chunk: only the bounds above hold there
        assert n_strict >= 0.3 * n_all and n_same >= 0.3 * n_all
