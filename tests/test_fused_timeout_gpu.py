"""An in-launch hand-off that never arrives must not produce silent garbage (SURVEY.md section 8b "Errors": no error
returns, abort with a message — the Rust caller ignores ggml_graph_compute's return, crates/ggml/src/lib.rs:374-376).

k_qkv_attn's attention workgroups wait for rows that other workgroups of the same launch publish; if those never run (the
launch is not fully resident: another process on the GPU, a CU mask) the wait gives up after GRAN_SPIN_MAX polls and raises the
plan's error word, which travels back with every token's results (llama_plan.inc token_finish).  Option test_fused_timeout
points layer 0's attention workgroups at granules nobody writes:
  * default (fused_fallback = 1): the token is re-run on the two-launch pair — logits bit-identical to fuse_attn = 0 — the slot
    keeps the pair from then on, and the counter says so;
  * fused_fallback = 0: ggml_graph_compute aborts with the message (checked in a child process)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HP = dict(n_vocab=256, n_embd=512, n_head=8, n_head_kv=8, n_layer=3, n_rot=64, n_ff=704, n_mult=32)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stat(G, key):
    return int(G.lib().ggml_hip_get_stat(key.encode()))


def _run(G, model, toks, n):
    s = model.start_session(n_batch=8)
    s.feed_prompt(toks)
    out = []
    for _ in range(n):
        t = s.infer_next_token()
        out.append((t, s.last_logits()))
    k, v = s.get_kv()
    s.free()
    return out, k, v


def test_a_hand_off_that_never_arrives_is_rerun_on_the_two_launch_pair(G):
    from llm_amd import llama, synth
    hp, w = synth.make_llama(HP, 2, seed=77)
    model = llama.Llama(hp, w, context_size=96)
    toks = np.random.default_rng(4).integers(0, hp["n_vocab"], 11).astype(np.int32)
    try:
        G.set_option("fuse_attn", 0)
        ref, k0, v0 = _run(G, model, toks, 6)
        G.set_option("fuse_attn", 2)
        t0, f0 = _stat(G, "fused_attn_timeouts"), _stat(G, "fused_attn_tokens")
        G.set_option("test_fused_timeout", 1)
        got, k1, v1 = _run(G, model, toks, 6)
        t1, f1 = _stat(G, "fused_attn_timeouts"), _stat(G, "fused_attn_tokens")
    finally:
        G.set_option("test_fused_timeout", 1)  # (the fallback switched the hook off on the slot: on again, so that "off" clears the counter)
        G.set_option("test_fused_timeout", 0)
        G.set_option("fuse_attn", 1)
        G.set_option("attn_one", 1)
        model.free()
    assert _stat(G, "fused_attn_timeouts") == 0  # switching the hook off clears its give-ups: later tests start from zero
    assert t1 - t0 == 1  # the first decode token ran into the dead hand-off once ...
    assert f1 - f0 == 1  # ... and the slot took the two-launch pair for the tokens after it
    for (ta, la), (tb, lb) in zip(ref, got):
        assert ta == tb and np.array_equal(la, lb)  # the re-run token included: nothing of the garbage run survived
    assert np.array_equal(k0, k1) and np.array_equal(v0, v1)


def test_with_the_fallback_off_the_compute_call_aborts_with_a_message():
    code = textwrap.dedent("""
        import numpy as np
        from llm_amd import ggml, llama, synth
        HP = %r
        hp, w = synth.make_llama(HP, 2, seed=77)
        model = llama.Llama(hp, w, context_size=96)
        ggml.set_option("fuse_attn", 2)
        ggml.set_option("fused_fallback", 0)
        ggml.set_option("test_fused_timeout", 1)
        s = model.start_session(n_batch=8)
        s.feed_prompt(np.arange(8, dtype=np.int32))  # one chunk of 8: the multi-token plan, no fused launch yet
        print("BEFORE", flush=True)
        s.infer_next_token()
        print("SURVIVED", flush=True)
    """ % (HP,))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert "BEFORE" in r.stdout and "SURVIVED" not in r.stdout
    assert r.returncode != 0
    assert "gave up waiting for rows of its own launch" in r.stderr


def test_the_fused_forms_come_back_after_a_clean_stretch(G):
    """one give-up used to leave the slot on the two-launch forms for the rest of its life.  After
    fused_rearm_tokens clean tokens (256 by default, 6 here) the fused launch is taken again — and the tokens stay what the
    two-launch pair gives."""
    from llm_amd import llama, synth
    hp, w = synth.make_llama(HP, 2, seed=78)
    model = llama.Llama(hp, w, context_size=96)
    toks = np.random.default_rng(5).integers(0, hp["n_vocab"], 11).astype(np.int32)
    try:
        G.set_option("fuse_attn", 0)
        ref, _, _ = _run(G, model, toks, 24)
        G.set_option("fuse_attn", 2)
        G.set_option("fused_rearm_tokens", 6)
        r0, t0 = _stat(G, "fused_rearms"), _stat(G, "fused_attn_timeouts")
        G.set_option("test_fused_timeout", 1)
        s = model.start_session(n_batch=8)
        s.feed_prompt(toks)
        got, fused_after = [], []
        for i in range(24):
            f0 = _stat(G, "fused_attn_tokens")
            t = s.infer_next_token()
            got.append((t, s.last_logits()))
            fused_after.append(_stat(G, "fused_attn_tokens") - f0)
        s.free()
        assert _stat(G, "fused_attn_timeouts") - t0 == 1  # the first token's hand-off gave up once (the fallback switches the hook off)
        assert _stat(G, "fused_rearms") - r0 == 1
        assert sum(fused_after[1:7]) == 0 and sum(fused_after[10:]) >= 10  # two-launch forms for the stretch, fused again behind it
    finally:
        G.set_option("fused_rearm_tokens", 256)
        G.set_option("test_fused_timeout", 1)
        G.set_option("test_fused_timeout", 0)
        G.set_option("fuse_attn", 1)
        G.set_option("attn_one", 1)
        model.free()
    for (ta, la), (tb, lb) in zip(ref, got):
        assert ta == tb and np.array_equal(la, lb)
