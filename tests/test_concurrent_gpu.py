"""Several InferenceSessions of one process on different threads (crates/llm-base/src/inference_session.rs:43-48: a session is
Send; crates/llm-base/src/model/mod.rs:275-276: a model serves several of them) — on DIFFERENT device slots they must overlap,
not queue behind one library lock: entry points lock the slot they act on (llm_amd/csrc/backend_state.inc SlotLock), the
current slot is per thread, the reference's process-wide ggml_cuda_set_main_device stays the default for threads that never
chose one.  A 1-GPU box has one device, so the second slot is virtual (GGML_HIP_VIRTUAL_DEVICES: own stream, shadows, weight
records and plan cache on the same GPU).  Results must equal the sessions run one after the other, bit for bit."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HP = dict(n_vocab=256, n_embd=256, n_head=8, n_head_kv=8, n_layer=4, n_rot=32, n_ff=704, n_mult=32)


def _stat(G, key):
    return int(G.lib().ggml_hip_get_stat(key.encode()))


def _decode(sess, toks, n):
    sess.feed_prompt(toks)
    out = []
    for _ in range(n):
        t = sess.infer_next_token()
        out.append((t, sess.last_logits()))
    return out


def test_sessions_on_two_slots_overlap_and_match_the_sequential_runs(G):
    from llm_amd import llama, synth
    L = G.lib()
    assert L.ggml_hip_get_main_device() == 0  # every test (and every entry point) leaves the main device as it found it
    hp, w = synth.make_llama(HP, 2, seed=23)
    os.environ["GGML_HIP_VIRTUAL_DEVICES"] = "2"
    models = []
    try:
        assert L.ggml_hip_device_count() >= 2
        for slot in (0, 1):
            L.ggml_hip_set_main_device(slot)  # the reference's hook: the slot the model's weights are uploaded to
            models.append(llama.Llama(hp, w, context_size=160))
            assert models[-1].stages() == [(0, HP["n_layer"], slot)]
        L.ggml_hip_set_main_device(0)
        prompts = [np.random.default_rng(s).integers(0, hp["n_vocab"], 19).astype(np.int32) for s in (5, 6)]
        N = 96
        # one after the other, on this thread
        ref = []
        for m, p in zip(models, prompts):
            s = m.start_session(n_batch=8)
            ref.append(_decode(s, p, N))
            s.free()
        # together, one thread per session.  The worker threads never choose a slot themselves: every call binds the slot of
        # the model it was given (host/llm_host.cpp HomeDevice) and restores the thread's own afterwards.
        got = [None, None]
        seen_default = [None, None]
        start = threading.Barrier(2)
        sessions = [m.start_session(n_batch=8) for m in models]

        def run(i):
            seen_default[i] = L.ggml_hip_get_main_device()
            start.wait()
            got[i] = _decode(sessions[i], prompts[i], N)

        peak0 = _stat(G, "peak_concurrent_calls")
        th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        peak = _stat(G, "peak_concurrent_calls")
        for s in sessions:
            s.free()
        assert seen_default == [0, 0]  # a fresh thread follows the process default
        assert L.ggml_hip_get_main_device() == 0
        assert _stat(G, "fused_attn_timeouts") == 0
    finally:
        for m in models:
            m.free()
        L.ggml_hip_set_main_device(0)
        os.environ.pop("GGML_HIP_VIRTUAL_DEVICES", None)
    for r, g_ in zip(ref, got):
        assert [t for t, _ in r] == [t for t, _ in g_]
        for (_, la), (_, lb) in zip(r, g_):
            assert np.array_equal(la, lb)
    print("peak concurrent entry points:", peak0, "->", peak)
    assert peak >= 2  # both threads were inside the library at once (with one library lock this could never exceed 1)


def test_sessions_of_one_model_on_sibling_slots_share_its_weights(G):
    """The reference's use (model/mod.rs:275-276): ONE model, several sessions, each on its own thread.  Every session sits on its
    own device slot of the model's GPU (llm_start_session_on: own stream, shadows, K/V, plans) and reads the model's one copy of
    the weights.  Results = the same sessions run one after the other on the model's own slot, bit for bit — decode through the
    fused launches (which stay legal: 2 slots x 8 attention workgroups), prompt chunks, and a 40-token prompt batch through the
    prompt plan (resident f16 weight copies made by whichever slot comes first, under the library's w16 lock)."""
    from llm_amd import llama, synth
    L = G.lib()
    assert L.ggml_hip_get_main_device() == 0
    hp, w = synth.make_llama(HP, 2, seed=29)
    os.environ["GGML_HIP_VIRTUAL_DEVICES"] = "3"
    model = None
    try:
        G.set_option("fuse_attn", 2)
        model = llama.Llama(hp, w, context_size=160)
        prompts = [np.random.default_rng(s).integers(0, hp["n_vocab"], 19 + 40).astype(np.int32) for s in (11, 12, 13)]
        N = 48

        def decode(sess, p):
            sess.feed_prompt(p[:19])          # chunks of 8: the multi-token plan
            out = _decode(sess, p[19:], N)    # then 40 tokens in one evaluation (n_batch = 64: the prompt plan), then decode
            return out, sess.get_kv()

        ref = []
        for p in prompts:
            s = model.start_session(n_batch=64)
            ref.append(decode(s, p))
            s.free()
        f0 = _stat(G, "fused_attn_tokens")
        sessions = [model.start_session_on(i, n_batch=64) for i in range(3)]
        got = [None] * 3
        start = threading.Barrier(3)

        def run(i):
            start.wait()
            got[i] = decode(sessions[i], prompts[i])

        th = [threading.Thread(target=run, args=(i,)) for i in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for s in sessions:
            s.free()
        assert _stat(G, "fused_attn_timeouts") == 0
        assert L.ggml_hip_get_main_device() == 0
    finally:
        G.set_option("fuse_attn", 1)
        if model is not None:
            model.free()
        L.ggml_hip_set_main_device(0)
        os.environ.pop("GGML_HIP_VIRTUAL_DEVICES", None)
    for (ro, (rk, rv)), (go, (gk, gv)) in zip(ref, got):
        assert [t for t, _ in ro] == [t for t, _ in go]
        for (_, la), (_, lb) in zip(ro, go):
            assert np.array_equal(la, lb)
        assert np.array_equal(rk, gk) and np.array_equal(rv, gv)


def test_bind_thread_device_is_per_thread(G):
    """ggml_hip_bind_thread_device pins the calling thread only; ggml_hip_set_main_device also moves the process default."""
    L = G.lib()
    assert L.ggml_hip_get_main_device() == 0  # every test (and every entry point) leaves the main device as it found it
    os.environ["GGML_HIP_VIRTUAL_DEVICES"] = "3"
    seen = {}
    try:
        def worker(name, bind):
            if bind is not None:
                L.ggml_hip_bind_thread_device(bind)
            seen[name] = L.ggml_hip_get_main_device()

        t = threading.Thread(target=worker, args=("bound2", 2)); t.start(); t.join()
        t = threading.Thread(target=worker, args=("fresh", None)); t.start(); t.join()
        assert seen == {"bound2": 2, "fresh": 0} and L.ggml_hip_get_main_device() == 0
        L.ggml_hip_set_main_device(1)
        t = threading.Thread(target=worker, args=("fresh_after_default_1", None)); t.start(); t.join()
        assert seen["fresh_after_default_1"] == 1 and L.ggml_hip_get_main_device() == 1
    finally:
        L.ggml_hip_set_main_device(0)
        os.environ.pop("GGML_HIP_VIRTUAL_DEVICES", None)


def test_unchanged_callers_on_two_threads_get_a_slot_each():
    """GGML_HIP_SESSION_SLOTS=2 (opt-in, environment only): two threads that only call start_session() / infer — the reference's
    contract, inference_session.rs:43-48 — are assigned different sibling slots of the GPU when they create their K/V memory, decode
    concurrently and produce the tokens a single session produces.  (Child process: the variable is read when the library starts.)"""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, threading
        import numpy as np
        sys.path.insert(0, %r)
        from llm_amd import ggml as G, llama, synth
        hp, w = synth.make_llama(dict(n_vocab=256, n_embd=512, n_head=8, n_head_kv=8, n_layer=3, n_rot=64, n_ff=704, n_mult=32), 2, seed=5)
        model = llama.Llama(hp, w, context_size=96)
        toks = np.random.default_rng(1).integers(0, 256, 9).astype(np.int32)
        s0 = model.start_session(n_batch=8); s0.feed_prompt(toks); want = [s0.infer_next_token() for _ in range(20)]; s0.free()
        out, slots, bar = [None, None], [None, None], threading.Barrier(2)
        def run(i):
            s = model.start_session(n_batch=8)
            slots[i] = G.lib().ggml_hip_thread_session_slot()
            bar.wait()  # both sessions alive at once
            s.feed_prompt(toks)
            out[i] = [s.infer_next_token() for _ in range(20)]
            bar.wait()
            s.free()
        th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
        assert sorted(slots) == [0, 1], slots
        assert out[0] == want and out[1] == want, (out, want)
        print("OK", slots)
    """ % root)
    env = dict(os.environ, GGML_HIP_SESSION_SLOTS="2")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
