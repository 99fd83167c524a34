"""ONE InferenceSession over several device slots of one process: the ggml-style layer split behind the C ABI (SURVEY.md
section 8e; the reference's hooks are ggml_cuda_set_tensor_split / ggml_cuda_set_main_device, crates/ggml/sys/src/cuda.rs:11,
:62, driven from crates/ggml/src/accelerator/mod.rs:68-77; the reference's split hook carries ONE float, so a split over
several slots is asked for through its explicit-length sibling ggml_hip_set_layer_split or GGML_HIP_LAYER_SPLIT).  A 1-GPU box has one device, so the slots are made virtual
(GGML_HIP_VIRTUAL_DEVICES: several slots — own stream, arena shadows, weight records, plan cache each — on the same GPU);
the residual then crosses between slots with a device copy instead of a peer copy over xGMI, everything else is the code a
multi-GPU node runs.  The split session must reproduce the unsplit one BIT FOR BIT: same kernels, same order, and the hop
moves f32 values unchanged."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HP = dict(n_vocab=256, n_embd=128, n_head=4, n_head_kv=4, n_layer=5, n_rot=32, n_ff=352, n_mult=32)


def _run(G, model, toks, n_batch):
    sess = model.start_session(n_batch=n_batch)
    outs = [sess.evaluate(toks[:40]), sess.evaluate(toks[40:48]), sess.evaluate(toks[48:51])]
    for i in range(4):
        outs.append(sess.evaluate(toks[51 + i:52 + i]))
    sess.feed_prompt(toks[55:60])
    ids = [sess.infer_next_token() for _ in range(3)]
    assert sess.rewind(2) == 0
    ids += [sess.infer_next_token() for _ in range(2)]
    k, v = sess.get_kv()
    last = sess.last_logits()
    sess.free()
    return outs, ids, k, v, last


@pytest.mark.parametrize("how", ["env3", "fractions"])
def test_layer_split_over_device_slots_reproduces_the_unsplit_session(G, how):
    from llm_amd import llama, synth
    assert G.lib().ggml_hip_get_main_device() == 0  # every test (and every entry point) leaves the main device as it found it
    hp, w = synth.make_llama(HP, 2, seed=17)
    toks = np.random.default_rng(2).integers(0, hp["n_vocab"], 64).astype(np.int32)
    whole = llama.Llama(hp, w, context_size=96)
    assert whole.stages() == [(0, 5, 0)]
    ref = _run(G, whole, toks, 64)
    whole.free()
    os.environ["GGML_HIP_VIRTUAL_DEVICES"] = "3"
    try:
        assert G.lib().ggml_hip_device_count() >= 3
        if how == "env3":
            os.environ["GGML_HIP_LAYER_SPLIT"] = "3"
            want = [(0, 2, 0), (2, 3, 1), (3, 5, 2)]
        else:
            fr = (np.array([0.2, 0.8, 0.0], np.float32))  # ggml's fractions: slot 0 takes 20 %, slot 1 the rest, slot 2 nothing
            G.lib().ggml_hip_set_layer_split(fr.ctypes.data, 3)
            want = None
        split = llama.Llama(hp, w, context_size=96)
        st = split.stages()
        print(how, st)
        if want:
            assert st == want
        else:
            assert st == [(0, 1, 0), (1, 5, 1)]  # 20 % of 5 layers on slot 0, the rest on slot 1, none on slot 2
        got = _run(G, split, toks, 64)
        split.free()
    finally:
        os.environ.pop("GGML_HIP_LAYER_SPLIT", None)
        G.lib().ggml_hip_set_layer_split(None, 0)
        G.lib().ggml_hip_set_main_device(0)
        os.environ.pop("GGML_HIP_VIRTUAL_DEVICES", None)
    for a, b in zip(ref[0], got[0]):
        assert np.array_equal(a, b)
    assert ref[1] == got[1]
    assert np.array_equal(ref[2], got[2]) and np.array_equal(ref[3], got[3]) and np.array_equal(ref[4], got[4])
    # the library is back on slot 0 and an unsplit model works as before
    again = llama.Llama(hp, w, context_size=96)
    assert again.stages() == [(0, 5, 0)]
    r2 = _run(G, again, toks, 64)
    again.free()
    assert np.array_equal(ref[4], r2[4])


def test_snapshot_moves_between_a_split_and_an_unsplit_session(G):
    """InferenceSnapshot (inference_session.rs:590-646) of a session that is split over device slots: the stages' K/V caches
    concatenated in layer order are the unsplit layout, so a snapshot taken from a 3-slot session restores into an unsplit
    model (and the other way round) and decoding continues with the same tokens and logits."""
    from llm_amd import llama, synth
    assert G.lib().ggml_hip_get_main_device() == 0  # every test (and every entry point) leaves the main device as it found it
    hp, w = synth.make_llama(HP, 2, seed=19)
    toks = np.random.default_rng(4).integers(0, hp["n_vocab"], 30).astype(np.int32)
    whole = llama.Llama(hp, w, context_size=96)
    ref = whole.start_session(n_batch=8)
    ref.feed_prompt(toks[:21])
    ids_a = [ref.infer_next_token() for _ in range(3)]
    snap_whole = ref.snapshot()
    ids_b = [ref.infer_next_token() for _ in range(4)]
    last = ref.last_logits()
    os.environ["GGML_HIP_VIRTUAL_DEVICES"] = "3"
    os.environ["GGML_HIP_LAYER_SPLIT"] = "3"
    try:
        split = llama.Llama(hp, w, context_size=96)
        assert len(split.stages()) == 3
        s = split.start_session(n_batch=8)
        s.feed_prompt(toks[:21])
        assert [s.infer_next_token() for _ in range(3)] == ids_a
        snap_split = s.snapshot()
        assert len(snap_split) == len(snap_whole)
        # unsplit snapshot -> split session
        s2 = split.session_from_snapshot(snap_whole)
        assert s2 is not None
        assert [s2.infer_next_token() for _ in range(4)] == ids_b
        assert np.array_equal(s2.last_logits(), last)
        s.free()
        s2.free()
        split.free()
    finally:
        os.environ.pop("GGML_HIP_LAYER_SPLIT", None)
        G.lib().ggml_hip_set_main_device(0)
        os.environ.pop("GGML_HIP_VIRTUAL_DEVICES", None)
    # split snapshot -> unsplit session
    r2 = whole.session_from_snapshot(snap_split)
    assert r2 is not None
    assert [r2.infer_next_token() for _ in range(4)] == ids_b
    assert np.array_equal(r2.last_logits(), last)
    assert snap_split == snap_whole  # the same bytes: the split is invisible in the snapshot
    for x in (ref, r2):
        x.free()
    whole.free()
