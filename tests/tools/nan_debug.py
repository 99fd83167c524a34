"""scratch: which option combination makes a tiny model's logits NaN"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from llm_amd import ggml as G, llama, synth
SHAPES = {"tiny": dict(n_vocab=256, n_embd=128, n_head=4, n_head_kv=4, n_layer=2, n_rot=32, n_ff=352, n_mult=32),
          "gqa": dict(n_vocab=256, n_embd=256, n_head=8, n_head_kv=2, n_layer=2, n_rot=32, n_ff=352, n_mult=32)}
for shape in ("gqa", "tiny"):
    for wtype in (8, 2):
        hp, w = synth.make_llama(SHAPES[shape], wtype, seed=31)
        for opts in ({}, {"warm_norm": 0}, {"big": 0}, {"fuse_attn": 0}, {"graph": 0}, {"plan": 0}):
            for k, v in opts.items():
                G.set_option(k, v)
            model = llama.Llama(hp, w, context_size=96)
            sess = model.start_session(n_batch=8)
            toks = np.random.default_rng(8).integers(0, hp["n_vocab"], 13).astype(np.int32)
            sess.feed_prompt(toks)
            l0 = sess.last_logits().copy()
            nan_at = []
            for i in range(4):
                sess.infer_next_token()
                nan_at.append(int(np.isnan(sess.last_logits()).sum()))
            print(shape, wtype, opts, "prompt nan", int(np.isnan(l0).sum()), "decode nan", nan_at, flush=True)
            sess.free(); model.free()
            for k in opts:
                G.set_option(k, 1)
