import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from llm_amd import ggml as G, llama, synth
from oracle import oracle as O
cases = [("13b", dict(n_vocab=512, n_embd=5120, n_head=40, n_head_kv=40, n_layer=2, n_rot=128, n_ff=13824, n_mult=256), G.TYPE_Q5_1),
         ("13b-q40", dict(n_vocab=512, n_embd=5120, n_head=40, n_head_kv=40, n_layer=2, n_rot=128, n_ff=13824, n_mult=256), G.TYPE_Q4_0),
         ("65b", dict(n_vocab=512, n_embd=8192, n_head=64, n_head_kv=64, n_layer=1, n_rot=128, n_ff=22016, n_mult=256), G.TYPE_Q8_0)]
for name, hp0, wtype in cases:
    hp, w = synth.make_llama_fast(hp0, wtype)
    model = llama.Llama(hp, w, context_size=64)
    sess = model.start_session(n_batch=8)
    orcs = [O.Llama(hp, w, 64) for _ in range(3)]
    toks = np.random.default_rng(9).integers(0, hp["n_vocab"], 9).astype(np.int32)
    for chunk in (toks[:6],) + tuple(toks[6 + i:7 + i] for i in range(3)):
        got = sess.evaluate(chunk)
        e0 = orcs[0].evaluate(chunk, mode=0); e0r = orcs[1].evaluate(chunk, mode=0, reverse_blocks=True); e1 = orcs[2].evaluate(chunk, mode=1)
        k, v = sess.get_kv()
        for o in orcs:
            o.memory_k[:] = k; o.memory_v[:] = v
        std = float(e1.std())
        print(name, len(chunk), f"gpu-vs-exact {np.abs(got-e0).max()/std:.2e} band {np.abs(e0-e0r).max()/std:.2e} exact-vs-math {np.abs(e0-e1).max()/std:.2e} std {std:.3g} absmax {np.abs(e1).max():.3g}")
    sess.free(); model.free()
