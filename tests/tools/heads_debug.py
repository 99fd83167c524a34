"""Diagnostic for attn_consumer_split (2-4 attention workgroups per head inside the wq|wk|wv launch): per-token |dlogit|/std of
the fused-heads path against itself (determinism), against the separate split attention and against the one-workgroup-per-head
form streaming beyond its window.   python tests/tools/heads_debug.py [mha|gqa]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from llm_amd import ggml as G, llama, synth

shape = sys.argv[1] if len(sys.argv) > 1 else "gqa"
HP = {"mha": dict(n_vocab=256, n_embd=2048, n_head=16, n_head_kv=16, n_layer=2, n_rot=128, n_ff=512, n_mult=32),
      "gqa": dict(n_vocab=256, n_embd=2048, n_head=16, n_head_kv=4, n_layer=2, n_rot=128, n_ff=512, n_mult=32)}[shape]
hp, w = synth.make_llama(HP, 2, seed=33)
ctx = 2048
model = llama.Llama(hp, w, context_size=ctx)
toks = np.random.default_rng(len(shape)).integers(0, hp["n_vocab"], ctx).astype(np.int32)
starts = [580, 1020, 1532, ctx - 6]


def run(heads, split):
    G.set_option("fuse_heads", heads)
    G.set_option("attn_split", split)
    s = model.start_session(n_batch=512)
    outs, pos = [], 0
    for st in starts:
        s.feed_prompt(toks[pos:st])
        for i in range(5):
            outs.append(s.evaluate(toks[st + i:st + i + 1])[-1].copy())
        pos = st + 5
    s.free()
    G.set_option("fuse_heads", 1)
    G.set_option("attn_split", 1)
    return outs


a1, a2 = run(1, 1), run(1, 1)
b = run(0, 512)       # separate split attention
c = run(0, 2047)      # one workgroup per head, positions beyond 512 streamed
d = lambda x, y: [float(np.max(np.abs(u - v)) / v.std()) for u, v in zip(x, y)]
print("fused heads vs itself     ", " ".join(f"{v:.1e}" for v in d(a1, a2)))
print("fused heads vs split      ", " ".join(f"{v:.1e}" for v in d(a1, b)))
print("fused heads vs one-wg     ", " ".join(f"{v:.1e}" for v in d(a1, c)))
print("split vs one-wg           ", " ".join(f"{v:.1e}" for v in d(b, c)))
print("timeouts", int(G.lib().ggml_hip_get_stat(b"fused_attn_timeouts")))

# teacher-forced against the oracle (tests/ infrastructure): every token evaluated from the device's own K/V state of that path
from oracle import oracle as O
orc = O.Llama(hp, w, ctx)
for name, heads, split in (("fused heads", 1, 1), ("split", 0, 512), ("one-wg", 0, 2047)):
    G.set_option("fuse_heads", heads)
    G.set_option("attn_split", split)
    s = model.start_session(n_batch=512)
    pos, res = 0, []
    for st in starts:
        s.feed_prompt(toks[pos:st])
        for i in range(5):
            k, v = s.get_kv()
            orc.memory_k[:] = k[:orc.memory_k.size]
            orc.memory_v[:] = v[:orc.memory_v.size]
            orc.n_past = st + i
            got = s.evaluate(toks[st + i:st + i + 1])[-1]
            ref = orc.evaluate(toks[st + i:st + i + 1], mode=O.ref_mode())[-1]
            res.append(float(np.max(np.abs(got - ref)) / ref.std()))
        pos = st + 5
    s.free()
    print(f"{name:12s} vs oracle ", " ".join(f"{v:.1e}" for v in res))
G.set_option("fuse_heads", 1)
G.set_option("attn_split", 1)
