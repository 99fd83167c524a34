import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from llm_amd import ggml as G, gpt2
from oracle import oracle as O
for wtype in (2, 8):
    for seed in (5, 6, 7):
        hp, w = gpt2.make_gpt2(gpt2.GPT2_TINY, wtype, seed=seed)
        model = gpt2.Gpt2(hp, w)
        orc = O.Gpt2(hp, w); orc1 = O.Gpt2(hp, w)
        toks = np.random.default_rng(6).integers(0, hp["n_vocab"], 12).astype(np.int32)
        ds = []
        for chunk in (toks[:5], toks[5:8]) + tuple(toks[8 + i:9 + i] for i in range(4)):
            got = model.evaluate(chunk)
            for o in (orc, orc1):
                o.memory_k[:] = model.memory_k.device_get(np.float16).reshape(o.memory_k.shape)
                o.memory_v[:] = model.memory_v.device_get(np.float16).reshape(o.memory_v.shape)
                o.n_past = model.n_past - len(chunk)
            ref = orc.evaluate(chunk, mode=0); r1 = orc1.evaluate(chunk, mode=1)
            ds.append((float(np.max(np.abs(got - ref)) / ref.std()), float(np.max(np.abs(ref - r1)) / ref.std())))
        print(wtype, seed, " ".join(f"{a:.1e}/{b:.1e}" for a, b in ds))
        model.free()
# op-level: gelu and norm
x = (np.random.default_rng(1).standard_normal((7, 128)) * 3).astype(np.float32)
with G.Context(1 << 22) as ctx:
    t = ctx.tensor_from(x)
    y = ctx.op_gelu(t); z = ctx.op_norm(t)
    g = ctx.graph(); g.build_forward_expand(y); g.build_forward_expand(z); g.compute()
    print("gelu max diff", np.abs(y.read_data().reshape(x.shape) - O.gelu(x, 0)).max(), "norm", np.abs(z.read_data().reshape(x.shape) - O.norm(x)).max())
