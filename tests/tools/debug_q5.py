import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from llm_amd import ggml, llama, synth
from oracle import oracle as O
wtype = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = dict(synth.TINY); cfg["n_layer"] = int(os.environ.get("NL", "2"))
hp, w = synth.make_llama(cfg, wtype)
m = llama.Llama(hp, w, context_size=64)
s = m.start_session()
orc = O.Llama(hp, w, 64)
toks = np.random.default_rng(42).integers(0, hp["n_vocab"], 8).astype(np.int32)
got = s.evaluate(toks)
ref, taps = orc.evaluate(toks, mode=0, taps=True)
L = hp["n_layer"]
def cmp(label, a, b):
    a = a.reshape(-1); b = b.reshape(-1)
    i = int(np.argmax(np.abs(a - b)))
    print(f"{label:18s} max|d|={np.abs(a-b).max():.3e} rel={np.abs(a-b).max()/ (np.abs(b).max()+1e-30):.3e} at {i} gpu={a[i]:.6f} ref={b[i]:.6f}")
cmp("inpL0", s.read_node(0), taps["inpL0"])
cmp("attn_norm", s.read_node(2), taps["layer0_attn_norm"])
cmp("Qcur", s.read_node(name="Qcur"), taps["layer0_q"])
cmp("KQ_soft_max", s.read_node(name="KQ_soft_max"), taps["layer0_kq"])
cmp("layer0_out", s.read_node(37), taps["layer0_out"])
cmp("final_norm", s.read_node(37 * L + 2), taps["final_norm"])
cmp("logits", got, ref)
