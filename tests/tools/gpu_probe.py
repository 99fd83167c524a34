"""Ad-hoc GPU probe (not a test): times the stages of a tiny-model session to find host-side stalls."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from llm_amd import ggml, llama, synth

t0 = time.perf_counter()
def lap(msg):
    global t0
    t1 = time.perf_counter()
    print(f"{msg}: {(t1 - t0) * 1e3:.1f} ms", flush=True)
    t0 = t1

ggml.lib(); lap("load lib")
hp, w = synth.make_llama(synth.TINY, ggml.TYPE_Q4_0); lap("make weights")
m = llama.Llama(hp, w, context_size=64); lap("model new (uploads)")
s = m.start_session(); lap("session new")
toks = np.arange(8, dtype=np.int32)
s.evaluate(toks); lap("evaluate #1 (N=8)")
s.evaluate(toks[:1]); lap("evaluate #2 (N=1)")
for i in range(5):
    s.evaluate(toks[:1]); lap(f"evaluate #{i+3} (N=1)")
s.free(); lap("session free")
s = m.start_session(); lap("session new #2")
s.evaluate(toks); lap("evaluate (N=8)")
s.free(); lap("session free #2")
m.free(); lap("model free")
