"""Decode rate as a function of the context length and of the switch between the two decode attention paths: k_qkv_attn (the
attention rides in the wq|wk|wv launch, one workgroup per head, register window of 512 positions) below option attn_split,
the position-split attention (k_attn_split_one after a plain wq|wk|wv launch) from it on.  LLaMA-7B Q4_0, random valid blocks.
    python tests/tools/ctx_sweep.py [thresholds ...]      default 256 384 512 768 1024"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from llm_amd import ggml, llama, synth

thresholds = [int(x) for x in sys.argv[1:]] or [256, 384, 512, 768, 1024]
NB = int(os.environ.get("CTX_SWEEP_NBATCH", "512"))  # the sessions' n_batch (the prompt is fed with it; the decode steps build their graphs in its arena)
positions = [int(x) for x in os.environ.get("CTX_SWEEP_POSITIONS", "200 300 400 480 560 700 900 1100 1500 1900").split()]
MODEL = os.environ.get("CTX_SWEEP_MODEL", "7b")  # 7b | 13b | 65b
WT = {"q4_0": ggml.TYPE_Q4_0, "q5_1": ggml.TYPE_Q5_1, "q8_0": ggml.TYPE_Q8_0}[os.environ.get("CTX_SWEEP_WTYPE", "q4_0")]
L = ggml.lib()
hp, w = synth.make_llama_fast({"7b": synth.LLAMA_7B, "13b": synth.LLAMA_13B, "65b": synth.LLAMA_65B}[MODEL], WT)
model = llama.Llama(hp, w, context_size=2048)
toks = np.random.default_rng(5).integers(0, hp["n_vocab"], 2048).astype(np.int32)
print(f"n_batch = {NB}, model {MODEL}")
print("ms per token by n_past; columns = option attn_split (positions from which the split attention is used)")
print("n_past  " + "".join(f"{t:>9d}" for t in thresholds))
rows = {p: [] for p in positions}
for th in thresholds:
    ggml.set_option("attn_split", th)
    for p in positions:
        s = model.start_session(n_batch=NB)
        s.feed_prompt(toks[:p])
        for _ in range(6):
            s.infer_next_token()
        L.ggml_hip_synchronize()
        t0 = time.perf_counter()
        for _ in range(24):
            s.infer_next_token()
        L.ggml_hip_synchronize()
        rows[p].append((time.perf_counter() - t0) / 24 * 1e3)
        s.free()
ggml.set_option("attn_split", 1)
for p in positions:
    print(f"{p:6d}  " + "".join(f"{v:9.4f}" for v in rows[p]), flush=True)
model.free()
