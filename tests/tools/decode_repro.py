#!/usr/bin/env python
"""Decode tokens/s of a synthetic model (two prompt patterns) and the per-class device time of one token: the quick
look for types without a fused plan (K-quants: `decode_repro.py 7b q4_k`)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from llm_amd import ggml, llama, synth  # noqa: E402

name, wt = sys.argv[1], sys.argv[2]
wtype = {"q4_0": ggml.TYPE_Q4_0, "q5_1": ggml.TYPE_Q5_1, "q8_0": ggml.TYPE_Q8_0, "q4_k": ggml.TYPE_Q4_K, "q6_k": ggml.TYPE_Q6_K}[wt]
hp0 = {"7b": synth.LLAMA_7B, "13b": synth.LLAMA_13B}[name]
hp, w = synth.make_llama_fast(hp0, wtype)
model = llama.Llama(hp, w, context_size=2048)
L = ggml.lib()
for pat in ("arange", "random"):
    s = model.start_session(n_batch=8)
    toks = (np.arange(128, dtype=np.int32) * 7 + 5) % hp["n_vocab"] if pat == "arange" else \
        np.random.default_rng(42).integers(0, hp["n_vocab"], 128).astype(np.int32)
    s.feed_prompt(toks)
    for _ in range(8):
        s.infer_next_token()
    L.ggml_hip_synchronize()
    t0 = time.perf_counter()
    for _ in range(64):
        s.infer_next_token()
    L.ggml_hip_synchronize()
    dt = time.perf_counter() - t0
    print(name, wt, pat, f"{64 / dt:.1f} tok/s", "generic graphs", int(L.ggml_hip_get_stat(b"generic_graphs")))
    if pat == "random":
        for cls, nm in ((ggml.KCLASS_MMVQ, "mmvq"), (ggml.KCLASS_ATTN, "attn"), (ggml.KCLASS_OTHER, "other")):
            pass
        L.ggml_hip_timing_begin()
        s.infer_next_token()
        L.ggml_hip_synchronize()
        L.ggml_hip_timing_end()
        for cls, nm in ((ggml.KCLASS_MMVQ, "mmvq"), (ggml.KCLASS_MMQ_MFMA, "mfma"), (ggml.KCLASS_ATTN, "attn"), (ggml.KCLASS_OTHER, "other")):
            ms, n, b = ggml.timing_query(cls)
            print(f"   class {nm}: {ms:.3f} ms in {n} launches, {b / 1e9 / max(ms, 1e-9) * 1e3:.0f} GB/s" if n else f"   class {nm}: -")
    s.free()
model.free()
