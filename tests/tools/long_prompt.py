#!/usr/bin/env python
"""A 2048-token prompt fed in four 512-token batches (LLaMA-7B Q4_0 synthetic): milliseconds per batch with the fused prompt
attention (32 queries per workgroup up to 1184 keys, 16 beyond) and with the three-launch attention.
python tests/tools/long_prompt.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from llm_amd import ggml, llama, synth  # noqa: E402

hp, w = synth.make_llama_fast(synth.LLAMA_7B, ggml.TYPE_Q4_0)
model = llama.Llama(hp, w, context_size=2048)
toks = (np.arange(2047, dtype=np.int32) * 7 + 5) % hp["n_vocab"]
L = ggml.lib()
for fused in (1, 0, 1):
    ggml.set_option("attn_fused", fused)
    s = model.start_session(n_batch=512)
    ms = []
    for i in range(0, 2047, 512):
        L.ggml_hip_synchronize()
        t0 = time.perf_counter()
        s.feed_prompt(toks[i:i + 512])
        L.ggml_hip_synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    s.free()
    print(f"attn_fused={fused}: ms per 512-token batch at n_past 0 / 512 / 1024 / 1536: " + " ".join("%.2f" % x for x in ms) +
          f"  -> {2047 / sum(ms) * 1e3:.0f} tok/s over the prompt")
ggml.set_option("attn_fused", 1)
model.free()
