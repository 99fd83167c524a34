#!/usr/bin/env python
"""L2 warm-up sweep (option "prefetch"): decode rate and per-launch times of the mat-vec kinds and of
the attention launch for several amounts of w1|w3 pulled by the idle CUs of the attention launch.
   python tests/tools/prefetch_probe.py [7b|13b] [q4_0|...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from llm_amd import ggml, llama, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "7b"
wt = sys.argv[2] if len(sys.argv) > 2 else "q4_0"
wtype = {"q4_0": ggml.TYPE_Q4_0, "q4_1": ggml.TYPE_Q4_1, "q5_0": ggml.TYPE_Q5_0, "q5_1": ggml.TYPE_Q5_1, "q8_0": ggml.TYPE_Q8_0}[wt]
hp, w = synth.make_llama_fast({"7b": synth.LLAMA_7B, "13b": synth.LLAMA_13B}[name], wtype)
model = llama.Llama(hp, w, context_size=2048)
L = ggml.lib()
prompt = (np.arange(128, dtype=np.int32) * 7 + 5) % hp["n_vocab"]
print("prefetch MB  wo delay | tok/s (two runs of 64 tokens from position 128, fresh session each) | attention launch us")
CASES = [(0, 1, 0, 0)] + [(mb, wo, 0, wgs) for wo in (0, 1) for mb in (8, 16) for wgs in (16, 32, 64, 128)] + [(4, 0, 0, 32), (12, 0, 0, 64), (0, 1, 0, 0)]
for mb, wo, delay, wgs in CASES:
    ggml.set_option("prefetch_wgs", wgs)
    ggml.set_option("prefetch_wo", wo)
    ggml.set_option("prefetch_delay", delay)
    ggml.set_option("prefetch", mb)
    rates = []
    for rep in range(2):
        s = model.start_session(n_batch=8)
        s.feed_prompt(prompt)
        for _ in range(6):
            s.infer_next_token()
        L.ggml_hip_synchronize()
        t0 = time.perf_counter()
        for _ in range(64):
            s.infer_next_token()
        L.ggml_hip_synchronize()
        rates.append(64 / (time.perf_counter() - t0))
        if rep == 1:
            ams, an, _ = ggml.bench_plan_class(ggml.KCLASS_ATTN, 20)
            at = ams * 1e3 / 20 / max(an, 1)
        s.free()
    print(f"   {mb:6d}   {wo:2d} {delay:4d} wgs {wgs:3d} | {rates[0]:7.1f} {rates[1]:7.1f} | {at:6.2f}")
