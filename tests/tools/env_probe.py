#!/usr/bin/env python
"""One line per process: decode tok/s, launch floor and us per launch of the five mat-vec kinds (LLaMA-7B Q4_0), for
comparing HIP runtime settings (the caller sets the environment):  python tests/tools/env_probe.py <label>"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from llm_amd import ggml, llama, synth  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "default"
hp, w = synth.make_llama_fast(synth.LLAMA_7B, ggml.TYPE_Q4_0)
model = llama.Llama(hp, w, context_size=2048)
s = model.start_session(n_batch=8)
s.feed_prompt((np.arange(128, dtype=np.int32) * 7 + 5) % hp["n_vocab"])
L = ggml.lib()
for _ in range(8):
    s.infer_next_token()
L.ggml_hip_synchronize()
t0 = time.perf_counter()
for _ in range(64):
    s.infer_next_token()
L.ggml_hip_synchronize()
dt = (time.perf_counter() - t0) / 64
fl = ggml.bench_empty(256, 1024, 20480, 448)
per = []
for k in range(5):
    ms, kn, kb = ggml.bench_plan_class(ggml.KKIND_BASE + k, 20)
    per.append(ms * 1e3 / max(kn * 20, 1))
ms, kn, kb = ggml.bench_plan_class(ggml.KCLASS_MMVQ, 20)
print(f"{label:40s} {1 / dt:7.1f} tok/s | floor {fl[0]:.2f} | qkv {per[0]:.2f} wo {per[1]:.2f} gate_up {per[2]:.2f} down {per[3]:.2f} "
      f"lm_head {per[4]:.2f} | all mat-vecs {ms / 20 * 1e3:.1f} us")
