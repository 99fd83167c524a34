#!/usr/bin/env python
"""In-kernel timeline of k_qkv_attn (wq|wk|wv + attention in one launch, LLaMA-7B Q4_0 synthetic, graph replay): when the
mat-vec workgroups finish, when the attention workgroups have their rows, and how long the attention tail takes — all on the
chip-wide 100 MHz clock, relative to the launch's first workgroup entry; averaged over layers 2..L-1.
    python tests/tools/fused_timeline.py [n_past]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from llm_amd import ggml, llama, synth  # noqa: E402


def main():
    n_prompt = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    hp, w = synth.make_llama_fast(synth.LLAMA_7B, ggml.TYPE_Q4_0)
    model = llama.Llama(hp, w, context_size=2048)
    s = model.start_session(n_batch=8)
    s.feed_prompt((np.arange(n_prompt, dtype=np.int32) * 7 + 5) % hp["n_vocab"])
    for _ in range(4):
        s.infer_next_token()
    nw = 224
    ggml.set_option("timeline", nw)
    for _ in range(3):
        s.infer_next_token()
    ggml.lib().ggml_hip_synchronize()
    t = ggml.read_timeline(1024 * nw).reshape(-1, nw, 8).astype(np.float64)
    ggml.set_option("timeline", 0)
    L = hp["n_layer"]
    us = lambda a: a / 100.0
    rows = []
    for il in range(2, L):
        q, a, wo = t[5 * il], t[5 * il + 1], t[5 * il + 2]
        qv = q[q[:, 0] > 0]
        av = a[:4][a[:4, 0] > 0]
        wv = wo[wo[:, 0] > 0]
        e0 = min(qv[:, 0].min(), av[:, 0].min())
        rows.append([us(qv[:, 0].max() - e0), us(np.median(qv[:, 2] - qv[:, 0])), us(np.median(qv[:, 5]) - e0), us(qv[:, 5].max() - e0),
                     us(av[:, 0].mean() - e0), us(av[:, 1].mean() - e0), us(av[:, 2].mean() - e0), us(av[:, 3].mean() - e0),
                     us(av[:, 4].mean() - e0), us(av[:, 5].mean() - e0), us(av[:, 5].max() - e0), us(wv[:, 0].min() - e0),
                     us(np.median(wv[:, 5]) - wv[:, 0].min()), av[0, 6]])
    r = np.array(rows).mean(axis=0)
    names = ["last producer entry", "producer staged (median, from its entry)", "producer exit median", "producer exit max",
             "consumer entry", "consumer has its rows (+barrier)", "scores done", "softmax done", "V.P done", "consumer exit mean",
             "consumer exit max", "next launch (wo) first entry", "wo median duration", "positions T"]
    for n, v in zip(names, r):
        print(f"{n:45s} {v:8.2f}")


if __name__ == "__main__":
    main()
