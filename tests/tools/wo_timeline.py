#!/usr/bin/env python
"""In-kernel timeline of the WO form of k_qkv_attn (wq|wk|wv + attention + wo in one launch, LLaMA-7B Q4_0 synthetic): the
attention workgroups' phases and the mat-vec workgroups' second phase (wo_tail: start, wo rows requested, rows in LDS, first
head output seen, all gathered, exit), microseconds from the attention workgroups' entry, averaged over layers 2..L-1.
    python tests/tools/wo_timeline.py [n_past]"""
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
        a, wo, gate = t[5 * il + 1], t[5 * il + 2], t[5 * il + 3]
        av = a[:32][a[:32, 0] > 0]
        wv = wo[wo[:, 0] > 0]
        gv = gate[gate[:, 0] > 0]
        e0 = av[:, 0].min()
        rows.append([us(av[:, 1].mean() - e0), us(av[:, 4].mean() - e0), us(av[:, 5].mean() - e0), us(av[:, 5].max() - e0),
                     us(np.median(wv[:, 0]) - e0), us(np.median(wv[:, 1]) - e0), us(np.median(wv[:, 2]) - e0), us(np.median(wv[:, 3]) - e0),
                     us(wv[:, 3].max() - e0), us(np.median(wv[:, 4]) - e0), us(wv[:, 4].max() - e0), us(np.median(wv[:, 5]) - e0),
                     us(wv[:, 5].max() - e0), us(gv[:, 0].min() - e0) if len(gv) else float("nan")])
    # spread of the mat-vec workgroups' exit from wq|wk|wv (= start of their wo phase) and of the attention workgroups' stamps
    sp = []
    for il in range(2, L):
        a, wo = t[5 * il + 1], t[5 * il + 2]
        av = a[:32][a[:32, 0] > 0]
        wv = wo[wo[:, 0] > 0]
        e0 = av[:, 0].min()
        sp.append([us(np.percentile(wv[:, 0], q) - e0) for q in (0, 10, 50, 90, 99, 100)] + [us(av[:, 1].min() - e0), us(av[:, 1].max() - e0)])
    sp = np.array(sp).mean(axis=0)
    print("wq|wk|wv done per mat-vec workgroup, percentiles 0/10/50/90/99/100: " + " ".join("%.2f" % v for v in sp[:6]))
    print("attention has its rows, earliest / latest sampled head: %.2f / %.2f" % (sp[6], sp[7]))
    # per head (all heads are sampled when the timeline has room for them): when it had its rows, when it had published, by XCD
    hd = []
    for il in range(2, L):
        a = t[5 * il + 1]
        av = a[:32]
        if (av[:, 0] > 0).sum() < 32:
            break
        e0 = av[:, 0].min()
        hd.append(np.stack([us(av[:, 1] - e0), us(av[:, 4] - e0), us(av[:, 5] - e0), us(av[:, 6] - e0)]))
    lat = []
    for il in range(2, L):
        a, wo = t[5 * il + 1], t[5 * il + 2]
        av = a[:32][a[:32, 0] > 0]
        wv = wo[wo[:, 0] > 0]
        last = av[:, 5].max()
        lat.append([us(np.median(wv[:, 3]) - last), us(wv[:, 3].max() - last), us(np.median(wv[:, 4]) - last), us(wv[:, 4].max() - last)])
    lat = np.array(lat).mean(axis=0)
    print("after the LAST sampled head published: wave 0 of a mat-vec workgroup holds its granules %.2f (median) %.2f (max); whole workgroup %.2f / %.2f"
          % tuple(lat))
    if hd:
        hd = np.array(hd).mean(axis=0)
        for nm, row in zip(("has its rows", "V.P done", "published", "stores acked"), hd):
            print(f"attention {nm:14s} per head: min %.2f  median %.2f  max %.2f   by head mod 8: " % (row.min(), np.median(row), row.max())
                  + " ".join("%.2f" % row[x::8].mean() for x in range(8)))
    r = np.array(rows).mean(axis=0)
    names = ["attention has its rows", "attention V.P done", "attention exit mean", "attention exit max", "wo phase starts (median)",
             "wo rows requested", "wo rows in LDS", "first head output seen (median)", "first head output seen (max)",
             "all outputs gathered (median)", "all outputs gathered (max)", "mat-vec workgroup exit (median)",
             "mat-vec workgroup exit (max)", "next launch (w1|w3) first entry"]
    for n, v in zip(names, r):
        print(f"{n:45s} {v:8.2f}")


if __name__ == "__main__":
    main()
