#!/usr/bin/env python
"""In-kernel timeline of the decode mat-vec launches of one token (LLaMA-7B Q4_0 synthetic, graph replay):
per launch and sampled workgroup, microseconds from kernel entry to: loads issued, x staged, barrier passed, first
weights consumed, exit; plus the gap to the previous instrumented launch.   python tests/tools/timeline.py [model]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from llm_amd import ggml, llama, synth  # noqa: E402


def distribution(t, L, nw):
    """Entry / exit spread over ALL sampled workgroups per launch kind (layers 2..L-1 averaged): is the launch
    waiting for a few slow workgroups?"""
    n = 5 * L + 1
    t = t[:n]
    keep = [i for i in range(n) if not (i % 5 == 1 and i < 5 * L)]
    t = t[keep].astype(np.float64) / 100.0
    names = ["qkv", "wo", "gate", "down"]
    pct = (0, 10, 50, 90, 99, 100)
    print("per launch kind, us relative to the first workgroup's entry; percentiles over workgroups " + str(pct))
    for k, nm in enumerate(names + ["lm_head"]):
        rows = t[k + 8:4 * L:4] if k < 4 else t[4 * L:4 * L + 1]
        e0 = rows[:, :, 0].min(axis=1, keepdims=True)
        ent = rows[:, :, 0] - e0
        ext = rows[:, :, 5] - e0
        dur = rows[:, :, 5] - rows[:, :, 0]
        stg = rows[:, :, 2] - rows[:, :, 0]
        f = lambda a: " ".join("%6.2f" % np.percentile(a, q, axis=1).mean() for q in pct)
        print(f"{nm:8s} entry  {f(ent)}")
        print(f"{nm:8s} staged {f(stg)}")
        print(f"{nm:8s} dur    {f(dur)}")
        print(f"{nm:8s} exit   {f(ext)}")
        # which workgroups are the slow ones?  (by XCD = blockIdx % 8 and by position in the grid)
        by_xcd = [dur[:, x::8].mean() for x in range(8)]
        print(f"{nm:8s} mean dur by blockIdx%8: " + " ".join("%5.2f" % v for v in by_xcd))
        late = np.argsort(-ext.mean(axis=0))[:8]
        print(f"{nm:8s} latest workgroups: " + " ".join("%d(%.2f)" % (int(rows[0, j, 7] * 100), ext[:, j].mean()) for j in late))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "7b"
    hp0 = {"7b": synth.LLAMA_7B, "13b": synth.LLAMA_13B, "tiny": synth.TINY}[name]
    hp, w = synth.make_llama_fast(hp0, ggml.TYPE_Q4_0)
    model = llama.Llama(hp, w, context_size=2048 if name != "tiny" else 256)
    s = model.start_session(n_batch=8)
    s.feed_prompt((np.arange(128, dtype=np.int32) * 7 + 5) % hp["n_vocab"])
    for _ in range(4):
        s.infer_next_token()
    nw = int(sys.argv[2]) if len(sys.argv) > 2 else 4  # sampled workgroups per launch (e.g. 256 = all of them)
    ggml.set_option("timeline", 1 if nw == 4 else nw)
    for _ in range(3):
        s.infer_next_token()
    ggml.lib().ggml_hip_synchronize()
    t = ggml.read_timeline(1024 * nw)
    ggml.set_option("timeline", 0)
    t = t.reshape(-1, nw, 8)
    if nw != 4:
        return distribution(t, hp["n_layer"], nw)
    L = hp["n_layer"]
    n = 5 * L + 1
    t = t[:n]
    att = t[1:5 * L:5]
    keep = [i for i in range(n) if not (i % 5 == 1 and i < 5 * L)]
    t = t[keep]
    n = 4 * L + 1
    a0 = att[2:, 0]
    print("attention (us from entry): loaded %.2f scores %.2f softmax %.2f vp %.2f exit %.2f  T=%d" % tuple(
        [float((a0[:, k] - a0[:, 0]).mean()) / 100 for k in (1, 2, 3, 4, 5)] + [int(a0[0, 6])]))
    names = ["qkv", "wo", "gate", "down"]
    us = lambda a: a / 100.0
    print("launch      wg  S | issued staged barrier first  dots  exit | gap_from_prev_exit(us)")
    prev_exit = None
    for i in range(n):
        nm = names[i % 4] + str(i // 4) if i < 4 * L else "lm_head"
        ent = t[i, :, 0].min()
        for j in range(4):
            r = t[i, j]
            if r[0] == 0:
                continue
            if not (i // 4 in (15,) or i == 4 * L):
                continue
            gap = us(r[0] - prev_exit) if prev_exit is not None else float("nan")
            print(f"{nm:10s} {int(r[7]):4d} {int(r[6]) & 0xffff:2d} | {us(r[1]-r[0]):6.2f} {us(r[2]-r[0]):6.2f} {us(r[3]-r[0]):7.2f} "
                  f"{us(r[4]-r[0]):6.2f} {us(int(r[6]) >> 32):5.2f} {us(r[5]-r[0]):6.2f} | {gap:6.2f}   entry_skew {us(r[0]-ent):5.2f}")
        prev_exit = t[i, :, 5].max()
    # averages over layers 2..L-1
    for k, nm in enumerate(names):
        rows = t[k + 8:4 * L:4, 0]
        d = lambda a, b: us((rows[:, a] - rows[:, b]).mean())
        dots = us((rows[:, 6] >> 32).mean())
        print(f"avg {nm:5s}: issued {d(1,0):5.2f} staged {d(2,0):5.2f} barrier {d(3,0):5.2f} first {d(4,0):5.2f} dots {dots:5.2f} exit {d(5,0):5.2f}")
    ex = t[:4 * L, :, 5].max(axis=1)
    en = t[:4 * L, :, 0].min(axis=1)
    gaps = us(en[1:] - ex[:-1])
    for k, nm in enumerate(names):
        print(f"gap before {names[(k + 1) % 4]:5s}: {gaps[k::4].mean():5.2f} us")
    print(f"token span (first entry -> lm_head exit): {us(t[n - 1, :, 5].max() - t[0, :, 0].min()):.1f} us")


if __name__ == "__main__":
    main()
