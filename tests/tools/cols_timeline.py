#!/usr/bin/env python
"""In-kernel timeline of the k_mmq_cols launches of one 8-token prompt chunk (LLaMA-7B Q4_0 synthetic): per launch kind,
microseconds from the workgroup's entry to: weight ring issued, activations staged (barrier), first step consumed, loop
end (wave 0), second barrier, exit; spread over the sampled workgroups.   python tests/tools/cols_timeline.py [nwgs] [n_batch]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from llm_amd import ggml, llama, synth  # noqa: E402


def main():
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    nbatch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    hp, w = synth.make_llama_fast(synth.LLAMA_7B, ggml.TYPE_Q4_0)
    model = llama.Llama(hp, w, context_size=2048)
    s = model.start_session(n_batch=nbatch)
    toks = (np.arange(256, dtype=np.int32) * 7 + 5) % hp["n_vocab"]
    s.feed_prompt(toks[:64])
    ggml.set_option("timeline", nw)
    s.feed_prompt(toks[64:64 + 2 * nbatch])  # the second chunk's records overwrite the first's (same launch order)
    ggml.lib().ggml_hip_synchronize()
    L = hp["n_layer"]
    t = ggml.read_timeline(1024 * nw).reshape(-1, nw, 8)[:4 * L + 1].astype(np.float64)
    ggml.set_option("timeline", 0)
    names = ["qkv", "wo", "gate", "down", "lm_head"]
    print(f"n_batch {nbatch}; {nw} sampled workgroups per launch; layers 2.. averaged; us from the workgroup's entry")
    print("kind     |    pre    dma issued staged   loop  sync2   exit | entry spread p50 p100 | launch span | loop mean min max")
    for k, nm in enumerate(names):
        rows = t[k + 8:4 * L:4] if k < 4 else t[4 * L:4 * L + 1]
        e0 = rows[:, :, 0].min(axis=1, keepdims=True)
        rel = lambda j: ((rows[:, :, j] - rows[:, :, 0]) / 100.0)
        seg = [rel(j).mean() for j in (1, 2, 3, 4, 5, 6, 7)]
        ent = (rows[:, :, 0] - e0) / 100.0
        ext = (rows[:, :, 7] - e0) / 100.0
        lp = (rows[:, :, 5] - rows[:, :, 4]) / 100.0
        print(f"{nm:8s} | " + " ".join("%6.2f" % v for v in seg) +
              " | %6.2f %6.2f | %6.2f | %6.2f %6.2f %6.2f" % (np.percentile(ent, 50, axis=1).mean(), ent.max(axis=1).mean(),
                                                             ext.max(axis=1).mean(), lp.mean(), lp.min(), lp.max()))
    s.free()
    model.free()


if __name__ == "__main__":
    main()
