#!/usr/bin/env python
"""In-kernel timeline of k_p_attn (fused prompt attention) on a LLaMA-7B-shaped layer: per query tile, microseconds from
the workgroup's entry to: Q fragments loaded, scores in LDS, softmax done, V.P stored; and the launch span.
python tests/tools/pattn_timeline.py [N] [n_past]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from llm_amd import ggml  # noqa: E402


def report(t, ntile, N, n_past, H):
    e0 = t[:, 0].min()
    print(f"N {N} n_past {n_past}: {ntile * H} workgroups; launch span {(t[:, 4].max() - e0) / 100:.2f} us")
    print("tile  T_hi | entry   qload  scores softmax     vp | dur   (means over heads, us; entry relative to the first workgroup)")
    for qt in range(ntile):
        r = t[t[:, 6] == qt]
        rel = lambda j: float(((r[:, j] - r[:, 0]) / 100).mean())
        print(f"{qt:4d} {int(r[0, 5]):5d} | {float(((r[:, 0] - e0) / 100).mean()):6.2f} {rel(1):6.2f} {rel(2):6.2f} {rel(3):6.2f} {rel(4):6.2f} | "
              f"{rel(4):6.2f}")


def plan_mode(N):
    """the kernel as the prompt plan launches it (Q rotated on load, wo's operand written by the epilogue): LLaMA-7B Q4_0"""
    from llm_amd import llama, synth
    hp, w = synth.make_llama_fast(synth.LLAMA_7B, ggml.TYPE_Q4_0)
    model = llama.Llama(hp, w, context_size=2048)
    s = model.start_session(n_batch=N)
    toks = (np.arange(N + 1, dtype=np.int32) * 7 + 5) % hp["n_vocab"]
    s.feed_prompt(toks[:1])
    for _ in range(2):
        s.feed_prompt(toks[1:])
        assert s.rewind(N) == 0
    H, ntile = hp["n_head"], (N + 31) // 32
    ggml.set_option("timeline", ntile * H)
    s.feed_prompt(toks[1:])
    ggml.lib().ggml_hip_synchronize()
    t = ggml.read_timeline(ntile * H).astype(np.float64)
    ggml.set_option("timeline", 0)
    report(t, ntile, N, 1, H)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "plan":
        return plan_mode(int(sys.argv[2]) if len(sys.argv) > 2 else 512)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    n_past = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    H, D, C = 32, 128, 2048
    E = H * D
    rng = np.random.default_rng(1)
    q = rng.standard_normal((N, E)).astype(np.float32)
    mk = (0.5 * rng.standard_normal((C, E))).astype(np.float16)
    mv = (0.5 * rng.standard_normal((E, C))).astype(np.float16)
    out = np.zeros((N, E), np.float32)
    ntile = (N + 31) // 32
    ggml.set_option("timeline", ntile * H)
    L = ggml.lib()
    for _ in range(3):
        rc = L.ggml_hip_debug_prompt_attention(q.ctypes.data, mk.ctypes.data, mv.ctypes.data, out.ctypes.data, N, E, E, H, n_past, C,
                                               1.0 / np.sqrt(D), 1)
        assert rc == 0
    t = ggml.read_timeline(ntile * H).astype(np.float64)
    ggml.set_option("timeline", 0)
    report(t, ntile, N, n_past, H)


if __name__ == "__main__":
    main()
