#!/usr/bin/env python
"""Where does a decode mat-vec launch spend its time?  (LLaMA-7B Q4_0, one token's launches replayed from a hipGraph)

1. launch floor: a do-nothing kernel with the decode launch shapes (ggml_hip_bench_empty): us per launch, end -> next
   first instruction, first instruction -> kernel arguments usable;
2. per mat-vec kind (wq|wk|wv, wo, w1|w3, w2, lm_head), us per launch incl. boundary with the kernel cut short:
     probe 1 = return at entry (kernel arguments read)         -> boundary + launch shape
     probe 2 = return once x is staged and the ring requested  -> + prologue
     probe 3 = everything but the epilogue stores
     probe 4 = every other step's dots skipped;  probe 5 = no wave reductions
     probe 0 = the real kernel
   for the 2-steps-before-staging kernel ("big" = 1) and the whole-ring-first kernel ("big" = 2);
3. decode tokens/s of both.

   python tests/tools/launch_probe.py [7b|13b] [q4_0|...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from llm_amd import ggml, llama, synth  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "7b"
    wt = sys.argv[2] if len(sys.argv) > 2 else "q4_0"
    wtype = {"q4_0": ggml.TYPE_Q4_0, "q4_1": ggml.TYPE_Q4_1, "q5_0": ggml.TYPE_Q5_0, "q5_1": ggml.TYPE_Q5_1,
             "q8_0": ggml.TYPE_Q8_0}[wt]
    print("== launch floor: do-nothing kernel, 64 launches per graph ==")
    print("  wgs threads   lds  kernarg | us/launch  end->first_instr  first_instr->kernarg")
    for wgs, thr, lds, ka in [(256, 1024, 0, 64), (256, 1024, 0, 448), (256, 1024, 20480, 448), (256, 1024, 65536, 448),
                              (256, 1024, 150 * 1024, 448), (256, 512, 20480, 448), (256, 256, 20480, 448),
                              (512, 512, 20480, 448), (1024, 256, 20480, 448), (2048, 256, 0, 64), (32, 1024, 40960, 192),
                              (256, 768, 20480, 448), (256, 960, 20480, 448)]:
        r = ggml.bench_empty(wgs, thr, lds, ka)
        print(f"  {wgs:4d} {thr:6d} {lds:6d} {ka:7d} | {r[0]:8.2f} {r[1]:14.2f} {r[2]:18.2f}")

    hp0 = {"7b": synth.LLAMA_7B, "13b": synth.LLAMA_13B}[name]
    hp, w = synth.make_llama_fast(hp0, wtype)
    model = llama.Llama(hp, w, context_size=2048)
    s = model.start_session(n_batch=8)
    s.feed_prompt((np.arange(128, dtype=np.int32) * 7 + 5) % hp["n_vocab"])
    kinds = ["qkv", "wo", "gate_up", "down", "lm_head"]
    L = ggml.lib()
    for big in (1,):
        ggml.set_option("big", big)
        for _ in range(4):
            s.infer_next_token()
        L.ggml_hip_synchronize()
        t0 = time.perf_counter()
        n = 64
        for _ in range(n):
            s.infer_next_token()
        L.ggml_hip_synchronize()
        dt = time.perf_counter() - t0
        print(f"== big={big}: decode {n / dt:.1f} tok/s ({dt / n * 1e3:.4f} ms/token) ==")
        print("  kind      bytes/launch | probe1  probe2  probe3  probe4  probe5   full  (us per launch incl. boundary) | full GB/s")
        rows = {}
        for probe in (1, 2, 3, 4, 5, 0):
            ggml.set_option("probe", probe)
            s.infer_next_token()  # rebuilds the plan with the probe level (results are garbage for probe != 0)
            for k, nm in enumerate(kinds):
                ms, kn, kb = ggml.bench_plan_class(ggml.KKIND_BASE + k, 20)
                rows.setdefault(nm, {})[probe] = (ms * 1e3 / max(kn * 20, 1), kb / max(kn, 1))
        for nm in kinds:
            r = rows[nm]
            print(f"  {nm:8s} {int(r[0][1]):12d} | {r[1][0]:6.2f} {r[2][0]:7.2f} {r[3][0]:7.2f} {r[4][0]:7.2f} {r[5][0]:7.2f} {r[0][0]:6.2f}"
                  f"                 | {r[0][1] / 1e3 / r[0][0]:8.1f}")
        ms, kn, kb = ggml.bench_plan_class(ggml.KCLASS_MMVQ, 20)
        print(f"  all mat-vecs of a token: {ms / 20:.4f} ms, {kb * 20 / 1e9 / (ms / 1e3):.1f} GB/s incl. boundaries")
        ams, an, _ = ggml.bench_plan_class(ggml.KCLASS_ATTN, 20)
        print(f"  attention launches: {ams / 20:.4f} ms per token ({ams * 1e3 / 20 / max(an, 1):.2f} us each)")
        # the session's K/V now holds garbage positions from the probe tokens: rewind is not needed for timing
    ggml.set_option("probe", 0)
    ggml.set_option("big", 1)
    s.free()
    model.free()


if __name__ == "__main__":
    main()
