#!/usr/bin/env python
"""Yardstick for the prompt GEMMs (NOT part of the product): what the vendor library behind torch.matmul (hipBLASLt / rocBLAS)
reaches on the five f16 x f16 -> f32-accumulate shapes of a 512-token LLaMA-7B batch, timed like bench.py's prefill leg (HIP
events around back-to-back launches over distinct weight buffers, so that a weight is not served from the last level cache).
    python tests/tools/gemm_yardstick.py [tokens]"""
import json
import sys

import torch

T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
SHAPES = {"wq|wk|wv": (12288, 4096), "wo": (4096, 4096), "w1|w3": (22016, 4096), "w2": (4096, 11008), "lm_head": (32000, 4096)}
OURS_US = {"wq|wk|wv": 55.0, "wo": 31.0, "w1|w3": 107.0, "w2": 60.0, "lm_head": 131.0}  # DESIGN.md section 4 "Round 3: the 256-tile GEMM"
dev = torch.device("cuda:0")
out = {"tokens": T, "shapes": {}}
tot_lib = tot_ours = 0.0
for name, (n, k) in SHAPES.items():
    copies = 8
    w = [torch.randn(n, k, device=dev, dtype=torch.float16) * 0.02 for _ in range(copies)]
    x = torch.randn(T, k, device=dev, dtype=torch.float16)
    for i in range(copies):
        y = x @ w[i].t()
    torch.cuda.synchronize()
    reps = 40
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        y = x @ w[i % copies].t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * T * n * k
    per_layer = 1 if name == "lm_head" else 32
    tot_lib += us * per_layer
    tot_ours += OURS_US[name] * per_layer
    out["shapes"][name] = {"n": n, "k": k, "library_us": round(us, 1), "library_PFLOPs": round(fl / us / 1e9, 3),
                           "this_repo_us": OURS_US[name] if T == 512 else None,
                           "this_repo_PFLOPs": round(fl / OURS_US[name] / 1e9, 3) if T == 512 else None}
    del w
out["per_batch_ms"] = {"library": round(tot_lib / 1e3, 2), "this_repo": round(tot_ours / 1e3, 2) if T == 512 else None}
print(json.dumps(out, indent=1))
