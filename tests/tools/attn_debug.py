"""Fused prompt attention (kernels/prompt_attn.h) against the three-launch path on random data: where do they differ?"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from llm_amd import ggml as G
L = G.lib()
f = L.ggml_hip_debug_prompt_attention
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_float, C.c_int]

def run(N, H, Hkv, D, n_past, Cc, fused, seed=0):
    rng = np.random.default_rng(seed)
    E, Eg = H * D, Hkv * D
    q = rng.standard_normal((N, E)).astype(np.float32)
    k = np.zeros((Cc, Eg), np.float16); v = np.zeros((Eg, Cc), np.float16)
    T = n_past + N
    k[:T] = rng.standard_normal((T, Eg)).astype(np.float16)
    v[:, :T] = rng.standard_normal((Eg, T)).astype(np.float16)
    out = np.zeros((N, E), np.float32)
    rc = f(q.ctypes.data, k.ctypes.data, v.ctypes.data, out.ctypes.data, N, E, Eg, H, n_past, Cc, 1.0 / np.sqrt(D), fused)
    return rc, out

for (N, H, Hkv, D, n_past) in [(64, 4, 4, 32, 0), (128, 4, 4, 32, 100), (128, 4, 4, 32, 128), (128, 4, 4, 32, 129), (128, 4, 4, 32, 230), (32, 4, 4, 32, 300),
                               (33, 2, 1, 64, 500), (512, 8, 8, 128, 1), (200, 4, 2, 128, 700), (96, 4, 4, 32, 1000)]:
    Cc = 2048
    ra, a = run(N, H, Hkv, D, n_past, Cc, 1)
    rb, b = run(N, H, Hkv, D, n_past, Cc, 0)
    bad = np.argwhere(a != b)
    msg = f"N={N} H={H} Hkv={Hkv} D={D} n_past={n_past}: rc {ra},{rb} equal={np.array_equal(a, b)} max|d|={np.max(np.abs(a - b)):.3e} nan={np.isnan(a).sum()},{np.isnan(b).sum()}"
    if len(bad):
        rows = np.unique(bad[:, 0]); cols = np.unique(bad[:, 1])
        msg += f"  differing rows {rows[:8]}..{rows[-1]} ({len(rows)}), cols {cols[:6]}..{cols[-1]} ({len(cols)})"
    print(msg, flush=True)
