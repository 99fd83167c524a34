"""One prompt-GEMM shape, repeated, for rocprofv3: python tests/tools/gemm_one.py M K N reps [opt=value ...]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from llm_amd import ggml as G

M, K, N, reps = (int(v) for v in sys.argv[1:5])
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    G.set_option(k, int(v))
L = G.lib()
nblk = M * K // 32
raw = np.empty(nblk * 18, dtype=np.uint8)
fill = L.llm_synth_blocks
fill.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float]
fill.restype = None
fill(2, raw.ctypes.data, nblk, 99, 0.0043)
X = np.random.default_rng(1).standard_normal((N, K)).astype(np.float32)
with G.Context(raw.nbytes + X.nbytes + M * N * 4 + (1 << 20)) as ctx:
    w = ctx.tensor_from(raw, 2, (K, M)).set_name("w")
    w.transfer_to_gpu()
    x = ctx.tensor_from(X, G.TYPE_F32, (K, N)).set_name("x")
    y = ctx.op_mul_mat(w, x)
    g = ctx.graph().build_forward_expand(y)
    for _ in range(reps):
        g.compute()
print("done")
