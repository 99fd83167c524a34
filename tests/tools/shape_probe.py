import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from llm_amd import ggml as G
from oracle import oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from test_ops_gpu import _mul_mat_gpu, _abs_scale
for wtype in (7, 8, 2):
    for (M, K) in ((64, 5120), (64, 13824), (64, 8192), (64, 22016), (64, 11008)):
        for N in (1, 6):
            rng = np.random.default_rng([wtype, M, K, N])
            W = (0.02 * rng.standard_normal((M, K))).astype(np.float32)
            X = rng.standard_normal((N, K)).astype(np.float32)
            W_raw = G.quantize(wtype, W)
            got = _mul_mat_gpu(G, wtype, W_raw, M, K, X)
            exact = O.mul_mat(wtype, W_raw, M, K, X, mode=0)
            scale = _abs_scale(O, wtype, W_raw, M, K, X)
            r = float(np.max(np.abs(got - exact) / (scale + 1e-12)))
            print(wtype, M, K, N, f"{r:.2e}", "OK" if r <= 2e-5 else "BAD")
