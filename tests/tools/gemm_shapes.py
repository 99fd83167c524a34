"""Per-shape timing of the prompt GEMM kernels at the LLaMA-7B shapes of a 512-token batch (one mul_mat node each through
ggml_graph_compute; per-launch HIP events of the backend).  python tests/tools/gemm_shapes.py [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from llm_amd import ggml as G

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
SHAPES = [("wq|wk|wv", 12288, 4096, 2), ("wo", 4096, 4096, 2), ("w1|w3", 22016, 4096, 1), ("w2", 4096, 11008, 2),
          ("lm_head", 32000, 4096, 1)]
if len(sys.argv) > 2 and sys.argv[2] == "big":
    SHAPES = [("4096^3", 4096, 4096, 1), ("8192x4096", 8192, 4096, 1)]
# (round 3 also measured two variants of the 256-tile kernel — counted wait one phase later, no s_setprio — both neutral;
# their instantiations and the option that selected them are gone)
VARIANTS = (("t256", {"mmq_t256": 2}), ("w16_128", {"mmq_t256": 0}), ("dma_p8", {"mmq_t256": 0, "mmq_w16": 0}))
RESET = {"mmq_t256": 1, "mmq_w16": 1}
L = G.lib()
rng = np.random.default_rng(1)
for name, M, K, sp in SHAPES:
    nblk = M * K // 32
    raw = np.empty(nblk * 18, dtype=np.uint8)
    import ctypes
    fill = L.llm_synth_blocks
    fill.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float]
    fill.restype = None
    fill(2, raw.ctypes.data, nblk, 1234 + M, 0.0043)
    X = rng.standard_normal((N, K)).astype(np.float32)
    mem = raw.nbytes + X.nbytes + M * N * 4 + (1 << 20)
    with G.Context(mem) as ctx:
        w = ctx.tensor_from(raw, 2, (K, M)).set_name("w")
        w.transfer_to_gpu()
        x = ctx.tensor_from(X, G.TYPE_F32, (K, N)).set_name("x")
        y = ctx.op_mul_mat(w, x)
        g = ctx.graph().build_forward_expand(y)
        line = f"{name:9s} M={M:6d} K={K:6d} N={N} splits(rule)={sp}:"
        for label, opts in VARIANTS:
            for k, v in opts.items():
                G.set_option(k, v)
            best = 1e9
            for it in range(6):
                L.ggml_hip_timing_begin()
                g.compute()
                L.ggml_hip_timing_end()
                ms, launches, flops = G.timing_query(G.KCLASS_MMQ_MFMA)
                if it >= 1:
                    best = min(best, ms)
            for k in opts:
                G.set_option(k, RESET[k])
            line += f"\n      {label:15s} {best * 1e3:7.1f} us {2.0 * M * N * K / best / 1e9:7.1f} TF"
        print(line, flush=True)
