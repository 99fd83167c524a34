import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from llm_amd import ggml as G, synth
from oracle import oracle as O
wtype = int(sys.argv[1]) if len(sys.argv) > 1 else 6
hp, w = synth.make_llama(synth.TINY, wtype)
shapes = synth.tensor_shapes(hp)
rng = np.random.default_rng(0)
for name, (ne0, ne1) in shapes.items():
    if ne1 is None: continue
    X = rng.standard_normal((8, ne0)).astype(np.float32)
    with G.Context(w[name].nbytes + (1 << 20)) as ctx:
        tw = ctx.tensor_from(w[name], wtype, (ne0, ne1)).transfer_to_gpu()
        tx = ctx.tensor_from(X, G.TYPE_F32, (ne0, 8))
        y = ctx.op_mul_mat(tw, tx)
        ctx.graph().build_forward_expand(y).compute()
        got = y.read_data().reshape(8, ne1)
    ref = O.mul_mat(wtype, w[name], ne1, ne0, X, 0)
    bad = np.argwhere(np.abs(got - ref) > 1e-4 * np.abs(ref).max())
    print(f"{name:34s} max|d|={np.abs(got-ref).max():.2e} n_bad={len(bad)} rows={sorted(set(bad[:,1].tolist()))[:10]}")
    if len(bad):
        m = bad[0][1]; rb = O.row_bytes(wtype, ne0)
        row = w[name][m*rb:(m+1)*rb].reshape(-1, 22)
        print("   first bad row", m, "d(f16 bits)=", [hex(int.from_bytes(bytes(b[0:2]), 'little')) for b in row], "qh=", [hex(int.from_bytes(bytes(b[2:6]),'little')) for b in row])
