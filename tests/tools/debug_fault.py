import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from llm_amd import ggml as G
E, F, C = 128, 352, 16
rng = np.random.default_rng(0)
wtype = 2
def qw(M, K):
    return G.quantize(wtype, (0.02 * rng.standard_normal((M, K))).astype(np.float32))
sctx = G.Context(1 << 22); wctx = G.Context(1 << 22); c = G.Context(1 << 24)
print("ctx ok", flush=True)
mk = sctx.tensor_from(np.zeros(C * E, np.float16), G.TYPE_F16); print("mk created", flush=True)
mk.transfer_to_gpu(); print("mk transferred", flush=True)
mv = sctx.tensor_from(np.zeros(C * E, np.float16), G.TYPE_F16).transfer_to_gpu(); print("mv transferred", flush=True)
for name, (M, K) in dict(wq=(E, E), wk=(E, E), w1=(F, E), w2=(E, F)).items():
    t = wctx.tensor_from(qw(M, K), wtype, (K, M)); print(name, "created", t.ne, t.nbytes(), hex(t.t.data), flush=True)
    t.transfer_to_gpu(); print(name, "transferred", flush=True)
an = wctx.tensor_from(np.ones(E, np.float32)).transfer_to_gpu(); print("an transferred", flush=True)
G.lib().ggml_hip_synchronize(); print("done", flush=True)
