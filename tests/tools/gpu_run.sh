cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_fullsize_gpu.py -q -m gpu 2>&1 | tail -3
python - <<'PY'
import time, numpy as np
from llm_amd import ggml as G, llama, synth
hp, w = synth.make_llama_fast(synth.LLAMA_7B, G.TYPE_Q4_0)
model = llama.Llama(hp, w, context_size=2048)
prompt = np.random.default_rng(42).integers(0, hp["n_vocab"], 128).astype(np.int32)
for rep in range(3):
    s = model.start_session(n_batch=8)
    G.lib().ggml_hip_synchronize(); t = time.perf_counter()
    s.feed_prompt(prompt)
    G.lib().ggml_hip_synchronize(); dt = time.perf_counter() - t
    s.free()
print(f"128-token prompt at n_batch=8: {dt*1e3:.1f} ms = {128/dt:.0f} tok/s")
PY
