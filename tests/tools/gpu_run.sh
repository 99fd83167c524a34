cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_llama_gpu.py tests/test_fullsize_gpu.py tests/test_entry_gpu.py -q -x 2>&1 | tail -3
for w in 16 0; do
GGML_HIP_BIG_WAVES=$w timeout 300 python - <<'PY'
import time, numpy as np, os
from llm_amd import ggml, llama, synth
hp, w = synth.make_llama_fast(synth.LLAMA_7B, ggml.TYPE_Q4_0)
model = llama.Llama(hp, w, context_size=2048)
s = model.start_session(n_batch=8)
toks = (np.arange(512, dtype=np.int32) * 7 + 5) % hp["n_vocab"]
s.feed_prompt(toks[:64]); ggml.lib().ggml_hip_synchronize()
t0 = time.perf_counter(); s.feed_prompt(toks[64:448]); ggml.lib().ggml_hip_synchronize(); dt = time.perf_counter() - t0
print("waves", os.environ.get("GGML_HIP_BIG_WAVES"), "prompt feed n_batch=8: %.0f tok/s (%.2f ms per 8-token chunk)" % (384 / dt, dt / 48 * 1e3))
PY
done
