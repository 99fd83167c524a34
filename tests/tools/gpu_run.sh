cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python - <<'PY'
import time, numpy as np
from llm_amd import ggml, llama, synth
hp, w = synth.make_llama_fast(synth.LLAMA_7B, ggml.TYPE_Q4_0)
model = llama.Llama(hp, w, context_size=2048)
toks = (np.arange(1900, dtype=np.int32) * 7 + 5) % hp["n_vocab"]
for n0 in (200, 330, 460, 590):
    for split in (0, 64):
        ggml.set_option("attn_split", split)
        s = model.start_session(n_batch=512)
        s.feed_prompt(toks[:n0])
        for _ in range(4): s.infer_next_token()
        ggml.lib().ggml_hip_synchronize(); t0 = time.perf_counter()
        for _ in range(48): s.infer_next_token()
        ggml.lib().ggml_hip_synchronize(); dt = time.perf_counter() - t0
        print("n_past %4d split %d: %.1f tok/s (%.3f ms)" % (n0, split, 48 / dt, dt / 48 * 1e3))
        s.free()
PY
