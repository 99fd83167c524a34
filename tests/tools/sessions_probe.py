#!/usr/bin/env python
"""Aggregate decode rate of N InferenceSessions running concurrently on ONE GPU (one thread each; the reference's "several
sessions on one model", crates/llm-base/src/inference_session.rs:43-48), against one session alone.
    python tests/tools/sessions_probe.py [n_sessions ...]      (env SESSIONS_SHARED=0: one model copy per virtual slot)
With SESSIONS_SHARED=1 (default where the library exports llm_start_session_on) the sessions share ONE model: every session
gets its own device slot (stream, shadows, plan workspace) on the same GPU and reads the weights of the model's slot."""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))


def main():
    counts = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
    nmax = max(counts)
    os.environ["GGML_HIP_VIRTUAL_DEVICES"] = str(max(nmax, 1))
    from llm_amd import ggml, llama, synth
    L = ggml.lib()
    shared = os.environ.get("SESSIONS_SHARED", "1") != "0" and hasattr(llama.Llama, "start_session_on")
    hp, w = synth.make_llama_fast(synth.LLAMA_7B, ggml.TYPE_Q4_0)
    models = []
    if shared:
        L.ggml_hip_set_main_device(0)
        models = [llama.Llama(hp, w, context_size=2048)]
    else:
        for slot in range(nmax):
            L.ggml_hip_set_main_device(slot)
            models.append(llama.Llama(hp, w, context_size=2048))
        L.ggml_hip_set_main_device(0)
    prompt = (np.arange(128, dtype=np.int32) * 7 + 5) % hp["n_vocab"]
    steps = int(os.environ.get("SESSIONS_STEPS", "192"))
    out = {"shared_model": shared, "steps_per_session": steps, "runs": []}
    for n in counts:
        sess = [(models[0].start_session_on(i, n_batch=8) if shared else models[i].start_session(n_batch=8)) for i in range(n)]
        for s in sess:
            s.feed_prompt(prompt)
            for _ in range(8):
                s.infer_next_token()
        for i in range(n):
            L.ggml_hip_bind_thread_device(i)
            L.ggml_hip_synchronize()
        L.ggml_hip_bind_thread_device(0)
        start = threading.Barrier(n + 1)
        lat = [None] * n
        ids = [None] * n

        def run(i):
            start.wait()
            t0 = time.perf_counter()
            ids[i] = [sess[i].infer_next_token() for _ in range(steps)]
            lat[i] = (time.perf_counter() - t0) / steps

        th = [threading.Thread(target=run, args=(i,)) for i in range(n)]
        for t in th:
            t.start()
        start.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        same = all(x == ids[0] for x in ids)
        out["runs"].append({"sessions": n, "aggregate_tokens_per_s": round(n * steps / el, 1),
                            "per_session_ms_per_token": [round(x * 1e3, 4) for x in lat], "all_sessions_same_ids": same,
                            "fused_attn_timeouts": int(L.ggml_hip_get_stat(b"fused_attn_timeouts"))})
        print(json.dumps(out["runs"][-1]), flush=True)
        for s in sess:
            s.free()
    base = out["runs"][0]["aggregate_tokens_per_s"] if out["runs"] and out["runs"][0]["sessions"] == 1 else None
    if base:
        for r in out["runs"]:
            r["vs_one_session"] = round(r["aggregate_tokens_per_s"] / base, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
