"""Generates tests/golden/oracle_golden.npz: small committed outputs of the oracle itself (codec bytes, exact
mat-mul, a tiny LLaMA run), so that edits to oracle/ggml_oracle.c that change results are caught by CPU tests.
    python tests/golden/make_oracle_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from llm_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

rng = np.random.default_rng(2024)
out = {}
x = (0.02 * rng.standard_normal((24, 128))).astype(np.float32)
x[0, :32] = 0
x[1, 7] = 1.5
out["quant_input"] = x
out["mm_x"] = rng.standard_normal((3, 128)).astype(np.float32)
for t in (2, 3, 6, 7, 8):
    out[f"quant_{t}"] = O.quantize(t, x)
    out[f"mm_exact_{t}"] = O.mul_mat(t, out[f"quant_{t}"], 24, 128, out["mm_x"], mode=0)
hp, w = synth.make_llama(synth.TINY, 2)
toks = np.random.default_rng(42).integers(0, hp["n_vocab"], 10).astype(np.int32)
lg = O.Llama(hp, w, 32).evaluate(toks, mode=0)
out["llama_tokens"] = toks
out["llama_logits_exact_q4_0"] = lg
out["llama_argmax_q4_0"] = np.argmax(lg, -1)
np.savez_compressed(os.path.join(os.path.dirname(__file__), "oracle_golden.npz"), **out)
print("wrote oracle_golden.npz", {k: v.shape for k, v in out.items()})
