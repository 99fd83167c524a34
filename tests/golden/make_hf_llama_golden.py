"""Generates tests/golden/hf_llama_tiny.npz: logits of Hugging Face transformers' LlamaForCausalLM (an
independent implementation of the architecture) on the dequantized synthetic tiny model.  Run in the build
container (needs torch + transformers; neither is needed to USE the fixture):
    python tests/golden/make_hf_llama_golden.py
The reference cannot serve this purpose: its arithmetic is an absent submodule and it needs Rust."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from llm_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402
from transformers import LlamaConfig, LlamaForCausalLM  # noqa: E402

WTYPE, SEED = 2, 1234
hp, w = synth.make_llama(synth.TINY, WTYPE, seed=SEED)
E, H, L, F, V = hp["n_embd"], hp["n_head"], hp["n_layer"], hp["n_ff"], hp["n_vocab"]
D = E // H


def deq(name, rows, cols):
    return O.dequantize(WTYPE, w[name], rows * cols).reshape(rows, cols)


def to_hf_rope_layout(m):
    """ggml mode-0 RoPE rotates adjacent pairs (2k,2k+1); HF rotates (k, k+D/2): permute each head's rows."""
    idx = np.concatenate([np.arange(0, D, 2), np.arange(1, D, 2)])
    return m.reshape(H, D, -1)[:, idx, :].reshape(H * D, -1)


cfg = LlamaConfig(vocab_size=V, hidden_size=E, intermediate_size=F, num_hidden_layers=L, num_attention_heads=H,
                  num_key_value_heads=H, rms_norm_eps=5e-6, rope_theta=10000.0, max_position_embeddings=64,
                  attention_bias=False, mlp_bias=False, tie_word_embeddings=False, attn_implementation="eager")
model = LlamaForCausalLM(cfg).to(torch.float32).eval()
sd = {"model.embed_tokens.weight": deq("tok_embeddings.weight", V, E), "model.norm.weight": w["norm.weight"],
      "lm_head.weight": deq("output.weight", V, E)}
for i in range(L):
    p, q = f"layers.{i}.", f"model.layers.{i}."
    sd[q + "input_layernorm.weight"] = w[p + "attention_norm.weight"]
    sd[q + "post_attention_layernorm.weight"] = w[p + "ffn_norm.weight"]
    sd[q + "self_attn.q_proj.weight"] = to_hf_rope_layout(deq(p + "attention.wq.weight", E, E))
    sd[q + "self_attn.k_proj.weight"] = to_hf_rope_layout(deq(p + "attention.wk.weight", E, E))
    sd[q + "self_attn.v_proj.weight"] = deq(p + "attention.wv.weight", E, E)
    sd[q + "self_attn.o_proj.weight"] = deq(p + "attention.wo.weight", E, E)
    sd[q + "mlp.gate_proj.weight"] = deq(p + "feed_forward.w1.weight", F, E)
    sd[q + "mlp.down_proj.weight"] = deq(p + "feed_forward.w2.weight", E, F)
    sd[q + "mlp.up_proj.weight"] = deq(p + "feed_forward.w3.weight", F, E)
missing = model.load_state_dict({k: torch.tensor(np.ascontiguousarray(v)) for k, v in sd.items()}, strict=False)
assert not [k for k in missing.missing_keys if "rotary" not in k], missing
toks = np.random.default_rng(42).integers(0, V, 12).astype(np.int64)
with torch.no_grad():
    logits = model(torch.tensor(toks)[None]).logits[0].numpy().astype(np.float32)
np.savez_compressed(os.path.join(os.path.dirname(__file__), "hf_llama_tiny.npz"), wtype=WTYPE, seed=SEED,
                    tokens=toks.astype(np.int32), logits=logits)
orc = O.Llama(hp, w, 32)
got = orc.evaluate(toks.astype(np.int32), mode=1)
print("oracle(math) vs HF: max|d|/std =", float(np.max(np.abs(got - logits)) / logits.std()),
      "argmax equal:", bool((got.argmax(-1) == logits.argmax(-1)).all()))
