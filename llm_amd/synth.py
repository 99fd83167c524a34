"""Synthetic GGML-format LLaMA weights: N(0, 0.02²) f32 matrices quantized with the
product's own ggml_quantize_q* (the function the reference's `llm quantize` calls,
crates/llm-base/src/quantize.rs:363-379), norm weights 1 + N(0, 0.01²) kept f32 (1-D tensors are never
quantized, quantize.rs:332-335).  Tensor names / dims follow crates/models/llama/src/lib.rs:52-91:
2-D weights are [in_features (ne0), out_features (ne1)]."""
import ctypes

import numpy as np

from . import ggml

LLAMA_7B = dict(n_vocab=32000, n_embd=4096, n_head=32, n_head_kv=32, n_layer=32, n_rot=128, n_ff=11008, n_mult=256)
LLAMA_13B = dict(n_vocab=32000, n_embd=5120, n_head=40, n_head_kv=40, n_layer=40, n_rot=128, n_ff=13824, n_mult=256)
LLAMA_65B = dict(n_vocab=32000, n_embd=8192, n_head=64, n_head_kv=64, n_layer=80, n_rot=128, n_ff=22016, n_mult=256)
TINY = dict(n_vocab=256, n_embd=128, n_head=4, n_head_kv=4, n_layer=2, n_rot=32, n_ff=352, n_mult=32)


def tensor_shapes(hp):
    """name -> (ne0, ne1 or None)"""
    E, F, V = hp["n_embd"], hp["n_ff"], hp["n_vocab"]
    Egqa = E // (hp["n_head"] // hp["n_head_kv"])
    s = {"tok_embeddings.weight": (E, V), "norm.weight": (E, None), "output.weight": (E, V)}
    for i in range(hp["n_layer"]):
        p = f"layers.{i}."
        s[p + "attention_norm.weight"] = (E, None)
        s[p + "attention.wq.weight"] = (E, E)
        s[p + "attention.wk.weight"] = (E, Egqa)
        s[p + "attention.wv.weight"] = (E, Egqa)
        s[p + "attention.wo.weight"] = (E, E)
        s[p + "ffn_norm.weight"] = (E, None)
        s[p + "feed_forward.w1.weight"] = (E, F)
        s[p + "feed_forward.w2.weight"] = (F, E)
        s[p + "feed_forward.w3.weight"] = (E, F)
    return s


def stage_tensor_names(hp, layer_begin, layer_end):
    """Tensors a layer-split stage owns: its layers, + tok_embeddings on the first, + norm/output on the last."""
    names = set()
    for name in tensor_shapes(hp):
        if name.startswith("layers."):
            if layer_begin <= int(name.split(".")[1]) < layer_end:
                names.add(name)
        elif name == "tok_embeddings.weight":
            if layer_begin == 0:
                names.add(name)
        elif layer_end == hp["n_layer"]:
            names.add(name)
    return names


def _layer_of(name):
    return int(name.split(".")[1]) if name.startswith("layers.") else -1


def make_llama(hp, wtype, seed=1234, std=0.02):
    """Returns (hp_with_wtype, {name: np.ndarray}) — 2-D weights as raw GGML bytes (uint8), norms f32.
    Exact path: gaussian f32 -> ggml_quantize_q*.  Use for parity tests (small models)."""
    out = {}
    for name, (ne0, ne1) in tensor_shapes(hp).items():
        rng = np.random.default_rng([seed, _layer_of(name) + 1, sum(map(ord, name))])
        if ne1 is None:
            out[name] = (1.0 + 0.01 * rng.standard_normal(ne0)).astype(np.float32)
        else:
            w = (std * rng.standard_normal((ne1, ne0))).astype(np.float32)
            out[name] = ggml.quantize(wtype, w)
    h = dict(hp)
    h["wtype"] = wtype
    return h, out


def make_llama_fast(hp, wtype, seed=1234, d_scale=0.0043, only=None):
    """Full-size synthetic weights for bench.py: writes random GGML blocks directly (uniform quants, f16
    scales around `d_scale` so that dequantized weights have std ≈ 0.02) instead of quantizing 6.7e9
    gaussians.  Same container format, same bytes-per-weight, valid for every block type."""
    out = {}
    bs, be = ggml.BLOCK_BYTES[wtype], ggml.BLOCK_ELEMS[wtype]
    for name, (ne0, ne1) in tensor_shapes(hp).items():
        if only is not None and name not in only:
            continue
        rng = np.random.default_rng([seed, _layer_of(name) + 1, sum(map(ord, name))])
        if ne1 is None:
            out[name] = (1.0 + 0.01 * rng.standard_normal(ne0)).astype(np.float32)
            continue
        nblk = ne1 * (ne0 // be)
        # per-type scale so that dequantized std stays ≈ 0.02: q4 std≈4.6, q5 std≈9.2, q8 std≈74
        sc = {ggml.TYPE_Q4_0: d_scale, ggml.TYPE_Q4_1: d_scale, ggml.TYPE_Q5_0: d_scale / 2,
              ggml.TYPE_Q5_1: d_scale / 2, ggml.TYPE_Q8_0: d_scale / 16, ggml.TYPE_Q4_K: d_scale, ggml.TYPE_Q6_K: d_scale,
              ggml.TYPE_Q2_K: d_scale, ggml.TYPE_Q3_K: d_scale, ggml.TYPE_Q5_K: d_scale}[wtype]
        raw = np.empty(nblk * bs, dtype=np.uint8)
        fill = ggml.lib().llm_synth_blocks
        fill.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float]
        fill.restype = None
        fill(wtype, raw.ctypes.data, nblk, int(rng.integers(0, 2**62)), sc)
        out[name] = raw.reshape(-1)
    h = dict(hp)
    h["wtype"] = wtype
    return h, out


def make_llama_gaussian(hp, wtype, seed=1234, std=0.02, only=None):
    """BASELINE.md section 4 at full size: 2-D weights N(0, std^2) quantized by the product's ggml_quantize_q* (the host
    function the reference's quantizer calls; byte-identical to the oracle's), norm weights 1 + N(0, 0.01^2).  The
    gaussians of the big tensors come from a counter-based generator in the library (llm_synth_gaussian, threaded):
    numpy's default_rng stream for 6.6e9 draws would take minutes before a bench could start."""
    out = {}
    bs, be = ggml.BLOCK_BYTES[wtype], ggml.BLOCK_ELEMS[wtype]
    fill = ggml.lib().llm_synth_gaussian
    fill.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float]
    fill.restype = None
    for name, (ne0, ne1) in tensor_shapes(hp).items():
        if only is not None and name not in only:
            continue
        rng = np.random.default_rng([seed, _layer_of(name) + 1, sum(map(ord, name))])
        if ne1 is None:
            out[name] = (1.0 + 0.01 * rng.standard_normal(ne0)).astype(np.float32)
            continue
        raw = np.empty(ne1 * (ne0 // be) * bs, dtype=np.uint8)
        fill(wtype, raw.ctypes.data, ne0, ne1, int(rng.integers(0, 2**62)), std)
        out[name] = raw
    h = dict(hp)
    h["wtype"] = wtype
    return h, out


def write_ggjt(path, hp, w, container="ggjt", version=3, vocab=None):
    """Writes a LLaMA model file the way crates/ggml/src/format/saver.rs:86-160 does: magic (+ version), the
    hyperparameters of models/llama/src/lib.rs:449-458, the vocabulary (u32 len, bytes, f32 score — no score in the
    legacy 'ggml' container), then per tensor (i32 n_dims, i32 name_len, u32 type, i32 dims[], name, padding to a
    32-byte boundary for ggjt, data).  `w`: the dict of make_llama*(): raw GGML bytes for 2-D weights, f32 for 1-D.  hp["wtypes"] (optional): {tensor name: ggml type}
    for files that mix types, as the *_K_S / *_K_M quantizations do (the tensor records carry their own type)."""
    import struct
    from . import ggml
    magic = {"ggml": 0x67676d6c, "ggmf": 0x67676d66, "ggjt": 0x67676a74}[container]
    shapes = tensor_shapes(hp)
    with open(path, "wb") as f:
        f.write(struct.pack("<I", magic))
        if container != "ggml":
            f.write(struct.pack("<I", version))
        ftype = 2 * 1000 + ggml.FTYPE_OF[hp["wtype"]]  # crates/llm-base/src/loader.rs:32-36
        f.write(struct.pack("<7i", hp["n_vocab"], hp["n_embd"], hp.get("n_mult", 256), hp["n_head"], hp["n_layer"],
                            hp["n_rot"], ftype))
        for i in range(hp["n_vocab"]):
            tok, score = vocab[i] if vocab else (f"<{i}>".encode(), -float(i))
            f.write(struct.pack("<I", len(tok)) + tok)
            if container != "ggml":
                f.write(struct.pack("<f", score))
        for name, (ne0, ne1) in shapes.items():
            nb = name.encode()
            typ = ggml.TYPE_F32 if ne1 is None else hp.get("wtypes", {}).get(name, hp["wtype"])  # per-tensor types of a mixed file
            dims = (ne0,) if ne1 is None else (ne0, ne1)
            f.write(struct.pack("<iiI", len(dims), len(nb), typ))
            f.write(struct.pack(f"<{len(dims)}i", *dims))
            f.write(nb)
            if container == "ggjt":
                pos = f.tell()
                f.write(b"\0" * (((pos + 31) & ~31) - pos))
            f.write(np.ascontiguousarray(w[name]).tobytes())
