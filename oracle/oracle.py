"""ctypes loader for the CPU oracle (oracle/ggml_oracle.c) — TEST INFRASTRUCTURE.

Modes of every matmul-bearing call: 0 = ggml's scalar code, 2 = the same in the arithmetic order of ggml's AVX2
branch (what the reference's build selects on x86-64), 1 = "math" (dequantized weights, f64 accumulation).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
PARITY UNPINNED: see the header of ggml_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libggml_oracle.so")

T_F32, T_F16, T_Q4_0, T_Q4_1, T_Q5_0, T_Q5_1, T_Q8_0, T_Q8_1 = 0, 1, 2, 3, 6, 7, 8, 9
T_Q2_K, T_Q3_K, T_Q4_K, T_Q5_K, T_Q6_K, T_Q8_K = 10, 11, 12, 13, 14, 15  # K-quants
QUANT_TYPES = (T_Q4_0, T_Q4_1, T_Q5_0, T_Q5_1, T_Q8_0)
TYPE_NAMES = {T_F32: "f32", T_F16: "f16", T_Q4_0: "q4_0", T_Q4_1: "q4_1", T_Q5_0: "q5_0", T_Q5_1: "q5_1",
              T_Q8_0: "q8_0", T_Q8_1: "q8_1"}


def build(force=False):
    src = os.path.join(_HERE, "ggml_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        # libgomp reads this at load time: idle team members sleep instead of spinning (containers often
        # expose more cpus than their quota lets them run, where spinning makes 8 threads slower than 1)
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        L = C.CDLL(_SO)
        L.orc_fp16_to_fp32.restype = C.c_float
        L.orc_fp16_to_fp32.argtypes = [C.c_uint16]
        L.orc_fp32_to_fp16.restype = C.c_uint16
        L.orc_fp32_to_fp16.argtypes = [C.c_float]
        L.orc_type_size.restype = C.c_int
        L.orc_blck_size.restype = C.c_int
        L.orc_vec_dot_type.restype = C.c_int
        L.orc_quantize.restype = C.c_size_t
        L.orc_quantize.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_quantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_dequantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_vec_dot.restype = C.c_float
        L.orc_vec_dot.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_vec_dot_simd.restype = C.c_float
        L.orc_vec_dot_simd.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_quantize_row_simd.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_vec_dot_avx2.restype = C.c_float
        L.orc_vec_dot_avx2.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_quantize_row_avx2.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_have_avx2.restype = C.c_int
        L.orc_mul_mat.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                  C.c_void_p, C.c_int]
        L.orc_rms_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float]
        L.orc_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
        L.orc_mul_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
        L.orc_silu.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.orc_gelu.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.orc_rope.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float]
        L.orc_scale_mask_softmax.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_int, C.c_int]
        L.orc_soft_max.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int]
        L.orc_fp32_to_fp16_row.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_fp16_to_fp32_row.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_llama_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_set_block_order.argtypes = [C.c_int]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        # bound the OpenMP team: tiny parity cases on a 100+-core host spend their time in barrier spin
        L.orc_set_num_threads(int(os.environ.get("ORACLE_THREADS", min(os.cpu_count() or 1, 16))))
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def type_size(t):
    return lib().orc_type_size(t)


def blck_size(t):
    return lib().orc_blck_size(t)


def row_bytes(t, k):
    return k // blck_size(t) * type_size(t)


def quantize(t, x, k=None):
    """ggml_quantize_q*: x f32 [..., k] -> raw block bytes (uint8 1-D)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    k = x.shape[-1] if k is None else k
    n = x.size
    out = np.zeros(n // blck_size(t) * type_size(t), dtype=np.uint8)
    hist = np.zeros(16, dtype=np.int64)
    if t == T_F32:
        return x.view(np.uint8).reshape(-1).copy()
    if t == T_F16:
        return x.astype(np.float16).view(np.uint8).reshape(-1).copy()
    got = lib().orc_quantize(t, _p(x), _p(out), n, k, _p(hist))
    assert got == out.size
    return out


def quantize_row(t, x, simd=False):
    """quantize_row_q*; simd=True: the AVX2 branch's arithmetic for Q8_0 / Q8_1 (id = 127/amax, round-half-even),
    i.e. the activation quantizer of oracle mode 2; simd="avx2": the same with the intrinsics (mode 3)."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    out = np.zeros(row_bytes(t, x.size), dtype=np.uint8)
    f = lib().orc_quantize_row_avx2 if simd == "avx2" else lib().orc_quantize_row_simd if simd else lib().orc_quantize_row
    f(t, _p(x), _p(out), x.size)
    return out


MODE_SCALAR, MODE_MATH, MODE_AVX2, MODE_AVX2_INTRINSICS = 0, 1, 2, 3  # see the header of ggml_oracle.c


def have_avx2():
    """True when mode 3 really runs the AVX2 / FMA / F16C intrinsics (built with them and the host has them)."""
    return bool(lib().orc_have_avx2())


def ref_mode():
    """The mode that restates what the reference's build EXECUTES on an x86-64 host: crates/ggml/sys/build.rs:46-62 passes
    -mavx2 -mfma -mf16c, so ggml's `#elif defined(__AVX2__)` branches run — mode 3 (the intrinsics) where this host has them,
    else mode 2 (the same arithmetic order in scalar code; bit-identical, tests/test_oracle.py).  The device's default
    activation quantizer (option act_quant = 0) follows this branch; the parity tests compare with this mode."""
    return MODE_AVX2_INTRINSICS if have_avx2() else MODE_AVX2


def dequantize(t, raw, n):
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    y = np.zeros(n, dtype=np.float32)
    lib().orc_dequantize_row(t, _p(raw), _p(y), n)
    return y


def mul_mat(t, A_raw, M, K, B, mode=0):
    """A_raw: raw bytes of [M rows of K] in type t.  B: f32 [N, K].  Returns f32 [N, M]."""
    A_raw = np.ascontiguousarray(A_raw)
    B = np.ascontiguousarray(B, dtype=np.float32)
    N = B.shape[0]
    dst = np.zeros((N, M), dtype=np.float32)
    lib().orc_mul_mat(t, _p(A_raw), M, K, _p(B), N, K, _p(dst), mode)
    return dst


def rms_norm(x, eps=5e-6):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib().orc_rms_norm(_p(x), _p(y), x.shape[-1], x.size // x.shape[-1], eps)
    return y


def norm(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib().orc_norm(_p(x), _p(y), x.shape[-1], x.size // x.shape[-1])
    return y


def silu(x, mode=0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib().orc_silu(_p(x), _p(y), x.size, mode)
    return y


def gelu(x, mode=0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib().orc_gelu(_p(x), _p(y), x.size, mode)
    return y


def rope(x, n_past, n_dims, freq_base=10000.0, freq_scale=1.0):
    """x: f32 [N, n_head, ne0] (numpy order; ggml [ne0, n_head, N]).  Returns rotated copy."""
    y = np.ascontiguousarray(x, dtype=np.float32).copy()
    N, H, D = y.shape
    lib().orc_rope(_p(y), D, H, N, n_past, n_dims, freq_base, freq_scale)
    return y


def scale_mask_softmax(x, scale, n_past, mode=0):
    """x: f32 [n_head, N, nc]."""
    y = np.ascontiguousarray(x, dtype=np.float32).copy()
    H, N, nc = y.shape
    lib().orc_scale_mask_softmax(_p(y), nc, N, H, scale, n_past, mode)
    return y


def soft_max(x, mode=0):
    y = np.ascontiguousarray(x, dtype=np.float32).copy()
    lib().orc_soft_max(_p(y), y.shape[-1], y.size // y.shape[-1], mode)
    return y


class _LlamaC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_vocab", "n_embd", "n_head", "n_head_kv", "n_layer", "n_rot", "n_ff", "n_ctx", "wtype")] + [
        ("rms_eps", C.c_float), ("freq_base", C.c_float), ("freq_scale", C.c_float),
        ("tok_embeddings", C.c_void_p), ("norm", C.c_void_p), ("output", C.c_void_p),
        ("attention_norm", C.c_void_p), ("wq", C.c_void_p), ("wk", C.c_void_p), ("wv", C.c_void_p),
        ("wo", C.c_void_p), ("ffn_norm", C.c_void_p), ("w1", C.c_void_p), ("w2", C.c_void_p), ("w3", C.c_void_p),
        ("memory_k", C.c_void_p), ("memory_v", C.c_void_p)]


class _Taps(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("inpL0", "layer0_attn_norm", "layer0_q", "layer0_kq", "layer0_out", "final_norm", "layer_out_all",
                 "inp_override")]


class Llama:
    """Oracle-side LLaMA session over a dict of raw weight arrays (see llm_amd.synth.make_llama)."""

    def __init__(self, hp, weights, n_ctx):
        self.hp = dict(hp)
        self.w = weights  # keep arrays alive
        self.n_ctx = n_ctx
        L = hp["n_layer"]
        egqa = hp["n_embd"] // (hp["n_head"] // hp["n_head_kv"])
        self.memory_k = np.zeros(L * n_ctx * egqa, dtype=np.uint16)
        self.memory_v = np.zeros(L * n_ctx * egqa, dtype=np.uint16)
        self.n_past = 0

        def arr(fmt):
            a = (C.c_void_p * L)()
            for i in range(L):
                a[i] = weights[fmt.format(i)].ctypes.data
            return a

        self._arrs = {k: arr(f) for k, f in {
            "attention_norm": "layers.{}.attention_norm.weight", "wq": "layers.{}.attention.wq.weight",
            "wk": "layers.{}.attention.wk.weight", "wv": "layers.{}.attention.wv.weight",
            "wo": "layers.{}.attention.wo.weight", "ffn_norm": "layers.{}.ffn_norm.weight",
            "w1": "layers.{}.feed_forward.w1.weight", "w2": "layers.{}.feed_forward.w2.weight",
            "w3": "layers.{}.feed_forward.w3.weight"}.items()}
        m = _LlamaC()
        for k in ("n_vocab", "n_embd", "n_head", "n_head_kv", "n_layer", "n_rot", "n_ff"):
            setattr(m, k, hp[k])
        m.n_ctx = n_ctx
        m.wtype = hp["wtype"]
        m.rms_eps = 5e-6
        m.freq_base = hp.get("freq_base", 10000.0)
        m.freq_scale = hp.get("freq_scale", 1.0)
        m.tok_embeddings = weights["tok_embeddings.weight"].ctypes.data
        m.norm = weights["norm.weight"].ctypes.data
        m.output = weights["output.weight"].ctypes.data
        for k, a in self._arrs.items():
            setattr(m, k, C.cast(a, C.c_void_p))
        m.memory_k = self.memory_k.ctypes.data
        m.memory_v = self.memory_v.ctypes.data
        self._m = m

    def evaluate(self, tokens, mode=0, taps=False, reverse_blocks=False, inp=None):
        """Feeds `tokens` at the current n_past; returns logits f32 [N, n_vocab] (+ taps dict).
        reverse_blocks: add each row's block terms in descending order (a legal re-association of ggml's f32
        block sum; the fwd-vs-rev distance is the yardstick for a model's own rounding sensitivity).
        inp: f32 [N, n_embd] that replaces the token embeddings (a layer stack fed with a given residual)."""
        lib().orc_set_block_order(1 if reverse_blocks else 0)
        try:
            return self._evaluate(tokens, mode, taps, inp)
        finally:
            lib().orc_set_block_order(0)

    def _evaluate(self, tokens, mode, taps, inp=None):
        hp = self.hp
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        N = tokens.size
        E, H = hp["n_embd"], hp["n_head"]
        logits = np.zeros((N, hp["n_vocab"]), dtype=np.float32)
        t = None
        tap_arrays = None
        if taps:
            T = self.n_past + N
            tap_arrays = {"inpL0": np.zeros((N, E), np.float32), "layer0_attn_norm": np.zeros((N, E), np.float32),
                          "layer0_q": np.zeros((N, H, E // H), np.float32),
                          "layer0_kq": np.zeros((H, N, T), np.float32), "layer0_out": np.zeros((N, E), np.float32),
                          "final_norm": np.zeros((N, E), np.float32),
                          "layer_out_all": np.zeros((hp["n_layer"], N, E), np.float32)}
            t = _Taps()
            for k, a in tap_arrays.items():
                setattr(t, k, a.ctypes.data)
        if inp is not None:
            inp = np.ascontiguousarray(inp, dtype=np.float32).reshape(N, E)
            if t is None:
                t = _Taps()
            t.inp_override = inp.ctypes.data
        lib().orc_llama_eval(C.byref(self._m), _p(tokens), N, self.n_past, _p(logits), mode,
                             C.byref(t) if t is not None else None)
        self.n_past += N
        return (logits, tap_arrays) if taps else logits


class Gpt2:
    """CPU restatement of the GPT-2 graph of crates/models/gpt2/src/lib.rs:156-329 (BASELINE configs[0], "plumbing"):
    numpy orchestration over the C primitives above (norm, quantized mul_mat, gelu, scale+mask+softmax), one Python
    loop over layers.  `w` holds ggml-layout arrays: quantized 2-D weights as raw block bytes (type `wtype`) with
    [out rows of `in` elements], 1-D f32 gains/biases, `model/wpe` f32 [n_ctx, n_embd].  K and V are cached as f16,
    token-major (GPT-2 does not transpose V in the cache, lib.rs:198-210)."""

    def __init__(self, hp, w, n_ctx=None):
        self.hp, self.w = hp, w
        self.C = n_ctx or hp["n_ctx"]
        E, L = hp["n_embd"], hp["n_layer"]
        self.memory_k = np.zeros((L, self.C, E), np.float16)
        self.memory_v = np.zeros((L, self.C, E), np.float16)
        self.n_past = 0

    def _mm(self, name, rows, cols, x, mode):
        return mul_mat(self.hp["wtype"], self.w[name], rows, cols, x, mode)

    def _ln(self, x, g, b):
        return norm(x) * self.w[g] + self.w[b]

    def evaluate(self, tokens, mode=0):
        hp, w = self.hp, self.w
        E, H, L, V, t = hp["n_embd"], hp["n_head"], hp["n_layer"], hp["n_vocab"], hp["wtype"]
        D, N, P = E // H, len(tokens), self.n_past
        T = P + N
        rb = row_bytes(t, E)
        wte = w["model/wte"]
        x = np.stack([dequantize(t, wte[int(tok) * rb:(int(tok) + 1) * rb], E) for tok in tokens])
        x = x + w["model/wpe"][P:T]  # :166-169
        f16r = (lambda a: a.astype(np.float16).astype(np.float32)) if mode != MODE_MATH else (lambda a: a)
        for il in range(L):
            pre = f"model/h{il}/"
            cur = self._ln(x, pre + "ln_1/g", pre + "ln_1/b")  # :178-183
            qkv = self._mm(pre + "attn/c_attn/w", 3 * E, E, cur, mode) + w[pre + "attn/c_attn/b"]  # :186-187
            q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
            self.memory_k[il, P:T] = k.astype(np.float16)  # :197-210
            self.memory_v[il, P:T] = v.astype(np.float16)
            Kf = self.memory_k[il, :T].astype(np.float32).reshape(T, H, D)
            Vf = self.memory_v[il, :T].astype(np.float32).reshape(T, H, D)
            Qf = f16r(q).reshape(N, H, D)  # src1 of an F16 mul_mat is rounded to f16
            kq = np.einsum("thd,nhd->hnt", Kf.astype(np.float64), Qf.astype(np.float64)).astype(np.float32)  # :234
            pr = scale_mask_softmax(kq, np.float32(1.0) / np.sqrt(np.float32(D)), P, mode)  # :235-241
            kqv = np.einsum("thd,hnt->nhd", Vf.astype(np.float64), f16r(pr).astype(np.float64)).astype(np.float32)
            cur = kqv.reshape(N, E)  # :266-272
            cur = self._mm(pre + "attn/c_proj/w", E, E, cur, mode) + w[pre + "attn/c_proj/b"]  # :275-276
            ff_in = cur + x  # :279
            cur = self._ln(ff_in, pre + "ln_2/g", pre + "ln_2/b")  # :287-291
            cur = self._mm(pre + "mlp/c_fc/w", 4 * E, E, cur, mode) + w[pre + "mlp/c_fc/b"]  # :294-295
            cur = gelu(cur, mode)  # :298
            cur = self._mm(pre + "mlp/c_proj/w", E, 4 * E, cur, mode) + w[pre + "mlp/c_proj/b"]  # :301-302
            x = cur + ff_in  # :305
        x = self._ln(x, "model/ln_f/g", "model/ln_f/b")  # :311-312
        self.n_past = T
        head = w.get("model/lm_head", wte)  # :319
        return self._mm_raw(head, V, E, x, mode)

    def _mm_raw(self, raw, rows, cols, x, mode):
        return mul_mat(self.hp["wtype"], raw, rows, cols, x, mode)
