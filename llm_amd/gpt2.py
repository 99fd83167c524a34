"""GPT-2 through the drop-in C ABI: the graph of crates/models/gpt2/src/lib.rs:156-329 built node by node with the
ctypes binding (llm_amd.ggml) and executed by ggml_graph_compute on the MI355X.  BASELINE configs[0] is the
reference's CPU plumbing case; here it is the second model family that exercises the generic executor (LayerNorm,
biases via broadcast add, GELU, strided K/V stores, permuted f16 copies) — there is no fused plan for it.
Synthetic weights follow the reference's quantization rule (lib.rs:357-366: wte, lm_head and the four 2-D weights
per layer are quantized; gains, biases and wpe stay f32)."""
import numpy as np

from . import ggml as G

GPT2_117M = dict(n_vocab=50257, n_ctx=1024, n_embd=768, n_head=12, n_layer=12)
GPT2_TINY = dict(n_vocab=256, n_ctx=64, n_embd=128, n_head=4, n_layer=2)


def make_gpt2(hp0, wtype, seed=1234, quantize=None):
    """ggml-layout weights: dict name -> raw block bytes (quantized 2-D) or f32 array."""
    quantize = quantize or G.quantize
    hp = dict(hp0, wtype=wtype)
    E, L, V, C = hp["n_embd"], hp["n_layer"], hp["n_vocab"], hp["n_ctx"]
    rng = np.random.default_rng(seed)
    w = {}

    def q2d(name, rows, cols, std=0.02):
        w[name] = quantize(wtype, (std * rng.standard_normal((rows, cols))).astype(np.float32))

    def f1d(name, n, mean, std):
        w[name] = (mean + std * rng.standard_normal(n)).astype(np.float32)

    q2d("model/wte", V, E)
    w["model/wpe"] = (0.01 * rng.standard_normal((C, E))).astype(np.float32)
    f1d("model/ln_f/g", E, 1.0, 0.01)
    f1d("model/ln_f/b", E, 0.0, 0.01)
    for il in range(L):
        p = f"model/h{il}/"
        f1d(p + "ln_1/g", E, 1.0, 0.01); f1d(p + "ln_1/b", E, 0.0, 0.01)
        f1d(p + "ln_2/g", E, 1.0, 0.01); f1d(p + "ln_2/b", E, 0.0, 0.01)
        q2d(p + "attn/c_attn/w", 3 * E, E); f1d(p + "attn/c_attn/b", 3 * E, 0.0, 0.01)
        q2d(p + "attn/c_proj/w", E, E); f1d(p + "attn/c_proj/b", E, 0.0, 0.01)
        q2d(p + "mlp/c_fc/w", 4 * E, E); f1d(p + "mlp/c_fc/b", 4 * E, 0.0, 0.01)
        q2d(p + "mlp/c_proj/w", E, 4 * E); f1d(p + "mlp/c_proj/b", E, 0.0, 0.01)
    return hp, w


class Gpt2:
    """Model (weights resident on the device) + one session (f16 K/V memory)."""

    def __init__(self, hp, w, n_ctx=None):
        self.hp = hp
        self.C = n_ctx or hp["n_ctx"]
        E, L, V = hp["n_embd"], hp["n_layer"], hp["n_vocab"]
        nbytes = sum(a.nbytes for a in w.values()) + 512 * (len(w) + 4) + (1 << 16)
        self.ctx = G.Context(nbytes)
        self.t = {}
        for name, a in w.items():  # Gpt2::new, lib.rs:48-128: every tensor is transfer_to(backend)
            if a.dtype == np.float32:
                ne = tuple(reversed(a.shape))
                self.t[name] = self.ctx.tensor_from(a, G.TYPE_F32, ne).set_name(name[-40:]).transfer_to_gpu()
            else:
                tail = name.rsplit("/", 2)[-2] + "/" + name.rsplit("/", 2)[-1] if name.count("/") >= 2 else name
                rows = {"model/wte": V, "model/lm_head": V}.get(name) or {
                    "c_attn/w": 3 * E, "c_fc/w": 4 * E, "c_proj/w": E}[tail]
                cols = 4 * E if name.endswith("mlp/c_proj/w") else E
                self.t[name] = self.ctx.tensor_from(a, hp["wtype"], (cols, rows)).set_name(name[-40:]).transfer_to_gpu()
        self.sctx = G.Context(2 * L * self.C * E * 2 + (1 << 16))
        self.memory_k = self.sctx.new_tensor(G.TYPE_F16, L * self.C * E).set_name("memory_k").offload_no_scratch()
        self.memory_v = self.sctx.new_tensor(G.TYPE_F16, L * self.C * E).set_name("memory_v").offload_no_scratch()
        self.n_past = 0

    def free(self):
        self.sctx.free()
        self.ctx.free()

    def evaluate(self, tokens):
        """Gpt2::evaluate (lib.rs:138-335): returns logits [N, n_vocab]."""
        hp, t = self.hp, self.t
        E, H, L, V = hp["n_embd"], hp["n_head"], hp["n_layer"], hp["n_vocab"]
        D, N, P, C = E // H, len(tokens), self.n_past, self.C
        T = P + N
        ctx0 = G.Context(64 * 1024 * 1024 + L * N * (12 * E + 4 * T * H) * 16)
        try:
            off = lambda x: x.offload()  # ctx0.set_offloading(true): intermediate results stay on the device
            embd = ctx0.tensor_from(np.asarray(tokens, np.int32))
            position = ctx0.tensor_from(np.arange(P, T, dtype=np.int32))  # :164-167
            x = off(ctx0.op_add(off(ctx0.op_get_rows(t["model/wte"], embd)), off(ctx0.op_get_rows(t["model/wpe"], position))))
            gf = ctx0.graph()
            for il in range(L):
                p = f"model/h{il}/"
                cur = off(ctx0.op_norm(x))  # :178
                cur = off(ctx0.op_add(off(ctx0.op_mul(cur, t[p + "ln_1/g"])), t[p + "ln_1/b"]))
                cur = off(ctx0.op_mul_mat(t[p + "attn/c_attn/w"], cur))  # :186
                cur = off(ctx0.op_add(cur, t[p + "attn/c_attn/b"]))
                nb1 = cur.nb[1]
                qcur = ctx0.op_view_2d(cur, E, N, nb1, 0)  # :192-195
                kcur = ctx0.op_view_2d(cur, E, N, nb1, 4 * E)
                vcur = ctx0.op_view_2d(cur, E, N, nb1, 8 * E)
                k = ctx0.op_view_1d(self.memory_k, N * E, 2 * E * (il * C + P))  # :198-207
                v = ctx0.op_view_1d(self.memory_v, N * E, 2 * E * (il * C + P))
                gf.build_forward_expand(off(ctx0.op_cpy(kcur, k)))
                gf.build_forward_expand(off(ctx0.op_cpy(vcur, v)))
                q = ctx0.op_permute(off(ctx0.op_cpy(qcur, ctx0.new_tensor(G.TYPE_F32, D, H, N))), 0, 2, 1, 3)  # :213-219
                kk = ctx0.op_permute(ctx0.op_reshape_3d(ctx0.op_view_1d(self.memory_k, T * E, il * C * 2 * E), D, H, T),
                                     0, 2, 1, 3)  # :221-232
                kq = off(ctx0.op_mul_mat(kk, q))
                kq = off(ctx0.op_scale_inplace(kq, ctx0.new_f32(1.0 / np.sqrt(np.float32(E) / np.float32(H)))))
                kq = off(ctx0.op_diag_mask_inf_inplace(kq, P))
                kq = off(ctx0.op_soft_max_inplace(kq))
                vt = off(ctx0.op_cpy(  # :243-264
                    ctx0.op_permute(ctx0.op_reshape_3d(ctx0.op_view_1d(self.memory_v, T * E, il * C * 2 * E), D, H, T),
                                    1, 2, 0, 3),
                    ctx0.new_tensor(G.TYPE_F16, T, D, H)))
                kqv = off(ctx0.op_mul_mat(vt, kq))
                cur = off(ctx0.op_cpy(ctx0.op_permute(kqv, 0, 2, 1, 3), ctx0.new_tensor(G.TYPE_F32, E, N)))  # :266-272
                cur = off(ctx0.op_mul_mat(t[p + "attn/c_proj/w"], cur))
                cur = off(ctx0.op_add(cur, t[p + "attn/c_proj/b"]))
                ff_in = off(ctx0.op_add(cur, x))  # :279
                cur = off(ctx0.op_norm(ff_in))
                cur = off(ctx0.op_add(off(ctx0.op_mul(cur, t[p + "ln_2/g"])), t[p + "ln_2/b"]))
                cur = off(ctx0.op_mul_mat(t[p + "mlp/c_fc/w"], cur))
                cur = off(ctx0.op_add(cur, t[p + "mlp/c_fc/b"]))
                cur = off(ctx0.op_gelu(cur))  # :298
                cur = off(ctx0.op_mul_mat(t[p + "mlp/c_proj/w"], cur))
                cur = off(ctx0.op_add(cur, t[p + "mlp/c_proj/b"]))
                x = off(ctx0.op_add(cur, ff_in))  # :305
            x = off(ctx0.op_norm(x))
            x = off(ctx0.op_add(off(ctx0.op_mul(x, t["model/ln_f/g"])), t["model/ln_f/b"]))
            head = t.get("model/lm_head", t["model/wte"])  # :319
            logits = ctx0.op_mul_mat(head, x)  # set_offloading(false): the result is read on the host
            gf.build_forward_expand(logits)
            gf.compute()
            self.n_past = T
            return logits.read_data().reshape(N, V).copy()
        finally:
            ctx0.free()
