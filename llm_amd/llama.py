"""ctypes handle on the host mirror (llm_amd/csrc/host/llm_host.cpp): the call sequence of
crates/llm-base (Model::start_session, InferenceSession::feed_prompt / infer_next_token,
Model::evaluate) as a rustformers/llm user drives it."""
import ctypes as C

import numpy as np

from . import ggml


class _HP(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_vocab", "n_embd", "n_mult", "n_head", "n_head_kv", "n_layer", "n_rot", "file_type")]


class _MP(C.Structure):
    _fields_ = [("context_size", C.c_int32), ("use_gpu", C.c_int32), ("gpu_layers", C.c_int32),
                ("has_rope_overrides", C.c_int32), ("rope_frequency_scale", C.c_float),
                ("rope_frequency_base", C.c_int32), ("layer_begin", C.c_int32), ("layer_end", C.c_int32), ("n_gqa", C.c_int32)]


class _SC(C.Structure):
    _fields_ = [("memory_k_type", C.c_int32), ("memory_v_type", C.c_int32), ("n_batch", C.c_int32),
                ("n_threads", C.c_int32)]


class _TD(C.Structure):
    _fields_ = [("name", C.c_char_p), ("type", C.c_int32), ("n_dims", C.c_int32), ("ne", C.c_int64 * 2),
                ("data", C.c_void_p)]


_bound = False


def _lib():
    global _bound
    L = ggml.lib()
    if not _bound:
        L.llm_llama_new.restype = C.c_void_p
        L.llm_llama_new.argtypes = [C.POINTER(_HP), C.POINTER(_MP), C.POINTER(_TD), C.c_int]
        L.llm_model_free.argtypes = [C.c_void_p]
        L.llm_model_stages.restype = C.c_int
        L.llm_model_stages.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.llm_ggml_file_open.restype = C.c_void_p
        L.llm_ggml_file_open.argtypes = [C.c_char_p]
        L.llm_ggml_file_close.argtypes = [C.c_void_p]
        L.llm_ggml_file_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_HP),
                                         C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.llm_ggml_file_tensor.restype = C.c_int
        L.llm_ggml_file_tensor.argtypes = [C.c_void_p, C.c_int, C.POINTER(_TD)]
        L.llm_ggml_file_vocab.restype = C.c_int
        L.llm_ggml_file_vocab.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_float)]
        L.llm_llama_load.restype = C.c_void_p
        L.llm_llama_load.argtypes = [C.c_char_p, C.POINTER(_MP)]
        L.llm_start_session.restype = C.c_void_p
        L.llm_start_session.argtypes = [C.c_void_p, C.POINTER(_SC)]
        L.llm_session_free.argtypes = [C.c_void_p]
        L.llm_start_session_on.restype = C.c_void_p
        L.llm_start_session_on.argtypes = [C.c_void_p, C.POINTER(_SC), C.c_int]
        L.llm_session_read_node_host.restype = C.c_size_t
        L.llm_session_read_node_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.llm_session_seek.argtypes = [C.c_void_p, C.c_int]
        L.llm_session_set_speculate.argtypes = [C.c_void_p, C.c_int]
        L.llm_evaluate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.llm_feed_prompt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.llm_infer_next_token_greedy.restype = C.c_int32
        L.llm_infer_next_token_greedy.argtypes = [C.c_void_p, C.c_void_p]
        L.llm_infer_next_token_topk.restype = C.c_int32
        L.llm_infer_next_token_topk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.POINTER(C.c_uint64), C.c_int]
        L.llm_session_snapshot.restype = C.c_size_t
        L.llm_session_snapshot.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.llm_session_from_snapshot.restype = C.c_void_p
        L.llm_session_from_snapshot.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.llm_infer_tokens_greedy_device.restype = C.c_int
        L.llm_infer_tokens_greedy_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.llm_host_timing.restype = None
        L.llm_host_timing.argtypes = [C.POINTER(C.c_double), C.c_int]
        L.llm_session_rewind.restype = C.c_int
        L.llm_session_rewind.argtypes = [C.c_void_p, C.c_int]
        L.llm_session_last_logits.restype = C.POINTER(C.c_float)
        L.llm_session_last_logits.argtypes = [C.c_void_p]
        L.llm_session_n_past.restype = C.c_int
        L.llm_session_n_past.argtypes = [C.c_void_p]
        L.llm_session_last_graph_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.llm_session_stage_buffers.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                C.POINTER(C.c_size_t)]
        L.llm_session_kv.restype = C.c_size_t
        L.llm_session_kv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.llm_session_topk.restype = C.c_int
        L.llm_session_topk.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.llm_session_read_node.restype = C.c_size_t
        L.llm_session_read_node.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t]
        _bound = True
    return L


def inspect_file(path):
    """Parses a GGML-family container with the C++ reader (no device needed): None if it is rejected, else
    {container, version, hp, vocab: [(bytes, score)], tensors: [{name, type, n_dims, ne, offset_mod32, head}]}."""
    L = _lib()
    f = L.llm_ggml_file_open(str(path).encode())
    if not f:
        return None
    try:
        c, v, nt, nv, hp = C.c_int(), C.c_int(), C.c_int(), C.c_int(), _HP()
        L.llm_ggml_file_info(f, C.byref(c), C.byref(v), C.byref(hp), C.byref(nt), C.byref(nv))
        vocab = []
        for i in range(nv.value):
            buf, sc = C.create_string_buffer(256), C.c_float()
            n = L.llm_ggml_file_vocab(f, i, buf, 256, C.byref(sc))
            vocab.append((buf.raw[:min(n, 256)], sc.value))
        tensors = []
        for i in range(nt.value):
            d = _TD()
            L.llm_ggml_file_tensor(f, i, C.byref(d))
            head = bytes((C.c_uint8 * 16).from_address(d.data))
            tensors.append(dict(name=d.name.decode(), type=d.type, n_dims=d.n_dims, ne=(d.ne[0], d.ne[1]),
                                offset_mod32=d.data % 32, head=head))
        return dict(container=c.value, version=v.value, hp=hp, vocab=vocab, tensors=tensors)
    finally:
        L.llm_ggml_file_close(f)


class Llama:
    """models/llama Llama + ModelParameters{use_gpu: true}.  `weights`: {name: ndarray} as produced by
    llm_amd.synth (raw GGML bytes for 2-D tensors) — must outlive the model (mmap semantics)."""

    def __init__(self, hp, weights, context_size=2048, gpu_layers=-1, rope_overrides=None, layer_range=None):
        from .synth import stage_tensor_names, tensor_shapes
        self.hp = dict(hp)
        self.weights = weights
        L = _lib()
        lb, le = layer_range if layer_range else (0, hp["n_layer"])
        self.layer_range = (lb, le)
        self.is_first, self.is_last = lb == 0, le == hp["n_layer"]
        keep = stage_tensor_names(hp, lb, le)
        shapes = {k: v for k, v in tensor_shapes(hp).items() if k in keep}
        descs = (_TD * len(shapes))()
        self._names = []
        for i, (name, (ne0, ne1)) in enumerate(shapes.items()):
            b = name.encode()
            self._names.append(b)
            descs[i].name = b
            wt = hp.get("wtypes", {}).get(name, hp["wtype"])  # per-tensor types of a mixed file (*_K_M: some tensors Q6_K)
            descs[i].type = ggml.TYPE_F32 if ne1 is None else wt
            descs[i].n_dims = 1 if ne1 is None else 2
            descs[i].ne[0] = ne0
            descs[i].ne[1] = 1 if ne1 is None else ne1
            arr = weights[name]
            exp = ne0 * 4 if ne1 is None else ggml.row_bytes(wt, ne0) * ne1
            assert arr.nbytes == exp, (name, arr.nbytes, exp)
            descs[i].data = arr.ctypes.data
        h = _HP(hp["n_vocab"], hp["n_embd"], hp.get("n_mult", 256), hp["n_head"], hp["n_head_kv"], hp["n_layer"],
                hp["n_rot"], 2 * 1000 + ggml.FTYPE_OF[hp["wtype"]])
        mp = _MP(context_size, 1, gpu_layers, 0, 1.0, 10000, lb, le, 0)
        if rope_overrides:
            mp.has_rope_overrides = 1
            mp.rope_frequency_scale = rope_overrides["frequency_scale"]
            mp.rope_frequency_base = rope_overrides["frequency_base"]
        self.context_size = context_size
        self.ptr = L.llm_llama_new(C.byref(h), C.byref(mp), descs, len(shapes))

    @classmethod
    def load(cls, path, context_size=2048, gpu_layers=-1, n_gqa=0):
        """llm::load::<Llama>(path, …, ModelParameters{prefer_mmap: true, use_gpu: true}): the C++ container reader
        (llm_ggml_file_open) maps the GGML/GGMF/GGJT file and the tensors point into the mapping."""
        L = _lib()
        info = inspect_file(path)
        if info is None:
            raise ValueError(f"{path}: not a loadable GGML-family container")
        self = cls.__new__(cls)
        mp = _MP(context_size, 1, gpu_layers, 0, 1.0, 10000, 0, -1, n_gqa)  # n_gqa: ModelParameters::n_gqa (80 layers and more)
        self.ptr = L.llm_llama_load(str(path).encode(), C.byref(mp))
        if not self.ptr:
            raise ValueError(f"{path}: load failed")
        h = info["hp"]
        wtype = next(t["type"] for t in info["tensors"] if t["n_dims"] == 2)
        n_ff = next(t["ne"][1] for t in info["tensors"] if t["name"].endswith("feed_forward.w1.weight"))
        if n_gqa > 0 and h.n_layer >= 80:
            h.n_head_kv = h.n_head // n_gqa
        self.hp = dict(n_vocab=h.n_vocab, n_embd=h.n_embd, n_mult=h.n_mult, n_head=h.n_head, n_head_kv=h.n_head_kv,
                       n_layer=h.n_layer, n_rot=h.n_rot, n_ff=n_ff, wtype=wtype)
        self.weights = None
        self.layer_range = (0, h.n_layer)
        self.is_first = self.is_last = True
        self.context_size = context_size
        return self

    def stages(self):
        """[(layer_begin, layer_end, device_slot)] — one entry for an unsplit model, one per device slot for a model that
        llm_llama_new split over the GPUs of this process (ggml_hip_set_tensor_split / GGML_HIP_LAYER_SPLIT)."""
        lb, le, dv = ((C.c_int * 16)() for _ in range(3))
        n = _lib().llm_model_stages(self.ptr, lb, le, dv, 16)
        return [(lb[i], le[i], dv[i]) for i in range(n)]

    def start_session(self, n_batch=8, kv_type=ggml.TYPE_F16):
        return Session(self, n_batch, kv_type)

    def start_session_on(self, slot, n_batch=8, kv_type=ggml.TYPE_F16):
        """A session of this (unsplit) model on another device slot of the model's GPU: its own stream, K/V and plans, the
        model's weights — sessions on different slots run concurrently (llm_start_session_on)."""
        return Session(self, n_batch, kv_type, slot=slot)

    def session_from_snapshot(self, blob):
        """InferenceSession::from_snapshot; None on SnapshotError."""
        blob = np.ascontiguousarray(np.frombuffer(blob, dtype=np.uint8))
        ptr = _lib().llm_session_from_snapshot(self.ptr, blob.ctypes.data, blob.size)
        if not ptr:
            return None
        s = Session.__new__(Session)
        s.model, s.ptr = self, ptr
        return s

    def free(self):
        if self.ptr:
            _lib().llm_model_free(self.ptr)
            self.ptr = None


class Session:
    def __init__(self, model, n_batch, kv_type, slot=-1):
        self.model = model
        cfg = _SC(kv_type, kv_type, n_batch, 8)
        self.ptr = (_lib().llm_start_session(model.ptr, C.byref(cfg)) if slot < 0 else
                    _lib().llm_start_session_on(model.ptr, C.byref(cfg), slot))

    def seek(self, n_past):
        """The session continues at position n_past (the caller has put the K/V before it in place: set_kv)."""
        _lib().llm_session_seek(self.ptr, int(n_past))

    def set_speculate(self, on):
        """False = the reference's own call sequence per token (build the graph, then ggml_graph_compute); True = the next
        token's graph is built between ggml_hip_graph_compute_begin / _end while the device runs."""
        _lib().llm_session_set_speculate(self.ptr, 1 if on else 0)

    @property
    def n_past(self):
        return _lib().llm_session_n_past(self.ptr)

    def evaluate(self, tokens, want_all_logits=True, want_embeddings=False):
        """Model::evaluate: returns all logits [N, n_vocab] (OutputRequest.all_logits) if requested."""
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        V, E = self.model.hp["n_vocab"], self.model.hp["n_embd"]
        if not self.model.is_last:
            want_all_logits = want_embeddings = False
        logits = np.zeros((tokens.size, V), np.float32) if want_all_logits else None
        emb = np.zeros(E, np.float32) if want_embeddings else None
        _lib().llm_evaluate(self.model.ptr, self.ptr, tokens.ctypes.data, tokens.size,
                            logits.ctypes.data if logits is not None else None,
                            emb.ctypes.data if emb is not None else None)
        if want_embeddings:
            return logits, emb
        return logits

    def feed_prompt(self, tokens):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        _lib().llm_feed_prompt(self.model.ptr, self.ptr, tokens.ctypes.data, tokens.size)

    def infer_next_token(self):
        return int(_lib().llm_infer_next_token_greedy(self.model.ptr, self.ptr))

    def infer_next_token_topk(self, rng, k=40, temperature=0.8, device_topk=True):
        """One token step with the shape of the reference's default sampler (llm_infer_next_token_topk); rng: ctypes c_uint64 state."""
        return int(_lib().llm_infer_next_token_topk(self.model.ptr, self.ptr, k, temperature, C.byref(rng), 1 if device_topk else 0))

    def snapshot(self):
        """InferenceSession::get_snapshot as bytes (npast, config, tokens, last_logits, memory_k, memory_v)."""
        n = _lib().llm_session_snapshot(self.ptr, None, 0)
        buf = np.zeros(n, np.uint8)
        _lib().llm_session_snapshot(self.ptr, buf.ctypes.data, n)
        return buf.tobytes()

    def infer_tokens_device(self, n):
        """n greedy tokens sampled on the device (llm_infer_tokens_greedy_device); returns their ids."""
        out = np.zeros(n, np.int32)
        _lib().llm_infer_tokens_greedy_device(self.model.ptr, self.ptr, n, out.ctypes.data)
        return out

    @staticmethod
    def host_timing(reset=False):
        """Accumulated host ns per phase of the decode loop (llm_host_timing)."""
        a = (C.c_double * 8)()
        _lib().llm_host_timing(a, 1 if reset else 0)
        return list(a)

    def rewind(self, num):
        return _lib().llm_session_rewind(self.ptr, num)

    def last_logits(self):
        V = self.model.hp["n_vocab"]
        return np.ctypeslib.as_array(_lib().llm_session_last_logits(self.ptr), shape=(V,)).copy()

    def stage_buffers(self):
        """(in_dev_ptr, out_dev_ptr, nbytes) of the layer-split residual hand-off buffers (None if absent)."""
        a, b, n = C.c_void_p(0), C.c_void_p(0), C.c_size_t(0)
        _lib().llm_session_stage_buffers(self.ptr, C.byref(a), C.byref(b), C.byref(n))
        return a.value, b.value, n.value

    def get_kv(self, dtype=np.uint16):
        """(memory_k, memory_v) raw contents — the InferenceSnapshot payload."""
        out = []
        for which in (0, 1):
            n = _lib().llm_session_kv(self.ptr, which, 0, None, 0)
            a = np.zeros(n // np.dtype(dtype).itemsize, dtype=dtype)
            _lib().llm_session_kv(self.ptr, which, 0, a.ctypes.data, a.nbytes)
            out.append(a)
        return out

    def set_kv(self, k, v):
        for which, a in ((0, k), (1, v)):
            a = np.ascontiguousarray(a)
            _lib().llm_session_kv(self.ptr, which, 1, a.ctypes.data, a.nbytes)

    def top_k(self, k, extra_ids=()):
        """(values, ids) of the k largest logits of the last evaluated token, best first (lower id first on ties), then
        the raw logits of extra_ids — computed on the device (llm_session_topk); only k + len(extra_ids) pairs are
        read back."""
        extra = np.ascontiguousarray(extra_ids, dtype=np.int32)
        vals = np.zeros(k + extra.size, dtype=np.float32)
        ids = np.zeros(k + extra.size, dtype=np.int32)
        rc = _lib().llm_session_topk(self.ptr, k, extra.ctypes.data if extra.size else None, extra.size, vals.ctypes.data,
                                     ids.ctypes.data)
        if rc != 0:
            raise ValueError("llm_session_topk: no evaluated graph or bad arguments")
        return vals, ids

    def read_node(self, index=-1, name=None, occurrence=0, dtype=np.float32):
        """Test hook: device contents of a node of the last evaluated graph."""
        nm = name.encode() if name else None
        n = _lib().llm_session_read_node(self.ptr, index, nm, occurrence, None, 0)
        if n == 0:
            raise KeyError((index, name, occurrence))
        out = np.zeros(n // np.dtype(dtype).itemsize, dtype=dtype)
        _lib().llm_session_read_node(self.ptr, index, nm, occurrence, out.ctypes.data, out.nbytes)
        return out

    def read_node_host(self, from_end, dtype=np.float32):
        """Test hook: the HOST bytes of node n_nodes - 1 - from_end of the last evaluated graph (tensor->data)."""
        n = _lib().llm_session_read_node_host(self.ptr, from_end, None, 0)
        out = np.zeros(n // np.dtype(dtype).itemsize, dtype=dtype)
        _lib().llm_session_read_node_host(self.ptr, from_end, out.ctypes.data, out.nbytes)
        return out

    def graph_stats(self):
        a, b = C.c_int(0), C.c_int(0)
        _lib().llm_session_last_graph_stats(self.ptr, C.byref(a), C.byref(b))
        return a.value, b.value

    def free(self):
        if self.ptr:
            _lib().llm_session_free(self.ptr)
            self.ptr = None
