"""CPU tests of the drop-in boundary (no GPU, no compute calls): libggml_hip.so loads, exports every symbol
include/ggml_hip.h declares, keeps the struct layouts of the reference's bindgen tests
(crates/ggml/sys/src/lib.rs:174-238, 261-446, 458-533, 546-651), and builds graphs with the reference's
semantics (result shapes, strides, views, leaf/node classification, scratch behaviour)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"#define GGML_API.*", "", src)
    return set(re.findall(r"GGML_API[^;(]*?\b(\w+)\s*\(", src))


def test_every_declared_symbol_is_exported_and_bound(G):
    names = _declared("include/ggml_hip.h")
    assert len(names) > 130
    lib = C.CDLL(G.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    # the Python binding covers the same set (so tests exercise what the header promises)
    unbound = sorted(names - set(G.PROTOTYPES))
    assert not unbound, unbound
    host = _declared("llm_amd/csrc/host/llm_host.h")
    assert not [n for n in sorted(host) if not hasattr(lib, n)]


def test_struct_layouts_match_bindgen_numbers(G):
    assert C.sizeof(G.ggml_tensor) == 272
    assert G.ggml_tensor.ne.offset == 16 and G.ggml_tensor.nb.offset == 48 and G.ggml_tensor.op.offset == 80
    assert G.ggml_tensor.op_params.offset == 84 and G.ggml_tensor.src.offset == 128
    assert G.ggml_tensor.data.offset == 200 and G.ggml_tensor.name.offset == 208 and G.ggml_tensor.extra.offset == 256
    assert C.sizeof(G.ggml_cplan) == 16424 and G.ggml_cplan.n_tasks.offset == 20
    assert C.sizeof(G.ggml_cgraph) == 164520 and G.ggml_cgraph.leafs.offset == 65544
    assert G.ggml_cgraph.visited_hash_table.offset == 98312
    assert C.sizeof(G.ggml_init_params) == 24 and C.sizeof(G.ggml_scratch) == 24
    assert C.sizeof(G.ggml_compute_params) == 32 and C.sizeof(G.ggml_type_traits_t) == 40
    L = G.lib()
    assert L.ggml_graph_overhead() == 32 + 164528  # OBJECT_SIZE + pad16(GRAPH_SIZE)
    assert L.ggml_tensor_overhead() == 32 + 272 + 16


def test_type_table(G):
    L = G.lib()
    for t, (bs, be) in {0: (4, 1), 1: (2, 1), 2: (18, 32), 3: (20, 32), 6: (22, 32), 7: (24, 32), 8: (34, 32),
                        9: (40, 32), 16: (1, 1), 18: (4, 1)}.items():
        assert L.ggml_type_size(t) == bs and L.ggml_blck_size(t) == be
    assert abs(L.ggml_type_sizef(2) - 18 / 32) < 1e-7
    assert L.ggml_is_quantized(2) and not L.ggml_is_quantized(1)
    assert L.ggml_op_name(21) == b"MUL_MAT" and L.ggml_op_name(51) == b"UNARY" and L.ggml_type_name(6) == b"q5_0"
    assert L.ggml_internal_get_type_traits(2).vec_dot_type == 8
    assert L.ggml_internal_get_type_traits(7).vec_dot_type == 9


@pytest.mark.parametrize("t", [2, 3, 6, 7, 8])
def test_product_quantizer_matches_oracle_bytes_and_hist(G, O, t):
    rng = np.random.default_rng(t)
    x = (0.02 * rng.standard_normal((32, 256))).astype(np.float32)
    x[3, 64:96] = 0
    assert np.array_equal(G.quantize(t, x), O.quantize(t, x))
    out = np.zeros(G.row_bytes(t, x.size), np.uint8)
    h1, h2 = np.zeros(16, np.int64), np.zeros(16, np.int64)
    fn = getattr(G.lib(), "ggml_quantize_" + G.TYPE_NAMES[t])
    assert fn(x.ctypes.data, out.ctypes.data, x.size, 256, h1.ctypes.data) == out.size
    O.lib().orc_quantize(t, x.ctypes.data, out.ctypes.data, x.size, 256, h2.ctypes.data)
    assert np.array_equal(h1, h2) and h1.sum() == x.size


def test_all_zero_block_keeps_the_sign_ggml_gives_its_scale(G, O):
    """d = max / -8 (Q4_0) and max / -16 (Q5_0) of an all-zero block is -0.0, stored as f16 0x8000; Q8_0 (amax / 127)
    and the min/max types store +0.  The host quantizer, the oracle and (tests/test_device_tools_gpu.py) the device
    quantizer agree byte for byte — hipcc folds `x * c -> f16` into v_fma_mixlo_f16 x, c, +0, which would lose that sign."""
    z = np.zeros((1, 32), np.float32)
    for t, d_bytes in ((2, (0x00, 0x80)), (6, (0x00, 0x80)), (8, (0x00, 0x00)), (3, (0x00, 0x00)), (7, (0x00, 0x00))):
        b = G.quantize(t, z)
        assert tuple(int(v) for v in b[:2]) == d_bytes, (t, b[:2])
        assert np.array_equal(b, O.quantize(t, z))


@pytest.mark.parametrize("t", [2, 7])
def test_product_quantizer_large_tensor_threaded_equals_serial(G, O, t):
    """Above 2^15 blocks ggml_quantize_q* cuts the tensor over threads: bytes and histogram must not
    depend on that."""
    x = (0.02 * np.random.default_rng(t).standard_normal((1100, 1024))).astype(np.float32)
    out1, out2 = np.zeros(G.row_bytes(t, x.size), np.uint8), np.zeros(G.row_bytes(t, x.size), np.uint8)
    h1, h2 = np.zeros(16, np.int64), np.zeros(16, np.int64)
    fn = getattr(G.lib(), "ggml_quantize_" + G.TYPE_NAMES[t])
    assert fn(x.ctypes.data, out1.ctypes.data, x.size, 1024, h1.ctypes.data) == out1.size
    O.lib().orc_quantize(t, x.ctypes.data, out2.ctypes.data, x.size, 1024, h2.ctypes.data)
    assert np.array_equal(out1, out2) and np.array_equal(h1, h2) and h1.sum() == x.size


def test_fp16_product_matches_numpy(G):
    x = np.random.default_rng(0).standard_normal(4096).astype(np.float32) * 100
    out = np.zeros(x.size, np.uint16)
    G.lib().ggml_fp32_to_fp16_row(x.ctypes.data, out.ctypes.data, x.size)
    assert np.array_equal(out, x.astype(np.float16).view(np.uint16))
    back = np.zeros(x.size, np.float32)
    G.lib().ggml_fp16_to_fp32_row(out.ctypes.data, back.ctypes.data, x.size)
    assert np.array_equal(back, x.astype(np.float16).astype(np.float32))


def test_builders_shapes_strides_and_views(G):
    with G.Context(1 << 22) as c:
        a = c.new_tensor(G.TYPE_Q4_0, 256, 64)     # [K=256, M=64]
        assert a.nb[:2] == (18, 18 * 8) and a.nbytes() == 64 * 8 * 18
        b = c.new_tensor(G.TYPE_F32, 256, 5)
        y = c.op_mul_mat(a, b)
        assert y.ne == (64, 5, 1, 1) and y.t.type == G.TYPE_F32 and y.t.op == 21
        assert C.addressof(y.t.src[0].contents) == C.addressof(a.t) and C.addressof(y.t.src[1].contents) == C.addressof(b.t)
        r = c.op_reshape_3d(y, 16, 4, 5)
        assert r.ne == (16, 4, 5, 1) and r.t.data == y.t.data
        p = c.op_permute(r, 0, 2, 1, 3)
        assert p.ne == (16, 5, 4, 1) and p.nb == (4, 16 * 4 * 4, 16 * 4, 16 * 4 * 5 * 4)
        t = c.op_transpose(b)
        assert t.ne[:2] == (5, 256) and t.nb[:2] == (1024, 4) and G.lib().ggml_is_transposed(t.ptr)
        kv = c.new_tensor(G.TYPE_F16, 4096)
        v1 = c.op_view_1d(kv, 128, 512)
        assert v1.t.data == kv.t.data + 512 and v1.ne[0] == 128
        off = C.c_size_t.from_buffer_copy(bytes(v1.t.op_params)[:8]).value
        assert off == 512  # offset travels in op_params (the accelerator reads it there)
        v2 = c.op_view_2d(kv, 3, 32, 64 * 2, 10)
        assert v2.ne[:2] == (3, 32) and v2.nb[:3] == (2, 128, 128 * 32)
        v3 = c.op_view_3d(kv, 8, 16, 4, 64, 1024, 0)
        assert v3.nb == (2, 64, 1024, 4096)
        cp = c.op_cpy(b, c.new_tensor(G.TYPE_F16, 256, 5))
        assert cp.t.op == 25 and cp.t.type == G.TYPE_F16
        rn = c.op_rms_norm(b, 5e-6)
        assert abs(np.frombuffer(bytes(rn.t.op_params)[:4], np.float32)[0] - 5e-6) < 1e-12
        ro = c.op_rope_custom_inplace(r, 7, 16, 0, 1, 26000.0, 0.5)
        pr = np.frombuffer(bytes(ro.t.op_params), np.int32)
        assert list(pr[:4]) == [7, 16, 0, 1] and ro.t.data == r.t.data
        assert np.frombuffer(bytes(ro.t.op_params)[16:24], np.float32).tolist() == [26000.0, 0.5]
        si = c.op_silu(b)
        assert si.t.op == 51 and si.t.op_params[0] == 9  # GGML_OP_UNARY / GGML_UNARY_OP_SILU
        dm = c.op_diag_mask_inf_inplace(b, 3)
        assert dm.t.data == b.t.data and dm.t.op_params[0] == 3
        gr = c.op_get_rows(a, c.new_tensor(G.TYPE_I32, 7))
        assert gr.ne[:2] == (256, 7)


def test_graph_order_leafs_and_scratch(G):
    with G.Context(1 << 22) as c:
        scratch = np.zeros(1 << 16, np.uint8)
        x = c.new_tensor(G.TYPE_F32, 32, 2)
        w = c.new_tensor(G.TYPE_F32, 32)
        used0 = G.lib().ggml_used_mem(c.ptr)
        c.use_scratch(scratch.ctypes.data, scratch.nbytes)
        n1 = c.op_rms_norm(x, 1e-5)
        assert scratch.ctypes.data <= n1.t.data < scratch.ctypes.data + scratch.nbytes  # data in the scratch
        s = c.new_f32(0.25)                                                           # constants are not
        assert not (scratch.ctypes.data <= s.t.data < scratch.ctypes.data + scratch.nbytes)
        assert np.frombuffer((C.c_char * 4).from_address(s.t.data), np.float32)[0] == 0.25
        n2 = c.op_mul(n1, w)
        n3 = c.op_scale_inplace(n2, s)
        c.use_scratch(None, 0)
        g = c.graph().build_forward_expand(n3)
        assert g.n_nodes == 3 and g.n_leafs == 3
        assert [g.node(i).t.op for i in range(3)] == [19, 6, 23]
        assert G.lib().ggml_used_mem(c.ptr) > used0
        plan = G.lib().ggml_graph_plan(g.ptr, 4)
        assert plan.n_threads == 4 and plan.work_size == 0
        # expanding the same result again adds nothing (visited hash table)
        g.build_forward_expand(n3)
        assert g.n_nodes == 3


def test_llama_layer_graph_has_the_reference_node_count(G):
    """One layer wired as crates/models/llama/src/lib.rs:174-338 yields 37 nodes."""
    E, H, F, C_, N, P = 64, 2, 96, 16, 3, 2
    D = E // H
    with G.Context(1 << 24) as c:
        mk, mv = c.new_tensor(G.TYPE_F16, C_ * E), c.new_tensor(G.TYPE_F16, C_ * E)
        W = {k: c.new_tensor(G.TYPE_Q4_0, *s) for k, s in dict(wq=(E, E), wk=(E, E), wv=(E, E), wo=(E, E), w1=(E, F),
                                                               w3=(E, F), w2=(F, E)).items()}
        an, fn = c.new_tensor(G.TYPE_F32, E), c.new_tensor(G.TYPE_F32, E)
        inp = c.new_tensor(G.TYPE_F32, E, N)
        g = c.graph()
        cur = c.op_mul(c.op_rms_norm(inp, 5e-6), an)
        q = c.op_rope_inplace(c.op_reshape_3d(c.op_mul_mat(W["wq"], cur), D, H, N), P, D, 0, 0)
        k = c.op_rope_inplace(c.op_reshape_3d(c.op_mul_mat(W["wk"], cur), D, H, N), P, D, 0, 0)
        v = c.op_transpose(c.op_reshape_2d(c.op_mul_mat(W["wv"], cur), E, N))
        g.build_forward_expand(c.op_cpy(k, c.op_view_1d(mk, N * E, 2 * E * P)))
        g.build_forward_expand(c.op_cpy(v, c.op_view_2d(mv, N, E, C_ * 2, P * 2)))
        Q = c.op_permute(q, 0, 2, 1, 3)
        K = c.op_permute(c.op_reshape_3d(c.op_view_1d(mk, (P + N) * E, 0), D, H, P + N), 0, 2, 1, 3)
        kq = c.op_soft_max_inplace(c.op_diag_mask_inf_inplace(c.op_scale_inplace(c.op_mul_mat(K, Q), c.new_f32(0.1)), P))
        assert kq.ne == (P + N, N, H, 1)
        V = c.op_view_3d(mv, P + N, D, H, C_ * 2, C_ * 2 * D, 0)
        kqv = c.op_mul_mat(V, kq)
        assert kqv.ne == (D, N, H, 1)
        cur = c.op_cpy(c.op_permute(kqv, 0, 2, 1, 3), c.new_tensor(G.TYPE_F32, E, N))
        ff_in = c.op_add(c.op_mul_mat(W["wo"], cur), inp)
        cur = c.op_mul(c.op_rms_norm(ff_in, 5e-6), fn)
        t3 = c.op_mul_mat(W["w3"], cur)
        cur = c.op_mul(c.op_silu(c.op_mul_mat(W["w1"], cur)), t3)
        out = c.op_add(c.op_mul_mat(W["w2"], cur), ff_in)
        g.build_forward_expand(out)
        assert g.n_nodes == 37
        ops = [g.node(i).t.op for i in range(g.n_nodes)]
        assert ops.count(21) == 9 and ops.count(25) == 3 and ops.count(38) == 2
        # cpy(k) precedes the K·Q matmul in execution order (the ordering the KV cache relies on)
        first_cpy = ops.index(25)
        kq_idx = [i for i in range(g.n_nodes) if g.node(i).t.op == 21 and g.node(i).ne[0] == P + N][0]
        assert first_cpy < kq_idx


def test_out_of_path_ops_abort_not_fallback(G):
    """alibi / flash_attn / map_* have no device implementation: the library aborts instead of computing on CPU."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from llm_amd import ggml as G\n"
            "c = G.Context(1 << 20); a = c.new_tensor(G.TYPE_F32, 8, 8); c.op_alibi(a, 0, 2, 8.0)") % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert p.returncode != 0 and "no CPU compute fallback" in p.stderr


def test_compute_without_gpu_fails_loudly(G):
    """The product path must not silently run anywhere else: without a HIP device graph compute aborts."""
    if G.has_gpu():
        pytest.skip("a GPU is present")
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from llm_amd import ggml as G\n"
            "import numpy as np\n"
            "c = G.Context(1 << 20); a = c.tensor_from(np.ones((2, 8), np.float32)); y = c.op_add(a, a)\n"
            "c.graph().build_forward_expand(y).compute()") % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert p.returncode != 0 and "no HIP device" in p.stderr


def test_ggjt_container_reader_roundtrip_and_rejections(tmp_path):
    """the C++ GGML/GGMF/GGJT reader (llm_ggml_file_open) against files written the way
    crates/ggml/src/format/saver.rs does — hyperparameters, vocabulary with scores, tensor directory in file order,
    32-byte alignment of GGJT tensor data, the tensor bytes themselves — and the reference's load errors
    (InvalidMagic, InvalidFormatVersion, dims[0] % 64 for Q4_0, truncated data).  No device involved."""
    import struct
    from llm_amd import ggml as G, llama, synth
    hp, w = synth.make_llama(dict(synth.TINY, n_ff=384), G.TYPE_Q4_0, seed=3)  # every ne0 a multiple of 64
    for container, version in (("ggjt", 3), ("ggjt", 1), ("ggmf", 1), ("ggml", 0)):
        p = tmp_path / f"m_{container}{version}.bin"
        synth.write_ggjt(p, hp, w, container=container, version=version)
        info = llama.inspect_file(p)
        assert info is not None, (container, version)
        assert info["container"] == {"ggml": 0, "ggmf": 1, "ggjt": 2}[container] and info["version"] == version
        h = info["hp"]
        assert (h.n_vocab, h.n_embd, h.n_head, h.n_head_kv, h.n_layer, h.n_rot) == (
            hp["n_vocab"], hp["n_embd"], hp["n_head"], hp["n_head"], hp["n_layer"], hp["n_rot"])
        assert h.file_type == 2000 + 2  # qnt version 2, MostlyQ4_0
        assert len(info["vocab"]) == hp["n_vocab"] and info["vocab"][7][0] == b"<7>"
        assert info["vocab"][7][1] == (0.0 if container == "ggml" else -7.0)
        shapes = synth.tensor_shapes(hp)
        assert [t["name"] for t in info["tensors"]] == list(shapes)
        for t in info["tensors"]:
            ne0, ne1 = shapes[t["name"]]
            assert t["ne"] == (ne0, 1 if ne1 is None else ne1) and t["n_dims"] == (1 if ne1 is None else 2)
            assert t["type"] == (G.TYPE_F32 if ne1 is None else G.TYPE_Q4_0)
            assert t["head"] == np.ascontiguousarray(w[t["name"]]).tobytes()[:16]
            if container == "ggjt":
                assert t["offset_mod32"] == 0
    good = (tmp_path / "m_ggjt3.bin").read_bytes()
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"GGUF" + good[4:])
    assert llama.inspect_file(bad) is None  # InvalidMagic
    bad.write_bytes(good[:4] + struct.pack("<I", 4) + good[8:])
    assert llama.inspect_file(bad) is None  # InvalidFormatVersion (ggjt v4)
    bad.write_bytes(good[:-100])
    assert llama.inspect_file(bad) is None  # tensor data past the end of the file
    hp2 = dict(hp, n_embd=96, n_rot=24)  # 96 % 64 != 0: the Q4_0 sanity check of loader.rs:248-255
    w2 = {k: (np.zeros(synth.tensor_shapes(hp2)[k][0], np.float32) if v.dtype == np.float32 else
              np.zeros(G.row_bytes(G.TYPE_Q4_0, synth.tensor_shapes(hp2)[k][0]) * synth.tensor_shapes(hp2)[k][1], np.uint8))
          for k, v in w.items()}
    synth.write_ggjt(bad, hp2, w2)
    assert llama.inspect_file(bad) is None


def test_comm_entry_points_exist_and_report_no_communicator(G):
    """The layer-split hop lives behind the C ABI (ggml_hip_comm_*, RCCL opened lazily): without a communicator the
    library reports 0 ranks and neither loads librccl nor needs a GPU for that answer."""
    L = G.lib()
    for name in ("ggml_hip_comm_unique_id", "ggml_hip_comm_init", "ggml_hip_comm_destroy", "ggml_hip_comm_ranks",
                 "ggml_hip_comm_send", "ggml_hip_comm_recv", "ggml_hip_comm_sendrecv"):
        assert hasattr(L, name), name
    assert L.ggml_hip_comm_ranks() == 0
    L.ggml_hip_comm_destroy()  # no-op without a communicator
    assert G.COMM_ID_BYTES == 128
    maps = open("/proc/self/maps").read()
    assert "librccl" not in maps


def test_layer_split_arithmetic_of_the_tensor_split_hook(G):
    """llm_split_layers: how ggml_cuda_set_tensor_split's fractions (crates/ggml/sys/src/cuda.rs:11; device i takes
    split[i] / sum) become contiguous LAYER ranges for an in-process split (contiguous L/G layers per GPU).  Pure
    host arithmetic, no device: equal shares for NULL / all zero, proportional otherwise, at least one layer per stage."""
    lib = C.CDLL(G.LIB_PATH)
    lib.llm_split_layers.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]

    def bounds(n_layer, fr):
        g = len(fr) if fr is not None else 4
        out = (C.c_int * (g + 1))()
        arr = (C.c_float * g)(*fr) if fr is not None else None
        lib.llm_split_layers(n_layer, g, arr, out)
        return list(out)

    assert bounds(32, None) == [0, 8, 16, 24, 32]                    # north_star: contiguous L/G
    assert bounds(32, [0, 0, 0, 0, 0, 0, 0, 0]) == [0, 4, 8, 12, 16, 20, 24, 28, 32]
    assert bounds(80, [1, 1, 1, 1, 1, 1, 1, 1]) == list(range(0, 81, 10))
    assert bounds(32, [0.5, 0.25, 0.25]) == [0, 16, 24, 32]
    assert bounds(5, [0.2, 0.8]) == [0, 1, 5]
    assert bounds(40, [3, 1]) == [0, 30, 40]
    assert bounds(2, [1, 1, 1, 1]) == [0, 1, 2, 2, 2]  # more slots than layers: the surplus ones stay empty
    # never an empty stage, always the whole model, monotone
    for n_layer, fr in ((3, [100, 1, 1]), (7, [1e-6, 1, 1e-6, 1]), (2, [1, 1]), (33, [1, 2, 3, 4, 5, 6, 7, 8])):
        b = bounds(n_layer, fr)
        assert b[0] == 0 and b[-1] == n_layer and all(y > x for x, y in zip(b, b[1:])), (n_layer, fr, b)


def test_tensor_split_hook_reads_exactly_one_float(G):
    """The reference hands ggml_cuda_set_tensor_split the address of ONE stack f32 (crates/ggml/src/accelerator/mod.rs:74-75).
    Whatever follows it in memory must not be read, let alone become a layer split — also with 8 device slots addressable
    (the hook used to read one float per slot).  A split over several slots has its own explicit-length entry point."""
    import os
    L = G.lib()
    os.environ["GGML_HIP_VIRTUAL_DEVICES"] = "8"
    try:
        for name in ("ggml_hip_set_tensor_split", "ggml_cuda_set_tensor_split"):
            stack = (C.c_float * 16)(*([1.0] + [777.0] * 15))  # the one float, then "stack garbage"
            getattr(L, name)(C.cast(stack, C.c_void_p))
            out = (C.c_float * 16)(*([-1.0] * 16))
            assert L.ggml_hip_get_tensor_split(out, 16) == 1
            assert out[0] == 1.0 and all(out[i] == 0.0 for i in range(1, 16)), list(out)
            assert L.ggml_hip_get_layer_split(out, 16) == 0  # no split came out of the hook
        L.ggml_hip_set_tensor_split(None)
        assert L.ggml_hip_get_tensor_split(out, 16) == 1 and out[0] == 0.0
        # the explicit-length sibling: exactly n fractions, capped at 16, cleared by (NULL, 0)
        fr = (C.c_float * 3)(0.2, 0.8, 0.0)
        L.ggml_hip_set_layer_split(fr, 3)
        assert L.ggml_hip_get_layer_split(out, 16) == 3 and [round(out[i], 3) for i in range(3)] == [0.2, 0.8, 0.0]
        L.ggml_hip_set_tensor_split(C.cast(stack, C.c_void_p))   # the reference's call leaves a configured split alone
        assert L.ggml_hip_get_layer_split(out, 16) == 3
        L.ggml_hip_set_layer_split(None, 0)
        assert L.ggml_hip_get_layer_split(out, 16) == 0
    finally:
        one = C.c_float(1.0)
        L.ggml_hip_set_tensor_split(C.byref(one))
        L.ggml_hip_set_layer_split(None, 0)
        os.environ.pop("GGML_HIP_VIRTUAL_DEVICES", None)


def test_host_greedy_argmax_follows_the_scalar_rule(G):
    """llm_infer_next_token_greedy's sampler (the reference's sampler chain reduced to argmax): the AVX2 version must return
    the index of `best = 0; if (l[i] > l[best]) best = i` — first maximum, NaNs never win, a NaN in front keeps 0, -0.0 == 0.0."""
    lib = C.CDLL(G.LIB_PATH)
    lib.llm_argmax_first.argtypes = [C.c_void_p, C.c_int, C.c_int]
    rng = np.random.default_rng(3)

    def ref(l):
        best = 0
        for i in range(1, len(l)):
            if l[i] > l[best]:
                best = i
        return best

    cases = []
    for n in (1, 2, 7, 8, 15, 16, 17, 31, 100, 257, 32000):
        a = rng.standard_normal(n).astype(np.float32)
        cases.append(a)
        b = np.round(a, 1).astype(np.float32)  # many ties
        cases.append(b)
        if n >= 8:
            c = a.copy(); c[rng.integers(0, n, 3)] = np.nan; cases.append(c)
            d = a.copy(); d[0] = np.nan; cases.append(d)
            e = np.zeros(n, np.float32); e[n // 2] = -0.0; e[n - 1] = 0.0; cases.append(e)
            f = np.full(n, -np.inf, np.float32); cases.append(f)
            g_ = a.copy(); g_[n - 1] = a.max() ; cases.append(g_)  # the maximum again at the end: the first one wins
    for a in cases:
        a = np.ascontiguousarray(a)
        want = ref(a) if a.size <= 300 else int(np.flatnonzero(a == np.nanmax(a))[0]) if not np.isnan(a[0]) else 0
        for which in (0, 1):
            assert lib.llm_argmax_first(a.ctypes.data, a.size, which) == want, (a.size, which)


def test_container_reader_takes_a_mixed_k_quant_file(tmp_path):
    """A *_K_M-style file: attention.wv, feed_forward.w2 and output as Q6_K, the other matrices Q4_K, norms f32 — every tensor
    record carries its own type (crates/ggml/src/format/loader.rs:160-281; file types crates/llm-base/src/loader.rs:80-93).
    The reader must report each tensor's type, dims and (32-byte aligned) data as written.  Blocks are random bytes: the reader
    does not interpret them.  No device involved."""
    from llm_amd import ggml as G, llama, synth
    hp = dict(n_vocab=64, n_embd=256, n_head=4, n_head_kv=4, n_layer=2, n_rot=64, n_ff=512, n_mult=32, wtype=G.TYPE_Q4_K)
    wt = {"output.weight": G.TYPE_Q6_K}
    for i in range(hp["n_layer"]):
        wt[f"layers.{i}.attention.wv.weight"] = G.TYPE_Q6_K
        wt[f"layers.{i}.feed_forward.w2.weight"] = G.TYPE_Q6_K
    hp["wtypes"] = wt
    rng = np.random.default_rng(12)
    w = {}
    for name, (ne0, ne1) in synth.tensor_shapes(hp).items():
        if ne1 is None:
            w[name] = rng.standard_normal(ne0).astype(np.float32)
        else:
            w[name] = rng.integers(0, 256, G.row_bytes(wt.get(name, G.TYPE_Q4_K), ne0) * ne1, dtype=np.uint8)
    p = tmp_path / "mixed_k.bin"
    synth.write_ggjt(p, hp, w)
    info = llama.inspect_file(p)
    assert info is not None and info["container"] == 2 and info["version"] == 3
    assert info["hp"].file_type == 2000 + G.FTYPE_OF[G.TYPE_Q4_K]
    shapes = synth.tensor_shapes(hp)
    assert [t["name"] for t in info["tensors"]] == list(shapes)
    n6 = 0
    for t in info["tensors"]:
        ne0, ne1 = shapes[t["name"]]
        want = G.TYPE_F32 if ne1 is None else wt.get(t["name"], G.TYPE_Q4_K)
        n6 += want == G.TYPE_Q6_K
        assert t["type"] == want and t["offset_mod32"] == 0 and t["ne"] == (ne0, 1 if ne1 is None else ne1)
        assert t["head"] == np.ascontiguousarray(w[t["name"]]).tobytes()[:16]
    assert n6 == 1 + 2 * hp["n_layer"]
