"""ctypes binding of libggml_hip.so — the Python twin of the reference's `ggml-sys` crate
(crates/ggml/sys/src/lib.rs + cuda.rs) plus thin `Context` / `Tensor` helpers shaped like
crates/ggml/src/{context,tensor}.rs.  Tests and bench.py call the product ONLY through this C ABI.

The library is built in-tree by `__graft_entry__.build()` (llm_amd/csrc/Makefile) and must exist:
there is no pure-Python or CPU fallback — importing works without a GPU, computing does not.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GGML_HIP_LIB") or os.path.join(_HERE, "libggml_hip.so")  # GGML_HIP_LIB: an experiment build

# enums (include/ggml_hip.h)
TYPE_F32, TYPE_F16, TYPE_Q4_0, TYPE_Q4_1, TYPE_Q5_0, TYPE_Q5_1, TYPE_Q8_0, TYPE_Q8_1 = 0, 1, 2, 3, 6, 7, 8, 9
TYPE_Q2_K, TYPE_Q3_K, TYPE_Q4_K, TYPE_Q5_K, TYPE_Q6_K, TYPE_Q8_K = 10, 11, 12, 13, 14, 15  # K-quants (kernels/kquant.h, kquant2.h); Q8_K is their activation type
TYPE_I8, TYPE_I16, TYPE_I32 = 16, 17, 18
BACKEND_CPU, BACKEND_GPU, BACKEND_GPU_SPLIT = 0, 10, 20
OP_NONE, OP_MUL_MAT = 0, 21
MAX_NODES, MAX_SRC, MAX_NAME, HASHTABLE = 4096, 6, 48, 8273
KCLASS_MMVQ, KCLASS_MMQ_MFMA, KCLASS_ATTN, KCLASS_OTHER = 0, 1, 2, 3
COMM_ID_BYTES = 128  # GGML_HIP_COMM_ID_BYTES
KKIND_BASE = 16  # + {0 wq|wk|wv, 1 wo, 2 w1|w3, 3 w2, 4 lm_head}: one kind of decode mat-vec (bench_plan_class)

TYPE_NAMES = {TYPE_F32: "f32", TYPE_F16: "f16", TYPE_Q4_0: "q4_0", TYPE_Q4_1: "q4_1", TYPE_Q5_0: "q5_0",
              TYPE_Q5_1: "q5_1", TYPE_Q8_0: "q8_0", TYPE_Q2_K: "q2_K", TYPE_Q3_K: "q3_K", TYPE_Q4_K: "q4_K", TYPE_Q5_K: "q5_K",
              TYPE_Q6_K: "q6_K", TYPE_I32: "i32"}
BLOCK_BYTES = {TYPE_F32: 4, TYPE_F16: 2, TYPE_Q4_0: 18, TYPE_Q4_1: 20, TYPE_Q5_0: 22, TYPE_Q5_1: 24, TYPE_Q8_0: 34,
               TYPE_Q8_1: 40, TYPE_Q2_K: 84, TYPE_Q3_K: 110, TYPE_Q4_K: 144, TYPE_Q5_K: 176, TYPE_Q6_K: 210, TYPE_Q8_K: 292,
               TYPE_I8: 1, TYPE_I16: 2, TYPE_I32: 4}
BLOCK_ELEMS = {TYPE_F32: 1, TYPE_F16: 1, TYPE_Q4_0: 32, TYPE_Q4_1: 32, TYPE_Q5_0: 32, TYPE_Q5_1: 32, TYPE_Q8_0: 32,
               TYPE_Q8_1: 32, TYPE_Q2_K: 256, TYPE_Q3_K: 256, TYPE_Q4_K: 256, TYPE_Q5_K: 256, TYPE_Q6_K: 256, TYPE_Q8_K: 256,
               TYPE_I8: 1, TYPE_I16: 1, TYPE_I32: 1}
QUANT_TYPES = (TYPE_Q4_0, TYPE_Q4_1, TYPE_Q5_0, TYPE_Q5_1, TYPE_Q8_0)
K_TYPES = (TYPE_Q2_K, TYPE_Q3_K, TYPE_Q4_K, TYPE_Q5_K, TYPE_Q6_K)  # files arrive pre-quantized: the library has no K-quant encoder (ggml_quantize_q4_K ...)
# llama.cpp ftype codes for GGJT files (crates/ggml/sys/src/llama.rs:16-32)
FTYPE_OF = {TYPE_F32: 0, TYPE_F16: 1, TYPE_Q4_0: 2, TYPE_Q4_1: 3, TYPE_Q8_0: 7, TYPE_Q5_0: 8, TYPE_Q5_1: 9,
            # synthetic files whose 2-D tensors are ALL of one K type carry the nearest llama.cpp code (real Q*_K_S / _M files
            # mix types per tensor; the loader reads each tensor's own type, the code is informational: loader.rs:32-36)
            TYPE_Q2_K: 10, TYPE_Q3_K: 11, TYPE_Q4_K: 14, TYPE_Q5_K: 16, TYPE_Q6_K: 18}


class ggml_tensor(C.Structure):
    pass


ggml_tensor._fields_ = [
    ("type", C.c_int), ("backend", C.c_int), ("n_dims", C.c_int),
    ("ne", C.c_int64 * 4), ("nb", C.c_size_t * 4),
    ("op", C.c_int), ("op_params", C.c_int32 * 8), ("is_param", C.c_bool),
    ("grad", C.POINTER(ggml_tensor)), ("src", C.POINTER(ggml_tensor) * MAX_SRC),
    ("perf_runs", C.c_int), ("perf_cycles", C.c_int64), ("perf_time_us", C.c_int64),
    ("data", C.c_void_p), ("name", C.c_char * MAX_NAME), ("extra", C.c_void_p), ("padding", C.c_char * 4)]


class ggml_init_params(C.Structure):
    _fields_ = [("mem_size", C.c_size_t), ("mem_buffer", C.c_void_p), ("no_alloc", C.c_bool)]


class ggml_scratch(C.Structure):
    _fields_ = [("offs", C.c_size_t), ("size", C.c_size_t), ("data", C.c_void_p)]


class ggml_cplan(C.Structure):
    _fields_ = [("work_size", C.c_size_t), ("work_data", C.c_void_p), ("n_threads", C.c_int),
                ("n_tasks", C.c_int * MAX_NODES), ("abort_callback", C.c_void_p), ("abort_callback_data", C.c_void_p)]


class ggml_cgraph(C.Structure):
    _fields_ = [("n_nodes", C.c_int), ("n_leafs", C.c_int),
                ("nodes", C.POINTER(ggml_tensor) * MAX_NODES), ("grads", C.POINTER(ggml_tensor) * MAX_NODES),
                ("leafs", C.POINTER(ggml_tensor) * MAX_NODES), ("visited_hash_table", C.c_void_p * HASHTABLE),
                ("perf_runs", C.c_int), ("perf_cycles", C.c_int64), ("perf_time_us", C.c_int64)]


class ggml_compute_params(C.Structure):
    _fields_ = [("type", C.c_int), ("ith", C.c_int), ("nth", C.c_int), ("wsize", C.c_size_t), ("wdata", C.c_void_p)]


class ggml_type_traits_t(C.Structure):
    _fields_ = [("to_float", C.c_void_p), ("from_float", C.c_void_p), ("from_float_reference", C.c_void_p),
                ("vec_dot", C.c_void_p), ("vec_dot_type", C.c_int)]


TP = C.POINTER(ggml_tensor)
CTX = C.c_void_p
GP = C.POINTER(ggml_cgraph)

# name -> (restype, argtypes): every symbol include/ggml_hip.h declares (tests/test_abi.py checks the list
# against the header, so a declaration without an export — or the reverse — fails on CPU)
PROTOTYPES = {
    "ggml_fp16_to_fp32": (C.c_float, [C.c_uint16]),
    "ggml_fp32_to_fp16": (C.c_uint16, [C.c_float]),
    "ggml_fp16_to_fp32_row": (None, [C.c_void_p, C.c_void_p, C.c_int]),
    "ggml_fp32_to_fp16_row": (None, [C.c_void_p, C.c_void_p, C.c_int]),
    "ggml_init": (CTX, [ggml_init_params]),
    "ggml_free": (None, [CTX]),
    "ggml_used_mem": (C.c_size_t, [CTX]),
    "ggml_set_scratch": (C.c_size_t, [CTX, ggml_scratch]),
    "ggml_get_no_alloc": (C.c_bool, [CTX]),
    "ggml_set_no_alloc": (None, [CTX, C.c_bool]),
    "ggml_get_mem_buffer": (C.c_void_p, [CTX]),
    "ggml_get_mem_size": (C.c_size_t, [CTX]),
    "ggml_get_max_tensor_size": (C.c_size_t, [CTX]),
    "ggml_print_objects": (None, [CTX]),
    "ggml_nelements": (C.c_int64, [TP]),
    "ggml_nrows": (C.c_int64, [TP]),
    "ggml_nbytes": (C.c_size_t, [TP]),
    "ggml_blck_size": (C.c_int, [C.c_int]),
    "ggml_type_size": (C.c_size_t, [C.c_int]),
    "ggml_type_sizef": (C.c_float, [C.c_int]),
    "ggml_type_name": (C.c_char_p, [C.c_int]),
    "ggml_op_name": (C.c_char_p, [C.c_int]),
    "ggml_element_size": (C.c_size_t, [TP]),
    "ggml_is_quantized": (C.c_bool, [C.c_int]),
    "ggml_is_transposed": (C.c_bool, [TP]),
    "ggml_is_contiguous": (C.c_bool, [TP]),
    "ggml_is_permuted": (C.c_bool, [TP]),
    "ggml_tensor_overhead": (C.c_size_t, []),
    "ggml_get_data": (C.c_void_p, [TP]),
    "ggml_get_data_f32": (C.c_void_p, [TP]),
    "ggml_get_name": (C.c_char_p, [TP]),
    "ggml_set_name": (TP, [TP, C.c_char_p]),
    "ggml_new_tensor": (TP, [CTX, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "ggml_new_tensor_1d": (TP, [CTX, C.c_int, C.c_int64]),
    "ggml_new_tensor_2d": (TP, [CTX, C.c_int, C.c_int64, C.c_int64]),
    "ggml_new_tensor_3d": (TP, [CTX, C.c_int, C.c_int64, C.c_int64, C.c_int64]),
    "ggml_new_tensor_4d": (TP, [CTX, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "ggml_new_i32": (TP, [CTX, C.c_int32]),
    "ggml_new_f32": (TP, [CTX, C.c_float]),
    "ggml_dup_tensor": (TP, [CTX, TP]),
    "ggml_view_tensor": (TP, [CTX, TP]),
    "ggml_dup": (TP, [CTX, TP]),
    "ggml_add": (TP, [CTX, TP, TP]),
    "ggml_add_inplace": (TP, [CTX, TP, TP]),
    "ggml_mul": (TP, [CTX, TP, TP]),
    "ggml_repeat": (TP, [CTX, TP, TP]),
    "ggml_silu": (TP, [CTX, TP]),
    "ggml_gelu": (TP, [CTX, TP]),
    "ggml_norm": (TP, [CTX, TP]),
    "ggml_rms_norm": (TP, [CTX, TP, C.c_float]),
    "ggml_mul_mat": (TP, [CTX, TP, TP]),
    "ggml_scale": (TP, [CTX, TP, TP]),
    "ggml_scale_inplace": (TP, [CTX, TP, TP]),
    "ggml_cpy": (TP, [CTX, TP, TP]),
    "ggml_cont": (TP, [CTX, TP]),
    "ggml_reshape": (TP, [CTX, TP, TP]),
    "ggml_reshape_1d": (TP, [CTX, TP, C.c_int64]),
    "ggml_reshape_2d": (TP, [CTX, TP, C.c_int64, C.c_int64]),
    "ggml_reshape_3d": (TP, [CTX, TP, C.c_int64, C.c_int64, C.c_int64]),
    "ggml_view_1d": (TP, [CTX, TP, C.c_int64, C.c_size_t]),
    "ggml_view_2d": (TP, [CTX, TP, C.c_int64, C.c_int64, C.c_size_t, C.c_size_t]),
    "ggml_view_3d": (TP, [CTX, TP, C.c_int64, C.c_int64, C.c_int64, C.c_size_t, C.c_size_t, C.c_size_t]),
    "ggml_permute": (TP, [CTX, TP, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ggml_transpose": (TP, [CTX, TP]),
    "ggml_get_rows": (TP, [CTX, TP, TP]),
    "ggml_diag_mask_inf": (TP, [CTX, TP, C.c_int]),
    "ggml_diag_mask_inf_inplace": (TP, [CTX, TP, C.c_int]),
    "ggml_soft_max": (TP, [CTX, TP]),
    "ggml_soft_max_inplace": (TP, [CTX, TP]),
    "ggml_rope": (TP, [CTX, TP, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ggml_rope_inplace": (TP, [CTX, TP, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ggml_rope_custom_inplace": (TP, [CTX, TP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float]),
    "ggml_alibi": (TP, [CTX, TP, C.c_int, C.c_int, C.c_float]),
    "ggml_flash_attn": (TP, [CTX, TP, TP, TP, C.c_bool]),
    "ggml_map_unary_f32": (TP, [CTX, TP, C.c_void_p]),
    "ggml_map_binary_f32": (TP, [CTX, TP, TP, C.c_void_p]),
    "ggml_new_graph": (GP, [CTX]),
    "ggml_graph_overhead": (C.c_size_t, []),
    "ggml_build_forward_expand": (None, [GP, TP]),
    "ggml_build_forward": (ggml_cgraph, [TP]),
    "ggml_graph_plan": (ggml_cplan, [GP, C.c_int]),
    "ggml_graph_compute": (C.c_int, [GP, C.POINTER(ggml_cplan)]),
    "ggml_graph_reset": (None, [GP]),
    "ggml_quantize_q4_0": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ggml_quantize_q4_1": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ggml_quantize_q5_0": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ggml_quantize_q5_1": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ggml_quantize_q8_0": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ggml_quantize_chunk": (C.c_size_t, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ggml_internal_get_type_traits": (ggml_type_traits_t, [C.c_int]),
    "ggml_cpu_has_blas": (C.c_int, []),
    "ggml_cpu_has_gpublas": (C.c_int, []),
}
_HOOKS = {
    "init_XXblas": (None, []),
    "PFX_set_tensor_split": (None, [C.c_void_p]),
    "PFX_mul": (None, [TP, TP, TP]),
    "PFX_can_mul_mat": (C.c_bool, [TP, TP, TP]),
    "PFX_mul_mat_get_wsize": (C.c_size_t, [TP, TP, TP]),
    "PFX_mul_mat": (None, [TP, TP, TP, C.c_void_p, C.c_size_t]),
    "PFX_host_malloc": (C.c_void_p, [C.c_size_t]),
    "PFX_host_free": (None, [C.c_void_p]),
    "PFX_transform_tensor": (None, [C.c_void_p, TP]),
    "PFX_free_data": (None, [TP]),
    "PFX_assign_buffers": (None, [TP]),
    "PFX_assign_buffers_no_scratch": (None, [TP]),
    "PFX_assign_buffers_force_inplace": (None, [TP]),
    "PFX_set_main_device": (None, [C.c_int]),
    "PFX_set_mul_mat_q": (None, [C.c_bool]),
    "PFX_set_scratch_size": (None, [C.c_size_t]),
    "PFX_free_scratch": (None, []),
    "PFX_compute_forward": (C.c_bool, [C.POINTER(ggml_compute_params), TP]),
}
for _k, _v in _HOOKS.items():
    if _k == "init_XXblas":
        PROTOTYPES["ggml_init_hipblas"] = _v
        PROTOTYPES["ggml_init_cublas"] = _v
    else:
        PROTOTYPES[_k.replace("PFX", "ggml_hip")] = _v
        PROTOTYPES[_k.replace("PFX", "ggml_cuda")] = _v
PROTOTYPES.update({
    "ggml_hip_device_count": (C.c_int, []),
    "ggml_hip_slot_physical_device": (C.c_int, [C.c_int]),
    "ggml_hip_thread_session_slot": (C.c_int, []),
    "ggml_hip_synchronize": (None, []),
    "ggml_hip_tensor_get": (None, [TP, C.c_void_p, C.c_size_t, C.c_size_t]),
    "ggml_hip_tensor_set": (None, [TP, C.c_void_p, C.c_size_t, C.c_size_t]),
    "ggml_hip_tensor_device_ptr": (C.c_void_p, [TP]),
    "ggml_hip_memcpy": (None, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "ggml_hip_timing_begin": (None, []),
    "ggml_hip_timing_end": (None, []),
    "ggml_hip_timing_query": (None, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "ggml_hip_set_option": (None, [C.c_char_p, C.c_int]),
    "ggml_hip_get_stat": (C.c_int64, [C.c_char_p]),
    "ggml_hip_read_timeline": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "ggml_hip_get_main_device": (C.c_int, []),
    "ggml_hip_bind_thread_device": (None, [C.c_int]),
    "ggml_hip_thread_pinned_device": (C.c_int, []),
    "ggml_hip_unbind_thread_device": (None, []),
    "ggml_hip_get_tensor_split": (C.c_int, [C.c_void_p, C.c_int]),
    "ggml_hip_set_layer_split": (None, [C.c_void_p, C.c_int]),
    "ggml_hip_get_layer_split": (C.c_int, [C.c_void_p, C.c_int]),
    "ggml_hip_copy_between_devices": (None, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "ggml_hip_share_stream": (C.c_int, [C.c_int, C.c_int]),
    "ggml_hip_debug_prompt_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.c_int, C.c_int64, C.c_float, C.c_int]),
    "ggml_hip_debug_mul_mat_cols": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "ggml_hip_debug_exp_le0": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ggml_hip_decode_greedy_chain": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ggml_hip_topk": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "ggml_hip_quantize": (C.c_size_t, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "ggml_hip_quantize_resident": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ggml_hip_graph_compute_begin": (C.c_int, [C.c_void_p]),
    "ggml_hip_graph_prepare": (C.c_int, [C.c_void_p]),
    "ggml_hip_graph_compute_end": (None, []),
    "ggml_hip_bench_plan_class": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                            C.POINTER(C.c_double)]),
    "ggml_hip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "ggml_hip_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "ggml_hip_comm_destroy": (None, []),
    "ggml_hip_comm_ranks": (C.c_int, []),
    "ggml_hip_comm_send": (None, [C.c_void_p, C.c_size_t, C.c_int]),
    "ggml_hip_comm_recv": (None, [C.c_void_p, C.c_size_t, C.c_int]),
    "ggml_hip_comm_sendrecv": (None, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t]),
    "ggml_hip_bench_empty": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "ggml_hip_version": (C.c_char_p, []),
})

_lib = None


def lib():
    """Loads libggml_hip.so (loudly fails if it was not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU / pure-Python fallback for the compute path.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def has_gpu():
    try:
        return lib().ggml_hip_device_count() > 0
    except Exception:
        return False


def row_bytes(t, k):
    return k // BLOCK_ELEMS[t] * BLOCK_BYTES[t]


def quantize(t, x):
    """ggml_quantize_q* of the product library (host function of the ABI): f32 [..., k] -> raw bytes."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if t == TYPE_F32:
        return x.view(np.uint8).reshape(-1).copy()
    if t == TYPE_F16:
        out = np.zeros(x.size, dtype=np.uint16)
        lib().ggml_fp32_to_fp16_row(x.ctypes.data, out.ctypes.data, x.size)
        return out.view(np.uint8)
    k = x.shape[-1]
    out = np.zeros(row_bytes(t, x.size), dtype=np.uint8)
    hist = np.zeros(16, dtype=np.int64)
    fn = getattr(lib(), "ggml_quantize_" + TYPE_NAMES[t])
    got = fn(x.ctypes.data, out.ctypes.data, x.size, k, hist.ctypes.data)
    assert got == out.size, (got, out.size)
    return out


def quantize_on_device(t, x):
    """ggml_hip_quantize: the same bytes as quantize(), computed on the GPU.  Returns (raw bytes, 16-bin histogram)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(row_bytes(t, x.size), dtype=np.uint8)
    hist = np.zeros(16, dtype=np.int64)
    got = lib().ggml_hip_quantize(t, x.ctypes.data, out.ctypes.data, x.size, x.shape[-1], hist.ctypes.data)
    assert got == out.size, (got, out.size)
    return out, hist


def topk(tensor, row, k, extra_ids=()):
    """ggml_hip_topk on a node of the last computed graph: (values, ids) of the k largest entries of `row` (descending,
    lower id first on ties) followed by the entries of extra_ids."""
    extra = np.ascontiguousarray(extra_ids, dtype=np.int32)
    vals = np.zeros(k + extra.size, dtype=np.float32)
    ids = np.zeros(k + extra.size, dtype=np.int32)
    ptr = tensor.ptr if isinstance(tensor, Tensor) else tensor
    rc = lib().ggml_hip_topk(ptr, row, k, extra.ctypes.data if extra.size else None, extra.size, vals.ctypes.data, ids.ctypes.data)
    if rc != 0:
        raise ValueError("ggml_hip_topk: bad arguments")
    return vals, ids


class Tensor:
    """crates/ggml/src/tensor.rs — thin handle over a ggml_tensor*."""

    def __init__(self, ptr, ctx):
        assert bool(ptr), "Should not be null"
        self.ptr = ptr
        self.ctx = ctx

    @property
    def t(self):
        return self.ptr.contents

    @property
    def ne(self):
        return tuple(self.t.ne)

    @property
    def nb(self):
        return tuple(self.t.nb)

    def nbytes(self):
        return lib().ggml_nbytes(self.ptr)

    def nelements(self):
        return lib().ggml_nelements(self.ptr)

    def set_name(self, name):
        lib().ggml_set_name(self.ptr, name.encode())
        return self

    def name(self):
        return lib().ggml_get_name(self.ptr).decode()

    def write_data(self, arr):
        """Tensor::write_data: copies into the HOST data of the tensor."""
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes(), (arr.nbytes, self.nbytes())
        C.memmove(self.t.data, arr.ctypes.data, arr.nbytes)
        return self

    def read_data(self, dtype=np.float32, count=None):
        """Tensor::read_data: reads the HOST data pointer (valid after compute for CPU-backend nodes)."""
        n = self.nbytes() if count is None else count * np.dtype(dtype).itemsize
        buf = (C.c_char * n).from_address(self.t.data)
        return np.frombuffer(buf, dtype=dtype).copy()

    def device_get(self, dtype=np.float32):
        """Extension: reads the device mirror (any node, any backend flag)."""
        out = np.zeros(self.nbytes() // np.dtype(dtype).itemsize, dtype=dtype)
        lib().ggml_hip_tensor_get(self.ptr, out.ctypes.data, 0, out.nbytes)
        return out

    def transfer_to_gpu(self):
        """Tensor::transfer_to(Backend::Gpu): tensor.rs:56-80."""
        self.t.backend = BACKEND_GPU
        lib().ggml_hip_transform_tensor(self.t.data, self.ptr)
        self.ctx._offloaded.append(self)
        return self

    def offload(self):
        lib().ggml_hip_assign_buffers(self.ptr)
        return self

    def offload_no_scratch(self):
        lib().ggml_hip_assign_buffers_no_scratch(self.ptr)
        self.ctx._offloaded.append(self)
        return self


class Context:
    """crates/ggml/src/context.rs — RAII over ggml_init / ggml_free, op_* builders."""

    def __init__(self, mem_size, no_alloc=False):
        self._buf = None
        p = ggml_init_params(mem_size, None, no_alloc)
        self.ptr = lib().ggml_init(p)
        assert self.ptr, "Should not be null"
        self._offloaded = []
        self._keep = []

    def free(self):
        if self.ptr:
            for t in self._offloaded:
                lib().ggml_hip_free_data(t.ptr)
            self._offloaded = []
            lib().ggml_free(self.ptr)
            self.ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.free()

    def _w(self, p):
        return Tensor(p, self)

    def new_tensor(self, typ, *ne):
        arr = (C.c_int64 * len(ne))(*ne)
        return self._w(lib().ggml_new_tensor(self.ptr, typ, len(ne), arr))

    def new_f32(self, x):
        return self._w(lib().ggml_new_f32(self.ptr, x))

    def tensor_from(self, arr, typ=None, ne=None):
        """Creates a tensor holding `arr` (numpy; raw bytes for quantized types)."""
        arr = np.ascontiguousarray(arr)
        if typ is None:
            typ = {np.dtype(np.float32): TYPE_F32, np.dtype(np.float16): TYPE_F16, np.dtype(np.int32): TYPE_I32}[
                arr.dtype]
        if ne is None:
            ne = tuple(reversed(arr.shape))
        t = self.new_tensor(typ, *ne)
        return t.write_data(arr)

    def use_scratch(self, data, size):
        return lib().ggml_set_scratch(self.ptr, ggml_scratch(0, size, data))

    def __getattr__(self, name):
        # op_xxx(...) -> ggml_xxx(ctx, ...) with Tensor handles unwrapped
        if name.startswith("op_"):
            fn = getattr(lib(), "ggml_" + name[3:])

            def call(*args):
                raw = [a.ptr if isinstance(a, Tensor) else a for a in args]
                return self._w(fn(self.ptr, *raw))

            return call
        raise AttributeError(name)

    def graph(self):
        return Graph(self)


class Graph:
    """ComputationGraph + GraphExecutionPlan (crates/ggml/src/lib.rs:322-378)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.ptr = lib().ggml_new_graph(ctx.ptr)

    def build_forward_expand(self, t):
        lib().ggml_build_forward_expand(self.ptr, t.ptr)
        return self

    @property
    def n_nodes(self):
        return self.ptr.contents.n_nodes

    @property
    def n_leafs(self):
        return self.ptr.contents.n_leafs

    def node(self, i):
        return Tensor(self.ptr.contents.nodes[i], self.ctx)

    def compute(self, n_threads=1):
        plan = lib().ggml_graph_plan(self.ptr, n_threads)
        work = self.ctx.new_tensor(TYPE_I8, max(plan.work_size, 0))
        plan.work_data = work.t.data
        return lib().ggml_graph_compute(self.ptr, C.byref(plan))


def bench_plan_class(kclass, replays):
    """(ms_total, launches_per_replay, algo_bytes_per_replay) of one kernel class of the last decode plan."""
    ms, n, b = C.c_double(0), C.c_int64(0), C.c_double(0)
    rc = lib().ggml_hip_bench_plan_class(kclass, replays, C.byref(ms), C.byref(n), C.byref(b))
    if rc != 0:
        raise RuntimeError("no fused decode plan to benchmark (the decode graph was not recognised)")
    return ms.value, n.value, b.value


def bench_empty(wgs, threads, lds_bytes, kernarg_bytes, n_launch=64, replays=20):
    """(us per launch, us end -> next first instruction, us first instruction -> kernel arguments) of a do-nothing
    kernel with the given launch shape (ggml_hip_bench_empty)."""
    out = (C.c_double * 3)()
    if lib().ggml_hip_bench_empty(wgs, threads, lds_bytes, kernarg_bytes, n_launch, replays, out) != 0:
        raise ValueError("bench_empty: bad launch shape")
    return tuple(out)


def timing_query(kclass):
    ms, n, b = C.c_double(0), C.c_int64(0), C.c_double(0)
    lib().ggml_hip_timing_query(kclass, C.byref(ms), C.byref(n), C.byref(b))
    return ms.value, n.value, b.value


def get_stat(key):
    """Backend counters (plan_tokens, graph_replays, plans, generic_graphs, ns_match, ns_launch, ns_wait, ns_compute)."""
    return int(lib().ggml_hip_get_stat(key.encode()))


def set_option(key, value):
    lib().ggml_hip_set_option(key.encode(), int(value))


def read_timeline(max_records=4096):
    """int64 [n, 8] records of the decode mat-vec timeline (see ggml_hip_read_timeline)."""
    import numpy as np
    buf = np.zeros((max_records, 8), np.int64)
    n = lib().ggml_hip_read_timeline(buf.ctypes.data, max_records)
    return buf[:n]
