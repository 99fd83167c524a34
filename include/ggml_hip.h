/*
 * ggml_hip.h — C ABI of libggml_hip.so, the MI355X-native drop-in for the one hot path of
 * rustformers/llm: ggml_compute_forward_mul_mat over Q4_0/Q4_1/Q5_0/Q5_1/Q8_0 (+F16 attention
 * matmuls), RMSNorm, RoPE, scale/mask/softmax, the KV-cache copy and the elementwise glue of the
 * LLaMA graph.
 *
 * Every declaration below replaces one `extern "C"` item of the reference's FFI layer
 * (crates/ggml/sys/src/lib.rs and crates/ggml/sys/src/cuda.rs — bindgen output); the comment on
 * each item names the reference line it stands in for.  Struct layouts are byte-identical to the
 * bindgen layout tests (sizes/offsets are static_assert'ed at the bottom of this file), so the
 * unmodified `crates/ggml` Rust wrapper can link against this library in place of the C sources
 * that crates/ggml/sys/build.rs:12-17 compiles.
 *
 * Plain C: pointers, sizes, POD structs.  No torch / HIP types cross this boundary.
 * Error behaviour follows ggml: no error returns — a failed assertion or HIP error prints a
 * message to stderr and abort()s (reference: SURVEY.md §8b "Errors").
 */
#ifndef GGML_HIP_H
#define GGML_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_API __attribute__((visibility("default")))

/* ---- constants: crates/ggml/sys/src/lib.rs:16-33 ------------------------------------------- */
#define GGML_FILE_MAGIC 0x67676d6c
#define GGML_FILE_VERSION 1
#define GGML_QNT_VERSION 2
#define GGML_QNT_VERSION_FACTOR 1000
#define GGML_MAX_DIMS 4
#define GGML_MAX_NODES 4096
#define GGML_MAX_PARAMS 256
#define GGML_MAX_CONTEXTS 64
#define GGML_MAX_SRC 6
#define GGML_MAX_NAME 48
#define GGML_MAX_OP_PARAMS 32
#define GGML_DEFAULT_N_THREADS 4
#define GGML_EXIT_SUCCESS 0
#define GGML_EXIT_ABORTED 1
#define GGML_GRAPH_HASHTABLE_SIZE 8273
#define GGML_MEM_ALIGN 16

/* crates/ggml/sys/src/llama.rs:15 (LLAMA_DEFAULT_RMS_EPS) */
#define LLAMA_DEFAULT_RMS_EPS 5e-6f

typedef uint16_t ggml_fp16_t; /* lib.rs:34 */

/* ---- enums: lib.rs:51-160 ------------------------------------------------------------------ */
enum ggml_type {
    GGML_TYPE_F32 = 0,
    GGML_TYPE_F16 = 1,
    GGML_TYPE_Q4_0 = 2,
    GGML_TYPE_Q4_1 = 3,
    /* 4, 5: removed upstream (Q4_2, Q4_3) */
    GGML_TYPE_Q5_0 = 6,
    GGML_TYPE_Q5_1 = 7,
    GGML_TYPE_Q8_0 = 8,
    GGML_TYPE_Q8_1 = 9,
    GGML_TYPE_Q2_K = 10,
    GGML_TYPE_Q3_K = 11,
    GGML_TYPE_Q4_K = 12,
    GGML_TYPE_Q5_K = 13,
    GGML_TYPE_Q6_K = 14,
    GGML_TYPE_Q8_K = 15,
    GGML_TYPE_I8 = 16,
    GGML_TYPE_I16 = 17,
    GGML_TYPE_I32 = 18,
    GGML_TYPE_COUNT = 19,
};

enum ggml_backend { /* lib.rs:70-73 */
    GGML_BACKEND_CPU = 0,
    GGML_BACKEND_GPU = 10,
    GGML_BACKEND_GPU_SPLIT = 20,
};

enum ggml_ftype { /* lib.rs:74-88 */
    GGML_FTYPE_UNKNOWN = -1,
    GGML_FTYPE_ALL_F32 = 0,
    GGML_FTYPE_MOSTLY_F16 = 1,
    GGML_FTYPE_MOSTLY_Q4_0 = 2,
    GGML_FTYPE_MOSTLY_Q4_1 = 3,
    GGML_FTYPE_MOSTLY_Q4_1_SOME_F16 = 4,
    GGML_FTYPE_MOSTLY_Q8_0 = 7,
    GGML_FTYPE_MOSTLY_Q5_0 = 8,
    GGML_FTYPE_MOSTLY_Q5_1 = 9,
};

enum ggml_op { /* lib.rs:89-149 */
    GGML_OP_NONE = 0,
    GGML_OP_DUP,
    GGML_OP_ADD,
    GGML_OP_ADD1,
    GGML_OP_ACC,
    GGML_OP_SUB,
    GGML_OP_MUL,
    GGML_OP_DIV,
    GGML_OP_SQR,
    GGML_OP_SQRT,
    GGML_OP_LOG,
    GGML_OP_SUM,
    GGML_OP_SUM_ROWS,
    GGML_OP_MEAN,
    GGML_OP_ARGMAX,
    GGML_OP_REPEAT,
    GGML_OP_REPEAT_BACK,
    GGML_OP_SILU_BACK,
    GGML_OP_NORM,
    GGML_OP_RMS_NORM,
    GGML_OP_RMS_NORM_BACK,
    GGML_OP_MUL_MAT, /* = 21 */
    GGML_OP_OUT_PROD,
    GGML_OP_SCALE,
    GGML_OP_SET,
    GGML_OP_CPY,
    GGML_OP_CONT,
    GGML_OP_RESHAPE,
    GGML_OP_VIEW,
    GGML_OP_PERMUTE,
    GGML_OP_TRANSPOSE,
    GGML_OP_GET_ROWS,
    GGML_OP_GET_ROWS_BACK,
    GGML_OP_DIAG,
    GGML_OP_DIAG_MASK_INF,
    GGML_OP_DIAG_MASK_ZERO,
    GGML_OP_SOFT_MAX,
    GGML_OP_SOFT_MAX_BACK,
    GGML_OP_ROPE,
    GGML_OP_ROPE_BACK,
    GGML_OP_ALIBI,
    GGML_OP_CLAMP,
    GGML_OP_CONV_1D,
    GGML_OP_CONV_2D,
    GGML_OP_POOL_1D,
    GGML_OP_POOL_2D,
    GGML_OP_FLASH_ATTN,
    GGML_OP_FLASH_FF,
    GGML_OP_FLASH_ATTN_BACK,
    GGML_OP_WIN_PART,
    GGML_OP_WIN_UNPART,
    GGML_OP_UNARY, /* = 51 */
    GGML_OP_MAP_UNARY,
    GGML_OP_MAP_BINARY,
    GGML_OP_MAP_CUSTOM1,
    GGML_OP_MAP_CUSTOM2,
    GGML_OP_MAP_CUSTOM3,
    GGML_OP_CROSS_ENTROPY_LOSS,
    GGML_OP_CROSS_ENTROPY_LOSS_BACK,
    GGML_OP_COUNT, /* = 59 */
};

enum ggml_unary_op { /* lib.rs:150-160 */
    GGML_UNARY_OP_ABS = 0,
    GGML_UNARY_OP_SGN,
    GGML_UNARY_OP_NEG,
    GGML_UNARY_OP_STEP,
    GGML_UNARY_OP_TANH,
    GGML_UNARY_OP_ELU,
    GGML_UNARY_OP_RELU,
    GGML_UNARY_OP_GELU,
    GGML_UNARY_OP_GELU_QUICK,
    GGML_UNARY_OP_SILU,
};

enum ggml_object_type { /* lib.rs:161-164 */
    GGML_OBJECT_TENSOR = 0,
    GGML_OBJECT_GRAPH = 1,
    GGML_OBJECT_WORK_BUFFER = 2,
};

enum ggml_task_type { /* lib.rs:756-759 */
    GGML_TASK_INIT = 0,
    GGML_TASK_COMPUTE = 1,
    GGML_TASK_FINALIZE = 2,
};

/* ---- structs ------------------------------------------------------------------------------- */
struct ggml_context; /* opaque, lib.rs:47-50 */

struct ggml_object { /* lib.rs:167-173, 32 bytes */
    size_t offs;
    size_t size;
    struct ggml_object *next;
    enum ggml_object_type type;
    char padding[4];
};

struct ggml_tensor { /* lib.rs:242-260, 272 bytes */
    enum ggml_type type;
    enum ggml_backend backend;
    int n_dims;
    int64_t ne[GGML_MAX_DIMS]; /* number of elements */
    size_t nb[GGML_MAX_DIMS];  /* stride in bytes: nb[0]=type size, nb[i]=nb[i-1]*ne[i-1] (blocks for q) */
    enum ggml_op op;
    int32_t op_params[GGML_MAX_OP_PARAMS / sizeof(int32_t)];
    bool is_param;
    struct ggml_tensor *grad;
    struct ggml_tensor *src[GGML_MAX_SRC];
    int perf_runs;
    int64_t perf_cycles;
    int64_t perf_time_us;
    void *data;
    char name[GGML_MAX_NAME];
    void *extra; /* backend handle: device allocation record of the HIP backend */
    char padding[4];
};

struct ggml_cplan { /* lib.rs:449-457, 16424 bytes */
    size_t work_size;
    uint8_t *work_data;
    int n_threads;
    int n_tasks[GGML_MAX_NODES];
    bool (*abort_callback)(void *data);
    void *abort_callback_data;
};

struct ggml_cgraph { /* lib.rs:535-545, 164520 bytes */
    int n_nodes;
    int n_leafs;
    struct ggml_tensor *nodes[GGML_MAX_NODES];
    struct ggml_tensor *grads[GGML_MAX_NODES];
    struct ggml_tensor *leafs[GGML_MAX_NODES];
    void *visited_hash_table[GGML_GRAPH_HASHTABLE_SIZE];
    int perf_runs;
    int64_t perf_cycles;
    int64_t perf_time_us;
};

struct ggml_scratch { /* lib.rs:654-658 */
    size_t offs;
    size_t size;
    void *data;
};

struct ggml_init_params { /* lib.rs:706-710 */
    size_t mem_size;
    void *mem_buffer;
    bool no_alloc;
};

struct ggml_compute_params { /* lib.rs:762-768 */
    enum ggml_task_type type;
    int ith, nth;
    size_t wsize;
    void *wdata;
};

typedef void (*ggml_to_float_t)(const void *x, float *y, int k);   /* lib.rs:2884 */
typedef void (*ggml_from_float_t)(const float *x, void *y, int k); /* lib.rs:2887 */
typedef void (*ggml_vec_dot_t)(int n, float *s, const void *x, const void *y); /* lib.rs:2890 */
typedef struct { /* lib.rs:2900-2906, 40 bytes */
    ggml_to_float_t to_float;
    ggml_from_float_t from_float;
    ggml_from_float_t from_float_reference;
    ggml_vec_dot_t vec_dot;
    enum ggml_type vec_dot_type;
} ggml_type_traits_t;

typedef void (*ggml_unary_op_f32_t)(const int, float *, const float *);
typedef void (*ggml_binary_op_f32_t)(const int, float *, const float *, const float *);

/* ---- core: context / arena (lib.rs:36-45, 916-940) ----------------------------------------- */
GGML_API float ggml_fp16_to_fp32(ggml_fp16_t x);
GGML_API ggml_fp16_t ggml_fp32_to_fp16(float x);
GGML_API void ggml_fp16_to_fp32_row(const ggml_fp16_t *x, float *y, int n);
GGML_API void ggml_fp32_to_fp16_row(const float *x, ggml_fp16_t *y, int n);

GGML_API struct ggml_context *ggml_init(struct ggml_init_params params); /* lib.rs:916 */
GGML_API void ggml_free(struct ggml_context *ctx);
GGML_API size_t ggml_used_mem(const struct ggml_context *ctx);
GGML_API size_t ggml_set_scratch(struct ggml_context *ctx, struct ggml_scratch scratch); /* :925 */
GGML_API bool ggml_get_no_alloc(struct ggml_context *ctx);
GGML_API void ggml_set_no_alloc(struct ggml_context *ctx, bool no_alloc);
GGML_API void *ggml_get_mem_buffer(const struct ggml_context *ctx);
GGML_API size_t ggml_get_mem_size(const struct ggml_context *ctx);
GGML_API size_t ggml_get_max_tensor_size(const struct ggml_context *ctx);
GGML_API void ggml_print_objects(const struct ggml_context *ctx);

/* ---- core: type and tensor introspection ---------------------------------------------------- */
GGML_API int64_t ggml_nelements(const struct ggml_tensor *t);
GGML_API int64_t ggml_nrows(const struct ggml_tensor *t);
GGML_API size_t ggml_nbytes(const struct ggml_tensor *t);
GGML_API int ggml_blck_size(enum ggml_type type);
GGML_API size_t ggml_type_size(enum ggml_type type);
GGML_API float ggml_type_sizef(enum ggml_type type);
GGML_API const char *ggml_type_name(enum ggml_type type);
GGML_API const char *ggml_op_name(enum ggml_op op);
GGML_API size_t ggml_element_size(const struct ggml_tensor *t);
GGML_API bool ggml_is_quantized(enum ggml_type type);
GGML_API bool ggml_is_transposed(const struct ggml_tensor *t);
GGML_API bool ggml_is_contiguous(const struct ggml_tensor *t);
GGML_API bool ggml_is_permuted(const struct ggml_tensor *t);
GGML_API size_t ggml_tensor_overhead(void);
GGML_API void *ggml_get_data(const struct ggml_tensor *t);
GGML_API float *ggml_get_data_f32(const struct ggml_tensor *t);
GGML_API const char *ggml_get_name(const struct ggml_tensor *t);
GGML_API struct ggml_tensor *ggml_set_name(struct ggml_tensor *t, const char *name);

/* ---- core: tensor constructors -------------------------------------------------------------- */
GGML_API struct ggml_tensor *ggml_new_tensor(struct ggml_context *ctx, enum ggml_type type, int n_dims,
                                             const int64_t *ne);
GGML_API struct ggml_tensor *ggml_new_tensor_1d(struct ggml_context *ctx, enum ggml_type type, int64_t ne0);
GGML_API struct ggml_tensor *ggml_new_tensor_2d(struct ggml_context *ctx, enum ggml_type type, int64_t ne0,
                                                int64_t ne1);
GGML_API struct ggml_tensor *ggml_new_tensor_3d(struct ggml_context *ctx, enum ggml_type type, int64_t ne0,
                                                int64_t ne1, int64_t ne2);
GGML_API struct ggml_tensor *ggml_new_tensor_4d(struct ggml_context *ctx, enum ggml_type type, int64_t ne0,
                                                int64_t ne1, int64_t ne2, int64_t ne3);
GGML_API struct ggml_tensor *ggml_new_i32(struct ggml_context *ctx, int32_t value);
GGML_API struct ggml_tensor *ggml_new_f32(struct ggml_context *ctx, float value);
GGML_API struct ggml_tensor *ggml_dup_tensor(struct ggml_context *ctx, const struct ggml_tensor *src);
GGML_API struct ggml_tensor *ggml_view_tensor(struct ggml_context *ctx, const struct ggml_tensor *src);

/* ---- core: graph builders (the operator surface of crates/ggml/src/context.rs:276-626) ------- */
GGML_API struct ggml_tensor *ggml_dup(struct ggml_context *ctx, struct ggml_tensor *a);
GGML_API struct ggml_tensor *ggml_add(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b);
GGML_API struct ggml_tensor *ggml_add_inplace(struct ggml_context *ctx, struct ggml_tensor *a,
                                              struct ggml_tensor *b);
GGML_API struct ggml_tensor *ggml_mul(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b);
GGML_API struct ggml_tensor *ggml_repeat(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b);
GGML_API struct ggml_tensor *ggml_silu(struct ggml_context *ctx, struct ggml_tensor *a);
GGML_API struct ggml_tensor *ggml_gelu(struct ggml_context *ctx, struct ggml_tensor *a);
GGML_API struct ggml_tensor *ggml_norm(struct ggml_context *ctx, struct ggml_tensor *a);
GGML_API struct ggml_tensor *ggml_rms_norm(struct ggml_context *ctx, struct ggml_tensor *a, float eps); /* :1265 */
GGML_API struct ggml_tensor *ggml_mul_mat(struct ggml_context *ctx, struct ggml_tensor *a,
                                          struct ggml_tensor *b); /* lib.rs:1283-1288 — THE hot op */
GGML_API struct ggml_tensor *ggml_scale(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b);
GGML_API struct ggml_tensor *ggml_scale_inplace(struct ggml_context *ctx, struct ggml_tensor *a,
                                                struct ggml_tensor *b); /* :1304 */
GGML_API struct ggml_tensor *ggml_cpy(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b);
GGML_API struct ggml_tensor *ggml_cont(struct ggml_context *ctx, struct ggml_tensor *a);
GGML_API struct ggml_tensor *ggml_reshape(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b);
GGML_API struct ggml_tensor *ggml_reshape_1d(struct ggml_context *ctx, struct ggml_tensor *a, int64_t ne0);
GGML_API struct ggml_tensor *ggml_reshape_2d(struct ggml_context *ctx, struct ggml_tensor *a, int64_t ne0,
                                             int64_t ne1);
GGML_API struct ggml_tensor *ggml_reshape_3d(struct ggml_context *ctx, struct ggml_tensor *a, int64_t ne0,
                                             int64_t ne1, int64_t ne2);
GGML_API struct ggml_tensor *ggml_view_1d(struct ggml_context *ctx, struct ggml_tensor *a, int64_t ne0,
                                          size_t offset);
GGML_API struct ggml_tensor *ggml_view_2d(struct ggml_context *ctx, struct ggml_tensor *a, int64_t ne0,
                                          int64_t ne1, size_t nb1, size_t offset);
GGML_API struct ggml_tensor *ggml_view_3d(struct ggml_context *ctx, struct ggml_tensor *a, int64_t ne0,
                                          int64_t ne1, int64_t ne2, size_t nb1, size_t nb2,
                                          size_t offset); /* :1446 */
GGML_API struct ggml_tensor *ggml_permute(struct ggml_context *ctx, struct ggml_tensor *a, int axis0, int axis1,
                                          int axis2, int axis3);
GGML_API struct ggml_tensor *ggml_transpose(struct ggml_context *ctx, struct ggml_tensor *a);
GGML_API struct ggml_tensor *ggml_get_rows(struct ggml_context *ctx, struct ggml_tensor *a,
                                           struct ggml_tensor *b); /* :1485 */
GGML_API struct ggml_tensor *ggml_diag_mask_inf(struct ggml_context *ctx, struct ggml_tensor *a, int n_past);
GGML_API struct ggml_tensor *ggml_diag_mask_inf_inplace(struct ggml_context *ctx, struct ggml_tensor *a,
                                                        int n_past); /* :1510 */
GGML_API struct ggml_tensor *ggml_soft_max(struct ggml_context *ctx, struct ggml_tensor *a);
GGML_API struct ggml_tensor *ggml_soft_max_inplace(struct ggml_context *ctx, struct ggml_tensor *a);
GGML_API struct ggml_tensor *ggml_rope(struct ggml_context *ctx, struct ggml_tensor *a, int n_past, int n_dims,
                                       int mode, int n_ctx); /* :1561 */
GGML_API struct ggml_tensor *ggml_rope_inplace(struct ggml_context *ctx, struct ggml_tensor *a, int n_past,
                                               int n_dims, int mode, int n_ctx);
GGML_API struct ggml_tensor *ggml_rope_custom_inplace(struct ggml_context *ctx, struct ggml_tensor *a,
                                                      int n_past, int n_dims, int mode, int n_ctx,
                                                      float freq_base, float freq_scale); /* :1583 */
/* Declared by the reference surface but outside the accelerated path: these abort with a message
 * (reference: crates/ggml/src/context.rs:592-626, 385-427; used only by bloom/mpt/none). */
GGML_API struct ggml_tensor *ggml_alibi(struct ggml_context *ctx, struct ggml_tensor *a, int n_past, int n_head,
                                        float bias_max);
GGML_API struct ggml_tensor *ggml_flash_attn(struct ggml_context *ctx, struct ggml_tensor *q,
                                             struct ggml_tensor *k, struct ggml_tensor *v, bool masked);
GGML_API struct ggml_tensor *ggml_map_unary_f32(struct ggml_context *ctx, struct ggml_tensor *a,
                                                ggml_unary_op_f32_t fun);
GGML_API struct ggml_tensor *ggml_map_binary_f32(struct ggml_context *ctx, struct ggml_tensor *a,
                                                 struct ggml_tensor *b, ggml_binary_op_f32_t fun);

/* ---- core: graph build / plan / compute (lib.rs:1877-1899; caller crates/ggml/src/lib.rs:332-376) */
GGML_API struct ggml_cgraph *ggml_new_graph(struct ggml_context *ctx);
GGML_API size_t ggml_graph_overhead(void);
GGML_API void ggml_build_forward_expand(struct ggml_cgraph *cgraph, struct ggml_tensor *tensor);
GGML_API struct ggml_cgraph ggml_build_forward(struct ggml_tensor *tensor);
GGML_API struct ggml_cplan ggml_graph_plan(struct ggml_cgraph *cgraph, int n_threads); /* by value (sret) */
/* Executes every node of the graph on the MI355X.  There is no CPU compute path in this library:
 * an op outside the supported set aborts with a message.  Returns GGML_EXIT_SUCCESS. */
GGML_API int ggml_graph_compute(struct ggml_cgraph *cgraph, struct ggml_cplan *cplan);
GGML_API void ggml_graph_reset(struct ggml_cgraph *cgraph);

/* ---- core: offline quantizer (lib.rs:2779-2822; caller crates/ggml/src/lib.rs:419-483) ------- */
GGML_API size_t ggml_quantize_q4_0(const float *src, void *dst, int n, int k, int64_t *hist);
GGML_API size_t ggml_quantize_q4_1(const float *src, void *dst, int n, int k, int64_t *hist);
GGML_API size_t ggml_quantize_q5_0(const float *src, void *dst, int n, int k, int64_t *hist);
GGML_API size_t ggml_quantize_q5_1(const float *src, void *dst, int n, int k, int64_t *hist);
GGML_API size_t ggml_quantize_q8_0(const float *src, void *dst, int n, int k, int64_t *hist);
GGML_API size_t ggml_quantize_chunk(enum ggml_type type, const float *src, void *dst, int start, int n,
                                    int64_t *hist);
GGML_API ggml_type_traits_t ggml_internal_get_type_traits(enum ggml_type i); /* lib.rs:2973 */

GGML_API int ggml_cpu_has_blas(void);
GGML_API int ggml_cpu_has_gpublas(void);

/* ---- accelerator hooks: one per entry of crates/ggml/sys/src/cuda.rs:6-77 -------------------- */
#define GGML_HIP_MAX_DEVICES 16 /* cuda.rs:6 GGML_CUDA_MAX_DEVICES */

GGML_API void ggml_init_hipblas(void);                                 /* cuda.rs:8  ggml_init_cublas */
GGML_API void ggml_hip_set_tensor_split(const float *tensor_split);    /* cuda.rs:11 */
GGML_API void ggml_hip_mul(const struct ggml_tensor *src0, const struct ggml_tensor *src1,
                           struct ggml_tensor *dst);                   /* cuda.rs:14 */
GGML_API bool ggml_hip_can_mul_mat(const struct ggml_tensor *src0, const struct ggml_tensor *src1,
                                   struct ggml_tensor *dst);           /* cuda.rs:17-22 */
GGML_API size_t ggml_hip_mul_mat_get_wsize(const struct ggml_tensor *src0, const struct ggml_tensor *src1,
                                           struct ggml_tensor *dst);   /* cuda.rs:24-29 */
GGML_API void ggml_hip_mul_mat(const struct ggml_tensor *src0, const struct ggml_tensor *src1,
                               struct ggml_tensor *dst, void *wdata, size_t wsize); /* cuda.rs:31-38 */
GGML_API void *ggml_hip_host_malloc(size_t size);                      /* cuda.rs:41 */
GGML_API void ggml_hip_host_free(void *ptr);                           /* cuda.rs:44 */
GGML_API void ggml_hip_transform_tensor(void *data, struct ggml_tensor *tensor); /* cuda.rs:47 — H2D boundary */
GGML_API void ggml_hip_free_data(struct ggml_tensor *tensor);          /* cuda.rs:50 */
GGML_API void ggml_hip_assign_buffers(struct ggml_tensor *tensor);     /* cuda.rs:53 */
GGML_API void ggml_hip_assign_buffers_no_scratch(struct ggml_tensor *tensor);    /* cuda.rs:56 */
GGML_API void ggml_hip_assign_buffers_force_inplace(struct ggml_tensor *tensor); /* cuda.rs:59 */
GGML_API void ggml_hip_set_main_device(int main_device);               /* cuda.rs:62 */
GGML_API void ggml_hip_set_mul_mat_q(bool mul_mat_q);                  /* cuda.rs:65 */
GGML_API void ggml_hip_set_scratch_size(size_t scratch_size);          /* cuda.rs:68 */
GGML_API void ggml_hip_free_scratch(void);                             /* cuda.rs:71 */
GGML_API bool ggml_hip_compute_forward(struct ggml_compute_params *params, struct ggml_tensor *tensor); /* :73-76 */

/* The same 19 entry points under the names the reference's `cublas` cfg arms already bind
 * (crates/ggml/src/tensor.rs:68-71,89-92,106-109,214-217; accelerator/mod.rs:68-94), so the Rust
 * crate links against this library with zero source changes. */
GGML_API void ggml_init_cublas(void);
GGML_API void ggml_cuda_set_tensor_split(const float *tensor_split);
GGML_API void ggml_cuda_mul(const struct ggml_tensor *, const struct ggml_tensor *, struct ggml_tensor *);
GGML_API bool ggml_cuda_can_mul_mat(const struct ggml_tensor *, const struct ggml_tensor *, struct ggml_tensor *);
GGML_API size_t ggml_cuda_mul_mat_get_wsize(const struct ggml_tensor *, const struct ggml_tensor *,
                                            struct ggml_tensor *);
GGML_API void ggml_cuda_mul_mat(const struct ggml_tensor *, const struct ggml_tensor *, struct ggml_tensor *,
                                void *, size_t);
GGML_API void *ggml_cuda_host_malloc(size_t size);
GGML_API void ggml_cuda_host_free(void *ptr);
GGML_API void ggml_cuda_transform_tensor(void *data, struct ggml_tensor *tensor);
GGML_API void ggml_cuda_free_data(struct ggml_tensor *tensor);
GGML_API void ggml_cuda_assign_buffers(struct ggml_tensor *tensor);
GGML_API void ggml_cuda_assign_buffers_no_scratch(struct ggml_tensor *tensor);
GGML_API void ggml_cuda_assign_buffers_force_inplace(struct ggml_tensor *tensor);
GGML_API void ggml_cuda_set_main_device(int main_device);
GGML_API void ggml_cuda_set_mul_mat_q(bool mul_mat_q);
GGML_API void ggml_cuda_set_scratch_size(size_t scratch_size);
GGML_API void ggml_cuda_free_scratch(void);
GGML_API bool ggml_cuda_compute_forward(struct ggml_compute_params *params, struct ggml_tensor *tensor);

/* ---- HIP-backend extensions (no reference counterpart; measurement and multi-GPU plumbing) ---- */
/* Number of visible devices (0 when no GPU / HIP runtime unusable). Never aborts. */
GGML_API int ggml_hip_device_count(void);
/* The physical GPU a device slot drives (slot s -> (GGML_HIP_DEVICE + s) mod visible GPUs), -1 for a slot that does not exist. */
GGML_API int ggml_hip_slot_physical_device(int slot);
/* GGML_HIP_SESSION_SLOTS=n (opt-in): the first K/V memory a thread creates (ggml_hip_assign_buffers_no_scratch, i.e.
 * InferenceSession::new) assigns the thread one of the n sibling slots of the GPU and pins it there, so that sessions started on
 * several threads by an UNCHANGED caller (model.start_session() / infer()) overlap instead of sharing one slot's stream and lock.
 * Returns the calling thread's assignment, -1 before it has one. */
GGML_API int ggml_hip_thread_session_slot(void);
/* Several devices in one process (the ggml-style layer split of one InferenceSession).  A device
 * "slot" owns a stream, the device shadows of the host arenas, the weights uploaded while it was current and its plan cache;
 * ggml_hip_set_main_device (cuda.rs:62) makes a slot current for every following call.  GGML_HIP_VIRTUAL_DEVICES=n maps n
 * slots onto the visible GPUs round-robin (several slots on one GPU: how the split is tested on a 1-GPU box).
 * Threads: the current slot is per thread.  ggml_hip_set_main_device sets the process default (followed by every thread
 * that never chose a slot — the reference sets it once at load, accelerator/mod.rs:72) and pins the calling thread;
 * ggml_hip_bind_thread_device pins the calling thread only.  Entry points lock the slot they act on, not the library:
 * sessions on different slots run concurrently (crates/llm-base/src/inference_session.rs:43-48: sessions are Send and
 * one model serves several of them), calls on one slot are serialised.  ggml_hip_get_main_device = the caller's slot. */
GGML_API int ggml_hip_get_main_device(void);
GGML_API void ggml_hip_bind_thread_device(int device);
/* -1 while the calling thread follows the process default, else the slot it is pinned to; unbind drops the pin again (the
 * session mirror pins a model's slot around each call and must leave a thread that followed the default following it). */
GGML_API int ggml_hip_thread_pinned_device(void);
GGML_API void ggml_hip_unbind_thread_device(void);
/* ggml_hip_set_tensor_split / ggml_cuda_set_tensor_split read exactly ONE float: the reference passes the address of a
 * single stack f32 (crates/ggml/src/accelerator/mod.rs:74-75).  get returns that value (out[0]; 1 written). */
GGML_API int ggml_hip_get_tensor_split(float *out, int cap);
/* A split of ONE model over several device slots of this process, by LAYERS (ggml's fractions convention: slot i takes
 * fractions[i] / sum; all zero = equal shares; rows of one tensor are never split).  Explicit length, so nothing is read
 * past the caller's array.  NULL or n <= 0 clears it.  llm_llama_new applies it (also: env GGML_HIP_LAYER_SPLIT=G for G
 * equal shares).  get returns the number of fractions written. */
GGML_API void ggml_hip_set_layer_split(const float *fractions, int n);
GGML_API int ggml_hip_get_layer_split(float *out, int cap);
/* dst (on slot dst_device) = src (on slot src_device), ordered after src_device's enqueued work and before dst_device's
 * later work: the residual hop of a layer split inside one process (peer copy over xGMI between two GPUs). */
GGML_API void ggml_hip_copy_between_devices(int dst_device, void *dst, int src_device, const void *src, size_t nbytes);
/* Blocks until all device work queued by this library has finished. */
GGML_API void ggml_hip_synchronize(void);
/* Copies nbytes of a tensor's device mirror to/from host memory (host pointer != tensor->data allowed).
 * Used by tests to inspect interior nodes and by the layer-split driver to hand the residual to RCCL. */
GGML_API void ggml_hip_tensor_get(const struct ggml_tensor *tensor, void *host_dst, size_t offset, size_t nbytes);
GGML_API void ggml_hip_tensor_set(struct ggml_tensor *tensor, const void *host_src, size_t offset, size_t nbytes);
/* Raw synchronous copy on the backend stream; kind: 0 = host->device, 1 = device->host, 2 = device->device. */
GGML_API void ggml_hip_memcpy(void *dst, const void *src, size_t nbytes, int kind);
/* Device address of a tensor's mirror (for handing buffers to torch.distributed / RCCL). */
GGML_API void *ggml_hip_tensor_device_ptr(const struct ggml_tensor *tensor);
/* Per-kernel-class timing with HIP events on the backend's own stream (bench.py roofline leg).
 * classes: see GGML_HIP_KCLASS_*.  begin() resets and enables; end() disables; query returns
 * accumulated milliseconds and launch count for one class. */
enum ggml_hip_kclass {
    GGML_HIP_KCLASS_MMVQ = 0,    /* quantized mat-vec (decode hot kernel) */
    GGML_HIP_KCLASS_MMQ_MFMA,    /* quantized GEMM on MFMA (prefill) */
    GGML_HIP_KCLASS_ATTN,        /* f16 KV matmuls / fused attention */
    GGML_HIP_KCLASS_OTHER,       /* norm, rope, softmax, cpy, elementwise, quantize-activation */
    GGML_HIP_KCLASS_COUNT
};
GGML_API void ggml_hip_timing_begin(void);
GGML_API void ggml_hip_timing_end(void);
GGML_API void ggml_hip_timing_query(int kclass, double *ms, int64_t *launches, double *algo_bytes);
/* Execution mode knobs (the full list with meanings: INTEGRATION.md section 4 "Runtime knobs"): "fuse" (peephole fusion in the
 * generic executor), "plan" / "plan_k" / "plan_multi" / "plan_prompt" (fused LLaMA plans), "graph" (hipGraph replay of a plan),
 * "big", "kbig", "fuse_attn", "fuse_wo", "fuse_heads", "warm_mb", "affine", "attn_split" (decode attention split over positions:
 * from n positions on), "attn_one", "fused_fallback", "fused_rearm_tokens", "speculate_next", "act_quant", "mmq_min", "mmq_i8",
 * "mmq_w16", "w16_headroom_gb", "w16_release", "serial_stage_slots", "probe" (measurement only: the decode mat-vec returns early),
 * "timeline" (1 = 4 sampled workgroups per launch, n > 1 = n of them); an unknown key aborts with a message.
 * Also env GGML_HIP_FUSE / GGML_HIP_PLAN / GGML_HIP_GRAPH / GGML_HIP_BIG / GGML_HIP_WARM_MB / GGML_HIP_AFFINE / GGML_HIP_MMQ_*. */
GGML_API void ggml_hip_set_option(const char *key, int value);
/* Replays the launches of one kernel class of the most recent fused decode plan `replays` times from a
 * dedicated hipGraph between two HIP events on the backend stream (bench.py roofline leg). 0 on success.
 * kclass may also be GGML_HIP_KKIND_BASE + {0 wq|wk|wv, 1 wo, 2 w1|w3, 3 w2, 4 lm_head}: that mat-vec alone; or
 * GGML_HIP_KKIND_BASE + 8 + the same: every mat-vec launch BUT that kind (all - all_but_k = what kind k costs in sequence). */
#define GGML_HIP_KKIND_BASE 16
GGML_API int ggml_hip_bench_plan_class(int kclass, int replays, double *ms_total, int64_t *launches_per_replay,
                                       double *algo_bytes_per_replay);
/* Layer split over RCCL (the reference has no multi-GPU path: LLAMA_MAX_DEVICES = 1,
 * crates/ggml/sys/src/llama.rs:3, and feeds ggml_cuda_set_tensor_split a single float, crates/ggml/src/accelerator/mod.rs:74-75).
 * One process per GPU; rank 0 draws the id, the launcher hands it to every rank, every rank calls init.  send / recv are
 * enqueued on the backend's stream (ordered with the kernels before and after them, no host synchronisation);
 * sendrecv = both in one RCCL group (needed when the peer is this rank itself).  librccl.so is opened on first use. */
#define GGML_HIP_COMM_ID_BYTES 128
GGML_API int ggml_hip_comm_unique_id(void *id_out /* GGML_HIP_COMM_ID_BYTES */);
GGML_API int ggml_hip_comm_init(int rank, int world, const void *id); /* returns the rank count RCCL reports; -1 (with a
                                                                          message) if the communicator cannot be formed */
GGML_API void ggml_hip_comm_destroy(void);
GGML_API int ggml_hip_comm_ranks(void); /* 0 = no communicator */
GGML_API void ggml_hip_comm_send(const void *dev_src, size_t nbytes, int peer);
GGML_API void ggml_hip_comm_recv(void *dev_dst, size_t nbytes, int peer);
GGML_API void ggml_hip_comm_sendrecv(const void *dev_src, int send_peer, void *dev_dst, int recv_peer, size_t nbytes);
/* Launch-floor probe (measurement only, tests/tools/launch_probe.py): a linear hipGraph of n_launch launches of a
 * kernel that only stamps the device wall clock, with the given launch shape.  out[3] = {us per launch, us from one
 * launch's end to the next launch's first instruction, us first instruction -> last kernel argument usable}. */
GGML_API int ggml_hip_bench_empty(int wgs, int threads, int lds_bytes, int kernarg_bytes, int n_launch, int replays,
                                  double *out);
/* Counters for tests: "plan_tokens" (tokens run by the fused decode plan), "attn_split_tokens", "graph_replays", "plans",
 * "generic_graphs".  -1 for an unknown key. */
GGML_API int64_t ggml_hip_get_stat(const char *key);
/* ggml_graph_compute split in two so that a caller can overlap host work (building the next token's graph) with
 * the device: begin() enqueues the graph and returns 1 if it is still running on the device (fused decode plan),
 * 0 if it already completed (any other graph runs synchronously); end() waits and completes the copy of the
 * host-visible results (CPU-backend nodes).  No result may be read, and no other graph computed, in between. */
GGML_API int ggml_hip_graph_compute_begin(struct ggml_cgraph *cgraph);
GGML_API void ggml_hip_graph_compute_end(void);
/* Optional companion of the pair above: a caller that builds the NEXT token's graph between begin() and end() may hand it over
 * right away; the structural match of that graph against the fused decode plan (~20 us of host work per token) then happens
 * while the device runs instead of between two tokens' device work.  Returns 1 if the graph was recognised and remembered (its
 * begin() then skips the match), 0 otherwise (nothing changes).  The remembered match is dropped by anything that could
 * invalidate it (ggml_init / ggml_free / ggml_set_scratch on any thread, an option change, freed weights).  The token id itself
 * is read at begin(), as always. */
GGML_API int ggml_hip_graph_prepare(struct ggml_cgraph *cgraph);
/* Greedy sampling on the device (the step after the path): decodes n more tokens — each the first
 * argmax of the previous logits, what `infer_next_token` with a greedy sampler yields (inference_session.rs:381-424,
 * samplers.rs:289-306) — without the per-token logits read-back and host sync.  `last` = the cgraph of the caller's
 * most recent ggml_graph_compute, which must have been a single-token LLaMA evaluation executed by the fused plan.
 * out_tokens[n] receives the ids; last_logits (V floats, may be NULL) the logits after the n-th token.  The K/V
 * memory advances by n positions; the caller adds n to its n_past.  Returns 0, or -1 if the precondition does not
 * hold (nothing was executed: decode token by token instead). */
GGML_API int ggml_hip_decode_greedy_chain(struct ggml_cgraph *last, int n, int32_t *out_tokens, float *last_logits);
/* Top-k prefilter of a logits row on the device: the k (<= 1024, <= ne0) largest entries of row `row` of
 * the f32 tensor `t` — a node of the caller's most recent ggml_graph_compute, normally the logits — as (value, id) pairs,
 * value descending, lower id first among equal values, followed by the entries of `extra_ids` (n_extra ids, e.g. the
 * tokens a repetition penalty or a bias list touches) in the order given.  out_vals / out_ids hold k + n_extra entries.
 * For the reference's sampler chain (crates/llm-base/src/samplers.rs:289-306 sample_token; top-k after flat bias and
 * repetition penalty): the tokens that can be among the k best after the host has changed the n_extra listed logits
 * are all in {raw top-(k + n_extra)} U extra_ids, so a caller asks for k + n_extra and sorts at most k + 2 n_extra
 * pairs instead of n_vocab logits; only those pairs cross PCIe.  Returns 0, -1 on bad arguments. */
GGML_API int ggml_hip_topk(const struct ggml_tensor *t, int64_t row, int k, const int32_t *extra_ids, int n_extra,
                           float *out_vals, int32_t *out_ids);
/* ggml_quantize_q4_0 / q4_1 / q5_0 / q5_1 / q8_0 (crates/ggml/src/lib.rs:419-483, called by
 * crates/llm-base/src/quantize.rs:363-379) computed on the device: n f32 values at `src` (host memory, rows
 * of k, k % 32 == 0) -> raw GGML blocks at `dst` (host memory), byte-identical to the host functions; the 16-bin
 * histogram is ADDED to hist (may be NULL).  Returns the bytes written.  PCIe-bound: ~1.2 bytes moved per weight. */
GGML_API size_t ggml_hip_quantize(enum ggml_type type, const float *src, void *dst, int64_t n, int64_t k, int64_t *hist);
/* The same for a matrix that already lives in HBM: `src` is a contiguous 2-D f32 or f16 tensor previously handed to
 * ggml_hip_transform_tensor; `dst` is a tensor of a 32-wide block type with the same shape whose data pointer names the
 * result (its host bytes are neither read nor written).  After the call dst is a device-resident quantized weight
 * (GGML_BACKEND_GPU, extra set) exactly as if the host-quantized blocks had been uploaded: mul_mat / get_rows accept
 * it; nothing crosses PCIe.  Returns 0, -1 if the precondition does not hold. */
GGML_API int ggml_hip_quantize_resident(const struct ggml_tensor *src, struct ggml_tensor *dst, int64_t *hist);
/* In-kernel timeline of the decode mat-vec launches (ggml_hip_set_option("timeline", 1), eager or graph mode):
 * records of 8 x int64 {entry, loads issued, x staged, barrier passed, first weights landed, exit (100 MHz
 * wall clock ticks), steps of wave 0, workgroup id}; 4 (or "timeline" = n) sampled workgroups per launch, launch order.
 * Returns the number of records copied. */
GGML_API size_t ggml_hip_read_timeline(int64_t *dst, size_t max_records);
/* Test hook (no reference counterpart): the attention of a prompt batch — mul_mat(K, Q), scale, diag_mask_inf, soft_max,
 * mul_mat(V, P), merge heads (crates/models/llama/src/lib.rs:246-307) — on host arrays, through the fused kernel
 * (fused != 0) or the three-launch path of the prompt plan.  q [N][E] f32 (RoPE applied), mem_k [C][Egqa] f16,
 * mem_v [Egqa][C] f16, out [N][E] f32.  0 on success, -1 if the shape is not accepted. */
GGML_API int ggml_hip_debug_prompt_attention(const float *q, const uint16_t *mem_k, const uint16_t *mem_v, float *out, int N, int E,
                                             int Egqa, int H, int n_past, int64_t C, float scale, int fused);
/* Test hook: the fused prompt attention's exponential against expf over ALL f16 arguments: out_fast[i] = f16(exp_le0(x_i)),
 * out_ref[i] = f16(expf(x_i)) for the 65536 f16 bit patterns i (65536 f16 bit patterns each).  The softmax only ever passes
 * x <= 0 or NaN (crates/models/llama/src/lib.rs:272-280: soft_max over masked, scaled scores).  Returns 0. */
GGML_API int ggml_hip_debug_exp_le0(uint16_t *out_fast, uint16_t *out_ref);
/* Test hook: w (quantized 2-D weight with a device copy) times N = 2..8 host rows x [N][K] through k_mmq_cols as the
 * multi-token plan launches it; out [N][M].  0, or -1 when that plan would not run this shape on k_mmq_cols. */
GGML_API int ggml_hip_debug_mul_mat_cols(const struct ggml_tensor *w, const float *x, float *out, int N);
/* Extension (layer split inside one process): slot `slot` enqueues on slot `with_slot`'s stream (with_slot < 0: on its own again).
 * Only for slots of ONE physical GPU whose work never overlaps — the stages of one split session: a wait on another queue's event
 * costs tens of microseconds per stage boundary on this runtime, the same wait inside one queue nothing.  1 = now shared, 0 = not
 * (different GPUs, a slot not initialised).  The host mirror calls it from llm_start_session / llm_session_free. */
GGML_API int ggml_hip_share_stream(int slot, int with_slot);
GGML_API const char *ggml_hip_version(void);

#ifdef __cplusplus
}
#endif

/* ---- ABI layout checks: the numbers of the bindgen layout tests (lib.rs:174-238, 261-446, 458-533,
 * 546-651, 659-705, 711-755, 769-830, 2907-2972) ------------------------------------------------ */
#ifdef __cplusplus
#define GGML_ABI_ASSERT(c, m) static_assert(c, m)
#else
#define GGML_ABI_ASSERT(c, m) _Static_assert(c, m)
#endif
GGML_ABI_ASSERT(sizeof(struct ggml_object) == 32, "ggml_object size (lib.rs:239)");
GGML_ABI_ASSERT(offsetof(struct ggml_object, type) == 24, "ggml_object.type");
GGML_ABI_ASSERT(sizeof(struct ggml_tensor) == 272, "ggml_tensor size (lib.rs:446)");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, ne) == 16, "ggml_tensor.ne");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, nb) == 48, "ggml_tensor.nb");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, op) == 80, "ggml_tensor.op");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, op_params) == 84, "ggml_tensor.op_params");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, is_param) == 116, "ggml_tensor.is_param");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, grad) == 120, "ggml_tensor.grad");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, src) == 128, "ggml_tensor.src");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, perf_runs) == 176, "ggml_tensor.perf_runs");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, data) == 200, "ggml_tensor.data");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, name) == 208, "ggml_tensor.name");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, extra) == 256, "ggml_tensor.extra");
GGML_ABI_ASSERT(offsetof(struct ggml_tensor, padding) == 264, "ggml_tensor.padding");
GGML_ABI_ASSERT(sizeof(struct ggml_cplan) == 16424, "ggml_cplan size");
GGML_ABI_ASSERT(offsetof(struct ggml_cplan, n_tasks) == 20, "ggml_cplan.n_tasks");
GGML_ABI_ASSERT(offsetof(struct ggml_cplan, abort_callback) == 16408, "ggml_cplan.abort_callback");
GGML_ABI_ASSERT(sizeof(struct ggml_cgraph) == 164520, "ggml_cgraph size (lib.rs:651)");
GGML_ABI_ASSERT(offsetof(struct ggml_cgraph, nodes) == 8, "ggml_cgraph.nodes");
GGML_ABI_ASSERT(offsetof(struct ggml_cgraph, leafs) == 65544, "ggml_cgraph.leafs");
GGML_ABI_ASSERT(offsetof(struct ggml_cgraph, visited_hash_table) == 98312, "ggml_cgraph.visited_hash_table");
GGML_ABI_ASSERT(offsetof(struct ggml_cgraph, perf_runs) == 164496, "ggml_cgraph.perf_runs");
GGML_ABI_ASSERT(sizeof(struct ggml_scratch) == 24, "ggml_scratch size");
GGML_ABI_ASSERT(sizeof(struct ggml_init_params) == 24, "ggml_init_params size");
GGML_ABI_ASSERT(sizeof(struct ggml_compute_params) == 32, "ggml_compute_params size");
GGML_ABI_ASSERT(offsetof(struct ggml_compute_params, wsize) == 16, "ggml_compute_params.wsize");
GGML_ABI_ASSERT(sizeof(ggml_type_traits_t) == 40, "ggml_type_traits_t size");

#endif /* GGML_HIP_H */
