// ggml_core.cpp — host side of libggml_hip.so: the ggml C API subset that rustformers/llm's
// `crates/ggml` wrapper links (declarations in include/ggml_hip.h), rebuilt from
// scratch around one idea: this library has NO CPU compute path.  It keeps ggml's arena / tensor /
// graph data model byte-for-byte (so the Rust wrapper's struct accesses stay valid) and hands every
// graph to the MI355X backend (hip_backend.hip).
//
// Reference interfaces replaced here (crates/ggml/sys/src/lib.rs): ggml_init :916, ggml_set_scratch
// :925, tensor constructors, the op builders used by crates/ggml/src/context.rs:276-626,
// ggml_new_graph :1877, ggml_build_forward_expand, ggml_graph_plan :1889, ggml_graph_compute :1895,
// ggml_quantize_q* :2779-2822.  Semantics (result shapes, op_params packing, arena alignment,
// scratch behaviour, leaf/node classification) restate upstream ggml of the 2023-08 window.
#include "ggml_hip.h"

#include <algorithm>
#include <array>
#include <thread>
#include <vector>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

#include "internal.h"

#define GGML_ASSERT(x)                                                                      \
    do {                                                                                    \
        if (!(x)) {                                                                         \
            fprintf(stderr, "GGML_ASSERT: %s:%d: %s\n", __FILE__, __LINE__, #x);            \
            abort();                                                                        \
        }                                                                                   \
    } while (0)

#define GGML_PAD(x, n) (((x) + (n)-1) & ~((size_t)(n)-1))

// ---------------------------------------------------------------------------------------------
// fp16 <-> fp32 (IEEE binary16, RNE).  Branch-light bit formulation (the classic "scale by magic
// power of two" construction), independent of the oracle's table-free loop version.
// ---------------------------------------------------------------------------------------------
static inline float bits_to_f32(uint32_t w) {
    float f;
    memcpy(&f, &w, 4);
    return f;
}
static inline uint32_t f32_to_bits(float f) {
    uint32_t w;
    memcpy(&w, &f, 4);
    return w;
}

float ggml_fp16_to_fp32(ggml_fp16_t h) {
    const uint32_t w = (uint32_t)h << 16;
    const uint32_t sign = w & 0x80000000u;
    const uint32_t two_w = w + w;
    const uint32_t exp_offset = 0xE0u << 23;
    const float exp_scale = bits_to_f32(0x7800000u);  // 2^-112
    const float normalized_value = bits_to_f32((two_w >> 4) + exp_offset) * exp_scale;
    const uint32_t magic_mask = 126u << 23;
    const float magic_bias = 0.5f;
    const float denormalized_value = bits_to_f32((two_w >> 17) | magic_mask) - magic_bias;
    const uint32_t denormalized_cutoff = 1u << 27;
    const uint32_t result =
        sign | (two_w < denormalized_cutoff ? f32_to_bits(denormalized_value) : f32_to_bits(normalized_value));
    return bits_to_f32(result);
}

ggml_fp16_t ggml_fp32_to_fp16(float f) {
    const float scale_to_inf = bits_to_f32(0x77800000u);   // 2^112
    const float scale_to_zero = bits_to_f32(0x08800000u);  // 2^-110
    float base = (fabsf(f) * scale_to_inf) * scale_to_zero;
    const uint32_t w = f32_to_bits(f);
    const uint32_t shl1_w = w + w;
    const uint32_t sign = w & 0x80000000u;
    uint32_t bias = shl1_w & 0xFF000000u;
    if (bias < 0x71000000u) bias = 0x71000000u;
    base = bits_to_f32((bias >> 1) + 0x07800000u) + base;
    const uint32_t bits = f32_to_bits(base);
    const uint32_t exp_bits = (bits >> 13) & 0x00007C00u;
    const uint32_t mantissa_bits = bits & 0x00000FFFu;
    const uint32_t nonsign = exp_bits + mantissa_bits;
    return (ggml_fp16_t)((sign >> 16) | (shl1_w > 0xFF000000u ? 0x7E00u : nonsign));
}

void ggml_fp16_to_fp32_row(const ggml_fp16_t *x, float *y, int n) {
    for (int i = 0; i < n; i++) y[i] = ggml_fp16_to_fp32(x[i]);
}
void ggml_fp32_to_fp16_row(const float *x, ggml_fp16_t *y, int n) {
    for (int i = 0; i < n; i++) y[i] = ggml_fp32_to_fp16(x[i]);
}

// ---------------------------------------------------------------------------------------------
// type table
// ---------------------------------------------------------------------------------------------
struct type_info {
    const char *name;
    int blck;
    size_t size;
    bool quantized;
};
static const type_info TYPE_INFO[GGML_TYPE_COUNT] = {
    /* F32  */ {"f32", 1, 4, false},
    /* F16  */ {"f16", 1, 2, false},
    /* Q4_0 */ {"q4_0", 32, 18, true},
    /* Q4_1 */ {"q4_1", 32, 20, true},
    /* 4    */ {"removed", 0, 0, false},
    /* 5    */ {"removed", 0, 0, false},
    /* Q5_0 */ {"q5_0", 32, 22, true},
    /* Q5_1 */ {"q5_1", 32, 24, true},
    /* Q8_0 */ {"q8_0", 32, 34, true},
    /* Q8_1 */ {"q8_1", 32, 40, true},
    /* K-quants (lib.rs:61-66): Q4_K and Q6_K run on the device (kernels/kquant.h); the others are sized only */
    {"q2_K", 256, 84, true},
    {"q3_K", 256, 110, true},
    {"q4_K", 256, 144, true},
    {"q5_K", 256, 176, true},
    {"q6_K", 256, 210, true},
    {"q8_K", 256, 292, true},
    /* I8   */ {"i8", 1, 1, false},
    /* I16  */ {"i16", 1, 2, false},
    /* I32  */ {"i32", 1, 4, false},
};

static const char *OP_NAME[GGML_OP_COUNT] = {
    "NONE", "DUP", "ADD", "ADD1", "ACC", "SUB", "MUL", "DIV", "SQR", "SQRT", "LOG", "SUM", "SUM_ROWS", "MEAN",
    "ARGMAX", "REPEAT", "REPEAT_BACK", "SILU_BACK", "NORM", "RMS_NORM", "RMS_NORM_BACK", "MUL_MAT", "OUT_PROD",
    "SCALE", "SET", "CPY", "CONT", "RESHAPE", "VIEW", "PERMUTE", "TRANSPOSE", "GET_ROWS", "GET_ROWS_BACK", "DIAG",
    "DIAG_MASK_INF", "DIAG_MASK_ZERO", "SOFT_MAX", "SOFT_MAX_BACK", "ROPE", "ROPE_BACK", "ALIBI", "CLAMP",
    "CONV_1D", "CONV_2D", "POOL_1D", "POOL_2D", "FLASH_ATTN", "FLASH_FF", "FLASH_ATTN_BACK", "WIN_PART",
    "WIN_UNPART", "UNARY", "MAP_UNARY", "MAP_BINARY", "MAP_CUSTOM1", "MAP_CUSTOM2", "MAP_CUSTOM3",
    "CROSS_ENTROPY_LOSS", "CROSS_ENTROPY_LOSS_BACK"};

int ggml_blck_size(enum ggml_type type) { return TYPE_INFO[type].blck; }
size_t ggml_type_size(enum ggml_type type) { return TYPE_INFO[type].size; }
float ggml_type_sizef(enum ggml_type type) { return (float)TYPE_INFO[type].size / (float)TYPE_INFO[type].blck; }
const char *ggml_type_name(enum ggml_type type) { return TYPE_INFO[type].name; }
const char *ggml_op_name(enum ggml_op op) { return OP_NAME[op]; }
bool ggml_is_quantized(enum ggml_type type) { return TYPE_INFO[type].quantized; }

int64_t ggml_nelements(const struct ggml_tensor *t) { return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }
int64_t ggml_nrows(const struct ggml_tensor *t) { return t->ne[1] * t->ne[2] * t->ne[3]; }
size_t ggml_nbytes(const struct ggml_tensor *t) {
    // upstream: max(ne[3]*nb[3], nelements*type_size/blck) — covers views with padded strides
    const size_t a = (size_t)t->ne[3] * t->nb[3];
    const size_t b = (size_t)ggml_nelements(t) * TYPE_INFO[t->type].size / (size_t)TYPE_INFO[t->type].blck;
    return a > b ? a : b;
}
size_t ggml_element_size(const struct ggml_tensor *t) { return TYPE_INFO[t->type].size; }
bool ggml_is_transposed(const struct ggml_tensor *t) { return t->nb[0] > t->nb[1]; }
bool ggml_is_contiguous(const struct ggml_tensor *t) {
    return t->nb[0] == TYPE_INFO[t->type].size &&
           t->nb[1] == (t->nb[0] * (size_t)t->ne[0]) / (size_t)TYPE_INFO[t->type].blck &&
           t->nb[2] == t->nb[1] * (size_t)t->ne[1] && t->nb[3] == t->nb[2] * (size_t)t->ne[2];
}
bool ggml_is_permuted(const struct ggml_tensor *t) {
    return t->nb[0] > t->nb[1] || t->nb[1] > t->nb[2] || t->nb[2] > t->nb[3];
}
size_t ggml_tensor_overhead(void) { return sizeof(struct ggml_object) + sizeof(struct ggml_tensor) + 16; }
void *ggml_get_data(const struct ggml_tensor *t) { return t->data; }
float *ggml_get_data_f32(const struct ggml_tensor *t) {
    GGML_ASSERT(t->type == GGML_TYPE_F32);
    return (float *)t->data;
}
const char *ggml_get_name(const struct ggml_tensor *t) { return t->name; }
struct ggml_tensor *ggml_set_name(struct ggml_tensor *t, const char *name) {
    strncpy(t->name, name, sizeof(t->name));
    t->name[sizeof(t->name) - 1] = '\0';
    return t;
}
static void format_name(struct ggml_tensor *t, const char *fmt, ...) {
    va_list args;
    va_start(args, fmt);
    vsnprintf(t->name, sizeof(t->name), fmt, args);
    va_end(args);
}
// The two name patterns on the per-token path — "<src> (view)" and friends for every view / reshape / permute, "node_<n>" /
// "leaf_<n>" for every unnamed tensor the graph visits — without vsnprintf: a LLaMA-7B graph formats ~1400 names per token, and
// the caller rebuilds the graph for every token (crates/llm-base/src/inference_session.rs:230); same bytes as the format strings
// "%s<suffix>" and "<prefix>%d" give, truncated to the name field like snprintf.
static void name_suffix(struct ggml_tensor *t, const char *src, const char *suffix) {
    char *d = t->name;
    char *const end = t->name + sizeof(t->name) - 1;
    if (src != t->name)
        while (*src && d < end) *d++ = *src++;
    else
        d += strlen(t->name) < (size_t)(end - d) ? strlen(t->name) : (size_t)(end - d);
    while (*suffix && d < end) *d++ = *suffix++;
    *d = 0;
}
static void name_index(struct ggml_tensor *t, const char *prefix, int n) {
    char *d = t->name;
    char *const end = t->name + sizeof(t->name) - 1;
    while (*prefix && d < end) *d++ = *prefix++;
    char digits[12];
    int k = 0;
    unsigned u = n < 0 ? 0u - (unsigned)n : (unsigned)n;
    do {
        digits[k++] = (char)('0' + u % 10);
        u /= 10;
    } while (u);
    if (n < 0 && d < end) *d++ = '-';
    while (k > 0 && d < end) *d++ = digits[--k];
    *d = 0;
}

static inline bool same_shape(const ggml_tensor *a, const ggml_tensor *b) {
    return a->ne[0] == b->ne[0] && a->ne[1] == b->ne[1] && a->ne[2] == b->ne[2] && a->ne[3] == b->ne[3];
}
static inline bool can_repeat(const ggml_tensor *a, const ggml_tensor *b) {  // a repeats into b
    return b->ne[0] % a->ne[0] == 0 && b->ne[1] % a->ne[1] == 0 && b->ne[2] % a->ne[2] == 0 &&
           b->ne[3] % a->ne[3] == 0;
}
static inline bool can_repeat_rows(const ggml_tensor *a, const ggml_tensor *b) {
    return a->ne[0] == b->ne[0] && can_repeat(a, b);
}

// ---------------------------------------------------------------------------------------------
// context: a bump arena of ggml_objects, exactly upstream's scheme (objects are a linked list laid
// out in the buffer; each holds a tensor header and, unless a scratch buffer is active or no_alloc
// is set, the tensor data right behind it).
// ---------------------------------------------------------------------------------------------
struct ggml_context {
    size_t mem_size;
    void *mem_buffer;
    bool mem_buffer_owned;
    bool no_alloc;
    bool no_alloc_save;
    int n_objects;
    struct ggml_object *objects_begin;
    struct ggml_object *objects_end;
    struct ggml_scratch scratch;
    struct ggml_scratch scratch_save;
};

struct ggml_context *ggml_init(struct ggml_init_params params) {
    ggml_context *ctx = (ggml_context *)calloc(1, sizeof(ggml_context));
    GGML_ASSERT(ctx != nullptr);
    const size_t mem_size = params.mem_buffer ? params.mem_size : GGML_PAD(params.mem_size, GGML_MEM_ALIGN);
    ctx->mem_size = mem_size;
    if (params.mem_buffer) {
        ctx->mem_buffer = params.mem_buffer;
    } else {
        void *p = nullptr;
        if (posix_memalign(&p, GGML_MEM_ALIGN, mem_size ? mem_size : GGML_MEM_ALIGN) != 0) p = nullptr;
        if (!p) {
            fprintf(stderr, "ggml_init: failed to allocate %zu bytes\n", mem_size);
            free(ctx);
            return nullptr;
        }
        ctx->mem_buffer = p;
    }
    ctx->mem_buffer_owned = params.mem_buffer == nullptr;
    ctx->no_alloc = params.no_alloc;
    ctx->no_alloc_save = params.no_alloc;
    GGML_ASSERT(((uintptr_t)ctx->mem_buffer) % GGML_MEM_ALIGN == 0);
    ggml_hip_internal_register_arena(ctx->mem_buffer, ctx->mem_size, 0);
    return ctx;
}

void ggml_free(struct ggml_context *ctx) {
    if (!ctx) return;
    ggml_hip_internal_unregister_arena(ctx->mem_buffer);
    if (ctx->mem_buffer_owned) free(ctx->mem_buffer);
    free(ctx);
}

size_t ggml_used_mem(const struct ggml_context *ctx) {
    return ctx->objects_end == nullptr ? 0 : ctx->objects_end->offs + ctx->objects_end->size;
}
size_t ggml_set_scratch(struct ggml_context *ctx, struct ggml_scratch scratch) {
    const size_t result = ctx->scratch.data ? ctx->scratch.offs : 0;
    ctx->scratch = scratch;
    if (scratch.data) ggml_hip_internal_register_arena(scratch.data, scratch.size, 1);  // idempotent
    return result;
}
bool ggml_get_no_alloc(struct ggml_context *ctx) { return ctx->no_alloc; }
void ggml_set_no_alloc(struct ggml_context *ctx, bool no_alloc) { ctx->no_alloc = no_alloc; }
void *ggml_get_mem_buffer(const struct ggml_context *ctx) { return ctx->mem_buffer; }
size_t ggml_get_mem_size(const struct ggml_context *ctx) { return ctx->mem_size; }
size_t ggml_get_max_tensor_size(const struct ggml_context *ctx) {
    size_t max_size = 0;
    for (ggml_object *obj = ctx->objects_begin; obj; obj = obj->next) {
        if (obj->type != GGML_OBJECT_TENSOR) continue;
        const ggml_tensor *t = (const ggml_tensor *)((char *)ctx->mem_buffer + obj->offs);
        max_size = std::max(max_size, ggml_nbytes(t));
    }
    return max_size;
}
void ggml_print_objects(const struct ggml_context *ctx) {
    fprintf(stderr, "%s: objects in context %p:\n", __func__, (const void *)ctx);
    for (ggml_object *obj = ctx->objects_begin; obj; obj = obj->next)
        fprintf(stderr, " - ggml_object: type = %d, offset = %zu, size = %zu, next = %p\n", (int)obj->type, obj->offs,
                obj->size, (void *)obj->next);
    fprintf(stderr, "%s: --- end ---\n", __func__);
}

static ggml_object *new_object(ggml_context *ctx, ggml_object_type type, size_t size) {
    ggml_object *obj_cur = ctx->objects_end;
    const size_t cur_offs = obj_cur == nullptr ? 0 : obj_cur->offs;
    const size_t cur_size = obj_cur == nullptr ? 0 : obj_cur->size;
    const size_t cur_end = cur_offs + cur_size;
    const size_t size_needed = GGML_PAD(size, GGML_MEM_ALIGN);
    char *const mem_buffer = (char *)ctx->mem_buffer;
    ggml_object *const obj_new = (ggml_object *)(mem_buffer + cur_end);
    if (cur_end + size_needed + sizeof(ggml_object) > ctx->mem_size) {
        fprintf(stderr, "%s: not enough space in the context's memory pool (needed %zu, available %zu)\n", __func__,
                cur_end + size_needed + sizeof(ggml_object), ctx->mem_size);
        abort();
    }
    obj_new->offs = cur_end + sizeof(ggml_object);
    obj_new->size = size_needed;
    obj_new->next = nullptr;
    obj_new->type = type;
    GGML_ASSERT(((uintptr_t)(mem_buffer + obj_new->offs)) % GGML_MEM_ALIGN == 0);
    if (obj_cur != nullptr)
        obj_cur->next = obj_new;
    else
        ctx->objects_begin = obj_new;
    ctx->objects_end = obj_new;
    ctx->n_objects++;
    return obj_new;
}

static ggml_tensor *new_tensor_impl(ggml_context *ctx, ggml_type type, int n_dims, const int64_t *ne, void *data) {
    GGML_ASSERT(n_dims >= 1 && n_dims <= GGML_MAX_DIMS);
    GGML_ASSERT(type < GGML_TYPE_COUNT && TYPE_INFO[type].blck > 0);
    size_t data_size = 0;
    if (data == nullptr && !ctx->no_alloc) {
        data_size = TYPE_INFO[type].size * (size_t)(ne[0] / TYPE_INFO[type].blck);
        for (int i = 1; i < n_dims; i++) data_size *= (size_t)ne[i];
    }
    if (ctx->scratch.data != nullptr && data == nullptr) {
        // allocate tensor data in the scratch buffer
        if (ctx->scratch.offs + data_size > ctx->scratch.size) {
            fprintf(stderr, "%s: not enough space in the scratch memory pool (needed %zu, available %zu)\n", __func__,
                    ctx->scratch.offs + data_size, ctx->scratch.size);
            abort();
        }
        data = (char *)ctx->scratch.data + ctx->scratch.offs;
        ctx->scratch.offs += GGML_PAD(data_size, GGML_MEM_ALIGN);
        data_size = 0;
    }
    ggml_object *const obj = new_object(ctx, GGML_OBJECT_TENSOR, sizeof(ggml_tensor) + data_size);
    ggml_tensor *const result = (ggml_tensor *)((char *)ctx->mem_buffer + obj->offs);
    memset(result, 0, sizeof(ggml_tensor));
    result->type = type;
    result->backend = GGML_BACKEND_CPU;
    result->n_dims = n_dims;
    result->ne[0] = result->ne[1] = result->ne[2] = result->ne[3] = 1;
    result->op = GGML_OP_NONE;
    result->data = (data == nullptr && !ctx->no_alloc) ? (void *)(result + 1) : data;
    for (int i = 0; i < n_dims; i++) result->ne[i] = ne[i];
    result->nb[0] = TYPE_INFO[type].size;
    result->nb[1] = result->nb[0] * (size_t)(result->ne[0] / TYPE_INFO[type].blck);
    for (int i = 2; i < GGML_MAX_DIMS; i++) result->nb[i] = result->nb[i - 1] * (size_t)result->ne[i - 1];
    return result;
}

struct ggml_tensor *ggml_new_tensor(struct ggml_context *ctx, enum ggml_type type, int n_dims, const int64_t *ne) {
    return new_tensor_impl(ctx, type, n_dims, ne, nullptr);
}
struct ggml_tensor *ggml_new_tensor_1d(struct ggml_context *ctx, enum ggml_type type, int64_t ne0) {
    return ggml_new_tensor(ctx, type, 1, &ne0);
}
struct ggml_tensor *ggml_new_tensor_2d(struct ggml_context *ctx, enum ggml_type type, int64_t ne0, int64_t ne1) {
    const int64_t ne[2] = {ne0, ne1};
    return ggml_new_tensor(ctx, type, 2, ne);
}
struct ggml_tensor *ggml_new_tensor_3d(struct ggml_context *ctx, enum ggml_type type, int64_t ne0, int64_t ne1,
                                       int64_t ne2) {
    const int64_t ne[3] = {ne0, ne1, ne2};
    return ggml_new_tensor(ctx, type, 3, ne);
}
struct ggml_tensor *ggml_new_tensor_4d(struct ggml_context *ctx, enum ggml_type type, int64_t ne0, int64_t ne1,
                                       int64_t ne2, int64_t ne3) {
    const int64_t ne[4] = {ne0, ne1, ne2, ne3};
    return ggml_new_tensor(ctx, type, 4, ne);
}
static void scratch_save(ggml_context *ctx) {
    // constants created with ggml_new_i32/f32 live in the context proper, not in the scratch
    ctx->no_alloc_save = ctx->no_alloc;
    ctx->no_alloc = false;
    ctx->scratch_save = ctx->scratch;
    ctx->scratch.data = nullptr;
}
static void scratch_load(ggml_context *ctx) {
    ctx->no_alloc = ctx->no_alloc_save;
    ctx->scratch = ctx->scratch_save;
}
struct ggml_tensor *ggml_new_i32(struct ggml_context *ctx, int32_t value) {
    scratch_save(ctx);
    ggml_tensor *result = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, 1);
    scratch_load(ctx);
    *(int32_t *)result->data = value;
    return result;
}
struct ggml_tensor *ggml_new_f32(struct ggml_context *ctx, float value) {
    scratch_save(ctx);
    ggml_tensor *result = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, 1);
    scratch_load(ctx);
    *(float *)result->data = value;
    return result;
}
struct ggml_tensor *ggml_dup_tensor(struct ggml_context *ctx, const struct ggml_tensor *src) {
    return new_tensor_impl(ctx, src->type, src->n_dims, src->ne, nullptr);
}
struct ggml_tensor *ggml_view_tensor(struct ggml_context *ctx, const struct ggml_tensor *src) {
    ggml_tensor *result = new_tensor_impl(ctx, src->type, src->n_dims, src->ne, src->data);
    name_suffix(result, src->name, " (view)");
    for (int i = 0; i < GGML_MAX_DIMS; i++) result->nb[i] = src->nb[i];
    return result;
}

static inline void set_op_params(ggml_tensor *t, const void *params, size_t size) {
    GGML_ASSERT(size <= GGML_MAX_OP_PARAMS);
    memcpy(t->op_params, params, size);
}

// ---------------------------------------------------------------------------------------------
// op builders
// ---------------------------------------------------------------------------------------------
static ggml_tensor *unary_like(ggml_context *ctx, ggml_tensor *a, ggml_op op, bool inplace) {
    ggml_tensor *result = inplace ? ggml_view_tensor(ctx, a) : ggml_dup_tensor(ctx, a);
    result->op = op;
    result->src[0] = a;
    return result;
}

struct ggml_tensor *ggml_dup(struct ggml_context *ctx, struct ggml_tensor *a) {
    return unary_like(ctx, a, GGML_OP_DUP, false);
}
static ggml_tensor *add_impl(ggml_context *ctx, ggml_tensor *a, ggml_tensor *b, bool inplace) {
    GGML_ASSERT(can_repeat_rows(b, a));
    ggml_tensor *result = unary_like(ctx, a, GGML_OP_ADD, inplace);
    result->src[1] = b;
    return result;
}
struct ggml_tensor *ggml_add(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b) {
    return add_impl(ctx, a, b, false);
}
struct ggml_tensor *ggml_add_inplace(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b) {
    return add_impl(ctx, a, b, true);
}
struct ggml_tensor *ggml_mul(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b) {
    GGML_ASSERT(can_repeat_rows(b, a));
    ggml_tensor *result = unary_like(ctx, a, GGML_OP_MUL, false);
    result->src[1] = b;
    return result;
}
struct ggml_tensor *ggml_repeat(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b) {
    GGML_ASSERT(can_repeat(a, b));
    ggml_tensor *result = ggml_new_tensor(ctx, a->type, b->n_dims, b->ne);
    result->op = GGML_OP_REPEAT;
    result->src[0] = a;
    result->src[1] = b;
    return result;
}
static ggml_tensor *unary_op(ggml_context *ctx, ggml_tensor *a, ggml_unary_op uop) {
    ggml_tensor *result = unary_like(ctx, a, GGML_OP_UNARY, false);
    const int32_t p = (int32_t)uop;
    set_op_params(result, &p, sizeof(p));
    return result;
}
struct ggml_tensor *ggml_silu(struct ggml_context *ctx, struct ggml_tensor *a) {
    return unary_op(ctx, a, GGML_UNARY_OP_SILU);
}
struct ggml_tensor *ggml_gelu(struct ggml_context *ctx, struct ggml_tensor *a) {
    return unary_op(ctx, a, GGML_UNARY_OP_GELU);
}
struct ggml_tensor *ggml_norm(struct ggml_context *ctx, struct ggml_tensor *a) {
    return unary_like(ctx, a, GGML_OP_NORM, false);
}
struct ggml_tensor *ggml_rms_norm(struct ggml_context *ctx, struct ggml_tensor *a, float eps) {
    ggml_tensor *result = unary_like(ctx, a, GGML_OP_RMS_NORM, false);
    set_op_params(result, &eps, sizeof(eps));
    return result;
}
struct ggml_tensor *ggml_mul_mat(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b) {
    GGML_ASSERT(a->ne[0] == b->ne[0] && b->ne[2] % a->ne[2] == 0 && b->ne[3] % a->ne[3] == 0);  // can_mul_mat
    GGML_ASSERT(!ggml_is_transposed(a));
    const int64_t ne[4] = {a->ne[1], b->ne[1], b->ne[2], b->ne[3]};
    ggml_tensor *result = ggml_new_tensor(ctx, GGML_TYPE_F32, std::max(a->n_dims, b->n_dims), ne);
    result->op = GGML_OP_MUL_MAT;
    result->src[0] = a;
    result->src[1] = b;
    return result;
}
static ggml_tensor *scale_impl(ggml_context *ctx, ggml_tensor *a, ggml_tensor *b, bool inplace) {
    GGML_ASSERT(ggml_nelements(b) == 1);  // is_scalar
    ggml_tensor *result = unary_like(ctx, a, GGML_OP_SCALE, inplace);
    result->src[1] = b;
    return result;
}
struct ggml_tensor *ggml_scale(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b) {
    return scale_impl(ctx, a, b, false);
}
struct ggml_tensor *ggml_scale_inplace(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b) {
    return scale_impl(ctx, a, b, true);
}
struct ggml_tensor *ggml_cpy(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b) {
    GGML_ASSERT(ggml_nelements(a) == ggml_nelements(b));
    ggml_tensor *result = ggml_view_tensor(ctx, b);  // "make a view of the destination"
    if (strlen(b->name) > 0)
        format_name(result, "%s (copy of %s)", b->name, a->name);
    else
        name_suffix(result, a->name, " (copy)");
    result->op = GGML_OP_CPY;
    result->src[0] = a;
    result->src[1] = b;
    return result;
}
struct ggml_tensor *ggml_cont(struct ggml_context *ctx, struct ggml_tensor *a) {
    ggml_tensor *result = ggml_dup_tensor(ctx, a);
    name_suffix(result, a->name, " (cont)");
    result->op = GGML_OP_CONT;
    result->src[0] = a;
    return result;
}
static ggml_tensor *reshape_nd(ggml_context *ctx, ggml_tensor *a, int n_dims, const int64_t *ne) {
    GGML_ASSERT(ggml_is_contiguous(a));
    int64_t n = 1;
    for (int i = 0; i < n_dims; i++) n *= ne[i];
    GGML_ASSERT(ggml_nelements(a) == n);
    ggml_tensor *result = new_tensor_impl(ctx, a->type, n_dims, ne, a->data);
    name_suffix(result, a->name, " (reshaped)");
    result->op = GGML_OP_RESHAPE;
    result->src[0] = a;
    return result;
}
struct ggml_tensor *ggml_reshape(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b) {
    GGML_ASSERT(ggml_is_contiguous(b));
    return reshape_nd(ctx, a, b->n_dims, b->ne);
}
struct ggml_tensor *ggml_reshape_1d(struct ggml_context *ctx, struct ggml_tensor *a, int64_t ne0) {
    return reshape_nd(ctx, a, 1, &ne0);
}
struct ggml_tensor *ggml_reshape_2d(struct ggml_context *ctx, struct ggml_tensor *a, int64_t ne0, int64_t ne1) {
    const int64_t ne[2] = {ne0, ne1};
    return reshape_nd(ctx, a, 2, ne);
}
struct ggml_tensor *ggml_reshape_3d(struct ggml_context *ctx, struct ggml_tensor *a, int64_t ne0, int64_t ne1,
                                    int64_t ne2) {
    const int64_t ne[3] = {ne0, ne1, ne2};
    return reshape_nd(ctx, a, 3, ne);
}
static ggml_tensor *view_nd(ggml_context *ctx, ggml_tensor *a, int n_dims, const int64_t *ne, size_t offset) {
    ggml_tensor *result = new_tensor_impl(ctx, a->type, n_dims, ne, (char *)a->data + offset);
    name_suffix(result, a->name, " (view)");
    set_op_params(result, &offset, sizeof(offset));
    result->op = GGML_OP_VIEW;
    result->src[0] = a;
    return result;
}
struct ggml_tensor *ggml_view_1d(struct ggml_context *ctx, struct ggml_tensor *a, int64_t ne0, size_t offset) {
    return view_nd(ctx, a, 1, &ne0, offset);
}
struct ggml_tensor *ggml_view_2d(struct ggml_context *ctx, struct ggml_tensor *a, int64_t ne0, int64_t ne1,
                                 size_t nb1, size_t offset) {
    const int64_t ne[2] = {ne0, ne1};
    ggml_tensor *result = view_nd(ctx, a, 2, ne, offset);
    result->nb[1] = nb1;
    result->nb[2] = result->nb[1] * (size_t)ne1;
    result->nb[3] = result->nb[2];
    return result;
}
struct ggml_tensor *ggml_view_3d(struct ggml_context *ctx, struct ggml_tensor *a, int64_t ne0, int64_t ne1,
                                 int64_t ne2, size_t nb1, size_t nb2, size_t offset) {
    const int64_t ne[3] = {ne0, ne1, ne2};
    ggml_tensor *result = view_nd(ctx, a, 3, ne, offset);
    result->nb[1] = nb1;
    result->nb[2] = nb2;
    result->nb[3] = result->nb[2] * (size_t)ne2;
    return result;
}
struct ggml_tensor *ggml_permute(struct ggml_context *ctx, struct ggml_tensor *a, int axis0, int axis1, int axis2,
                                 int axis3) {
    GGML_ASSERT(axis0 >= 0 && axis0 < 4 && axis1 >= 0 && axis1 < 4 && axis2 >= 0 && axis2 < 4 && axis3 >= 0 &&
                axis3 < 4);
    GGML_ASSERT(axis0 != axis1 && axis0 != axis2 && axis0 != axis3 && axis1 != axis2 && axis1 != axis3 &&
                axis2 != axis3);
    ggml_tensor *result = ggml_view_tensor(ctx, a);
    name_suffix(result, a->name, " (permuted)");
    int64_t ne[4];
    size_t nb[4];
    ne[axis0] = a->ne[0];
    ne[axis1] = a->ne[1];
    ne[axis2] = a->ne[2];
    ne[axis3] = a->ne[3];
    nb[axis0] = a->nb[0];
    nb[axis1] = a->nb[1];
    nb[axis2] = a->nb[2];
    nb[axis3] = a->nb[3];
    for (int i = 0; i < 4; i++) {
        result->ne[i] = ne[i];
        result->nb[i] = nb[i];
    }
    result->op = GGML_OP_PERMUTE;
    result->src[0] = a;
    const int32_t params[4] = {axis0, axis1, axis2, axis3};
    set_op_params(result, params, sizeof(params));
    return result;
}
struct ggml_tensor *ggml_transpose(struct ggml_context *ctx, struct ggml_tensor *a) {
    ggml_tensor *result = ggml_view_tensor(ctx, a);
    name_suffix(result, a->name, " (transposed)");
    result->ne[0] = a->ne[1];
    result->ne[1] = a->ne[0];
    result->nb[0] = a->nb[1];
    result->nb[1] = a->nb[0];
    result->op = GGML_OP_TRANSPOSE;
    result->src[0] = a;
    return result;
}
struct ggml_tensor *ggml_get_rows(struct ggml_context *ctx, struct ggml_tensor *a, struct ggml_tensor *b) {
    GGML_ASSERT(a->ne[2] == 1 && a->ne[3] == 1 && b->ne[1] == 1 && b->ne[2] == 1 && b->ne[3] == 1);  // matrix, vector
    GGML_ASSERT(b->type == GGML_TYPE_I32);
    ggml_tensor *result = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, a->ne[0], b->ne[0]);
    result->op = GGML_OP_GET_ROWS;
    result->src[0] = a;
    result->src[1] = b;
    return result;
}
static ggml_tensor *diag_mask_inf_impl(ggml_context *ctx, ggml_tensor *a, int n_past, bool inplace) {
    ggml_tensor *result = unary_like(ctx, a, GGML_OP_DIAG_MASK_INF, inplace);
    const int32_t params[2] = {n_past, inplace ? 1 : 0};
    set_op_params(result, params, sizeof(params));
    return result;
}
struct ggml_tensor *ggml_diag_mask_inf(struct ggml_context *ctx, struct ggml_tensor *a, int n_past) {
    return diag_mask_inf_impl(ctx, a, n_past, false);
}
struct ggml_tensor *ggml_diag_mask_inf_inplace(struct ggml_context *ctx, struct ggml_tensor *a, int n_past) {
    return diag_mask_inf_impl(ctx, a, n_past, true);
}
struct ggml_tensor *ggml_soft_max(struct ggml_context *ctx, struct ggml_tensor *a) {
    return unary_like(ctx, a, GGML_OP_SOFT_MAX, false);
}
struct ggml_tensor *ggml_soft_max_inplace(struct ggml_context *ctx, struct ggml_tensor *a) {
    return unary_like(ctx, a, GGML_OP_SOFT_MAX, true);
}
static ggml_tensor *rope_impl(ggml_context *ctx, ggml_tensor *a, int n_past, int n_dims, int mode, int n_ctx,
                              float freq_base, float freq_scale, bool inplace) {
    GGML_ASSERT(n_past >= 0);
    ggml_tensor *result = unary_like(ctx, a, GGML_OP_ROPE, inplace);
    int32_t params[6] = {n_past, n_dims, mode, n_ctx, 0, 0};
    memcpy(params + 4, &freq_base, sizeof(float));
    memcpy(params + 5, &freq_scale, sizeof(float));
    set_op_params(result, params, sizeof(params));
    return result;
}
struct ggml_tensor *ggml_rope(struct ggml_context *ctx, struct ggml_tensor *a, int n_past, int n_dims, int mode,
                              int n_ctx) {
    return rope_impl(ctx, a, n_past, n_dims, mode, n_ctx, 10000.0f, 1.0f, false);
}
struct ggml_tensor *ggml_rope_inplace(struct ggml_context *ctx, struct ggml_tensor *a, int n_past, int n_dims,
                                      int mode, int n_ctx) {
    return rope_impl(ctx, a, n_past, n_dims, mode, n_ctx, 10000.0f, 1.0f, true);
}
struct ggml_tensor *ggml_rope_custom_inplace(struct ggml_context *ctx, struct ggml_tensor *a, int n_past, int n_dims,
                                             int mode, int n_ctx, float freq_base, float freq_scale) {
    return rope_impl(ctx, a, n_past, n_dims, mode, n_ctx, freq_base, freq_scale, true);
}

static void out_of_path(const char *what) {
    fprintf(stderr,
            "libggml_hip: %s is outside the accelerated LLaMA/GPT-2 path of this library (SURVEY.md §2 rows 12, "
            "§8a) and there is no CPU compute fallback\n",
            what);
    abort();
}
struct ggml_tensor *ggml_alibi(struct ggml_context *, struct ggml_tensor *, int, int, float) {
    out_of_path("ggml_alibi");
    return nullptr;
}
struct ggml_tensor *ggml_flash_attn(struct ggml_context *, struct ggml_tensor *, struct ggml_tensor *,
                                    struct ggml_tensor *, bool) {
    out_of_path("ggml_flash_attn");
    return nullptr;
}
struct ggml_tensor *ggml_map_unary_f32(struct ggml_context *, struct ggml_tensor *, ggml_unary_op_f32_t) {
    out_of_path("ggml_map_unary_f32 (host callbacks cannot run on the device)");
    return nullptr;
}
struct ggml_tensor *ggml_map_binary_f32(struct ggml_context *, struct ggml_tensor *, struct ggml_tensor *,
                                        ggml_binary_op_f32_t) {
    out_of_path("ggml_map_binary_f32 (host callbacks cannot run on the device)");
    return nullptr;
}

// ---------------------------------------------------------------------------------------------
// graph
// ---------------------------------------------------------------------------------------------
static bool hash_insert(void *table[], void *p) {  // true if already present
    const size_t h = (size_t)((uintptr_t)p % GGML_GRAPH_HASHTABLE_SIZE);
    size_t i = h;
    while (table[i] != nullptr) {
        if (table[i] == p) return true;
        i = (i + 1) % GGML_GRAPH_HASHTABLE_SIZE;
        GGML_ASSERT(i != h && "visited hash table is full");
    }
    table[i] = p;
    return false;
}

static void visit_parents(ggml_cgraph *cgraph, ggml_tensor *node) {
    if (hash_insert(cgraph->visited_hash_table, node)) return;
    for (int i = 0; i < GGML_MAX_SRC; ++i)
        if (node->src[i]) visit_parents(cgraph, node->src[i]);
    if (node->op == GGML_OP_NONE && node->grad == nullptr) {
        GGML_ASSERT(cgraph->n_leafs < GGML_MAX_NODES);
        if (node->name[0] == 0) name_index(node, "leaf_", cgraph->n_leafs);
        cgraph->leafs[cgraph->n_leafs++] = node;
    } else {
        GGML_ASSERT(cgraph->n_nodes < GGML_MAX_NODES);
        if (node->name[0] == 0) name_index(node, "node_", cgraph->n_nodes);
        cgraph->nodes[cgraph->n_nodes] = node;
        cgraph->grads[cgraph->n_nodes] = node->grad;
        cgraph->n_nodes++;
    }
}

void ggml_build_forward_expand(struct ggml_cgraph *cgraph, struct ggml_tensor *tensor) {
    const int n0 = cgraph->n_nodes;
    visit_parents(cgraph, tensor);
    const int n_new = cgraph->n_nodes - n0;
    if (n_new > 0) GGML_ASSERT(cgraph->nodes[cgraph->n_nodes - 1] == tensor);  // the last added node is `tensor`
}
struct ggml_cgraph ggml_build_forward(struct ggml_tensor *tensor) {
    ggml_cgraph result;
    memset(&result, 0, sizeof(result));
    ggml_build_forward_expand(&result, tensor);
    return result;
}
size_t ggml_graph_overhead(void) { return sizeof(ggml_object) + GGML_PAD(sizeof(ggml_cgraph), GGML_MEM_ALIGN); }
struct ggml_cgraph *ggml_new_graph(struct ggml_context *ctx) {
    ggml_object *obj = new_object(ctx, GGML_OBJECT_GRAPH, sizeof(ggml_cgraph));
    ggml_cgraph *cgraph = (ggml_cgraph *)((char *)ctx->mem_buffer + obj->offs);
    memset(cgraph, 0, sizeof(ggml_cgraph));
    return cgraph;
}
void ggml_graph_reset(struct ggml_cgraph *) {}

struct ggml_cplan ggml_graph_plan(struct ggml_cgraph *cgraph, int n_threads) {
    // The device executor keeps its own workspace (activation re-quantization buffers etc.), so the host
    // work buffer the caller allocates from this number (crates/ggml/src/lib.rs:354-367) is empty.
    ggml_cplan cplan;
    memset(&cplan, 0, sizeof(cplan));
    cplan.n_threads = n_threads > 0 ? n_threads : GGML_DEFAULT_N_THREADS;
    for (int i = 0; i < cgraph->n_nodes; i++) cplan.n_tasks[i] = 1;
    cplan.work_size = 0;
    cplan.work_data = nullptr;
    return cplan;
}

static int64_t wall_us() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000000 + ts.tv_nsec / 1000;
}
int ggml_graph_compute(struct ggml_cgraph *cgraph, struct ggml_cplan *cplan) {
    (void)cplan;
    // ggml's tracing hook (crates/ggml/sys/src/lib.rs:253-255, 542-544): upstream bumps perf_runs of the graph and of every
    // node on each compute; the time fields only move in a GGML_PERF build, which the reference's build.rs does not enable.
    // Here the graph's fields carry the wall time of the call (enqueue + device wait: what a caller timing the FFI call
    // would see); a node's time stays 0 — under the fused plans a node has no launch of its own to time.
    const int64_t t0 = wall_us();
    const clock_t c0 = clock();
    ggml_hip_internal_graph_compute(cgraph);
    cgraph->perf_runs++;
    cgraph->perf_cycles += (int64_t)(clock() - c0);
    cgraph->perf_time_us += wall_us() - t0;
    for (int i = 0; i < cgraph->n_nodes; i++) cgraph->nodes[i]->perf_runs++;
    return GGML_EXIT_SUCCESS;
}

int ggml_cpu_has_blas(void) { return 1; }     // "a BLAS-class accelerator is present" (upstream: cublas => 1)
int ggml_cpu_has_gpublas(void) { return 1; }

// ---------------------------------------------------------------------------------------------
// offline quantizer (host): ggml_quantize_q* — crates/ggml/src/lib.rs:419-483 calls these from
// crates/llm-base/src/quantize.rs:363-379.  Restates upstream quantize_row_q*_reference; written
// block-at-a-time with a shared min/max scan (different structure from the oracle's copy on purpose).
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int QK = 32;

struct scan_t {
    float vmin, vmax, amax_signed;  // amax_signed: value with the largest magnitude (first wins on ties)
};
inline scan_t scan_block(const float *x) {
    scan_t s{FLT_MAX, -FLT_MAX, 0.0f};
    float amax = 0.0f;
    for (int j = 0; j < QK; j++) {
        const float v = x[j];
        if (v < s.vmin) s.vmin = v;
        if (v > s.vmax) s.vmax = v;
        if (amax < fabsf(v)) {
            amax = fabsf(v);
            s.amax_signed = v;
        }
    }
    return s;
}
inline void put_f16(uint8_t *p, float f) {
    const ggml_fp16_t h = ggml_fp32_to_fp16(f);
    memcpy(p, &h, 2);
}
inline int imin(int a, int b) { return a < b ? a : b; }

// upstream's 16-bin histogram for the 5-bit types: walks j = 0,2,..,30 pairing qs[j/2] with the high bits at
// positions j and j+16 ("cast to 16 bins") — kept as is so `llm quantize` prints what the reference prints
void hist_q5(uint32_t qh, const uint8_t *qs, int64_t *hist) {
    for (int j = 0; j < QK; j += 2) {
        const uint8_t vh0 = (uint8_t)(((qh & (1u << (j + 0))) >> (j + 0)) << 4);
        const uint8_t vh1 = (uint8_t)((qh & (1u << ((j + 16) & 31))) >> ((j + 12) & 31));  // x86 shift-count wrap of upstream
        hist[((qs[j / 2] & 0x0F) | vh0) / 2]++;
        hist[((qs[j / 2] >> 4) | vh1) / 2]++;
    }
}

void quant_block(ggml_type type, const float *x, uint8_t *out, int64_t *hist) {
    const scan_t s = scan_block(x);
    switch (type) {
        case GGML_TYPE_Q4_0: {
            const float d = s.amax_signed / -8;
            const float id = d ? 1.0f / d : 0.0f;
            put_f16(out, d);
            for (int j = 0; j < QK / 2; j++) {
                const uint8_t a = (uint8_t)imin(15, (int8_t)(x[j] * id + 8.5f));
                const uint8_t b = (uint8_t)imin(15, (int8_t)(x[j + QK / 2] * id + 8.5f));
                out[2 + j] = a | (b << 4);
                if (hist) { hist[a]++; hist[b]++; }
            }
        } break;
        case GGML_TYPE_Q4_1: {
            const float d = (s.vmax - s.vmin) / 15;
            const float id = d ? 1.0f / d : 0.0f;
            put_f16(out, d);
            put_f16(out + 2, s.vmin);
            for (int j = 0; j < QK / 2; j++) {
                const uint8_t a = (uint8_t)imin(15, (int8_t)((x[j] - s.vmin) * id + 0.5f));
                const uint8_t b = (uint8_t)imin(15, (int8_t)((x[j + QK / 2] - s.vmin) * id + 0.5f));
                out[4 + j] = a | (b << 4);
                if (hist) { hist[a]++; hist[b]++; }
            }
        } break;
        case GGML_TYPE_Q5_0: {
            const float d = s.amax_signed / -16;
            const float id = d ? 1.0f / d : 0.0f;
            put_f16(out, d);
            uint32_t qh = 0;
            for (int j = 0; j < QK / 2; j++) {
                const uint8_t a = (uint8_t)imin(31, (int8_t)(x[j] * id + 16.5f));
                const uint8_t b = (uint8_t)imin(31, (int8_t)(x[j + QK / 2] * id + 16.5f));
                out[6 + j] = (a & 0x0F) | ((b & 0x0F) << 4);
                qh |= (uint32_t)((a & 0x10u) >> 4) << j;
                qh |= (uint32_t)((b & 0x10u) >> 4) << (j + QK / 2);
            }
            memcpy(out + 2, &qh, 4);
            if (hist) hist_q5(qh, out + 6, hist);
        } break;
        case GGML_TYPE_Q5_1: {
            const float d = (s.vmax - s.vmin) / 31;
            const float id = d ? 1.0f / d : 0.0f;
            put_f16(out, d);
            put_f16(out + 2, s.vmin);
            uint32_t qh = 0;
            for (int j = 0; j < QK / 2; j++) {
                const uint8_t a = (uint8_t)((x[j] - s.vmin) * id + 0.5f);
                const uint8_t b = (uint8_t)((x[j + QK / 2] - s.vmin) * id + 0.5f);
                out[8 + j] = (a & 0x0F) | ((b & 0x0F) << 4);
                qh |= (uint32_t)((a & 0x10u) >> 4) << j;
                qh |= (uint32_t)((b & 0x10u) >> 4) << (j + QK / 2);
            }
            memcpy(out + 4, &qh, 4);
            if (hist) hist_q5(qh, out + 8, hist);
        } break;
        case GGML_TYPE_Q8_0: {
            const float amax = fabsf(s.amax_signed);
            const float d = amax / 127;
            const float id = d ? 1.0f / d : 0.0f;
            put_f16(out, d);
            for (int j = 0; j < QK; j++) {
                const int8_t q = (int8_t)roundf(x[j] * id);
                memcpy(out + 2 + j, &q, 1);
                if (hist) hist[q / 16 + 8]++;
            }
        } break;
        default:
            fprintf(stderr, "ggml_quantize: unsupported type %d\n", (int)type);
            abort();
    }
}

size_t quantize_rows(ggml_type type, const float *src, void *dst, int n, int k, int64_t *hist) {
    GGML_ASSERT(k % QK == 0 && n % k == 0);
    const size_t bs = TYPE_INFO[type].size;
    uint8_t *out = (uint8_t *)dst;
    const int nblocks = n / QK;
    // SURVEY 8f N2 ("fast CPU"): the blocks are independent, so a large tensor is cut into contiguous ranges, one per
    // thread, each with its own histogram; bytes and histogram are identical to the serial loop (integer sums).
    const unsigned hw = std::thread::hardware_concurrency();
    const int nthr = nblocks >= (1 << 15) ? (int)std::min<unsigned>(hw ? hw : 1, 16) : 1;
    if (nthr <= 1) {
        for (int b = 0; b < nblocks; b++) quant_block(type, src + (size_t)b * QK, out + (size_t)b * bs, hist);
        return (size_t)nblocks * bs;
    }
    std::vector<std::array<int64_t, 16>> hs((size_t)nthr);
    std::vector<std::thread> th;
    const int per = (nblocks + nthr - 1) / nthr;
    for (int t = 0; t < nthr; t++) {
        hs[(size_t)t].fill(0);
        th.emplace_back([=, &hs] {
            int64_t local[16] = {0};  // on the thread's own stack: neighbouring histograms would share cache lines
            const int b0 = t * per, b1 = std::min(nblocks, b0 + per);
            for (int b = b0; b < b1; b++) quant_block(type, src + (size_t)b * QK, out + (size_t)b * bs, local);
            for (int i = 0; i < 16; i++) hs[(size_t)t][(size_t)i] = local[i];
        });
    }
    for (auto &x : th) x.join();
    if (hist)
        for (int t = 0; t < nthr; t++)
            for (int i = 0; i < 16; i++) hist[i] += hs[(size_t)t][(size_t)i];
    return (size_t)nblocks * bs;
}
}  // namespace

size_t ggml_quantize_q4_0(const float *src, void *dst, int n, int k, int64_t *hist) {
    return quantize_rows(GGML_TYPE_Q4_0, src, dst, n, k, hist);
}
size_t ggml_quantize_q4_1(const float *src, void *dst, int n, int k, int64_t *hist) {
    return quantize_rows(GGML_TYPE_Q4_1, src, dst, n, k, hist);
}
size_t ggml_quantize_q5_0(const float *src, void *dst, int n, int k, int64_t *hist) {
    return quantize_rows(GGML_TYPE_Q5_0, src, dst, n, k, hist);
}
size_t ggml_quantize_q5_1(const float *src, void *dst, int n, int k, int64_t *hist) {
    return quantize_rows(GGML_TYPE_Q5_1, src, dst, n, k, hist);
}
size_t ggml_quantize_q8_0(const float *src, void *dst, int n, int k, int64_t *hist) {
    return quantize_rows(GGML_TYPE_Q8_0, src, dst, n, k, hist);
}
size_t ggml_quantize_chunk(enum ggml_type type, const float *src, void *dst, int start, int n, int64_t *hist) {
    GGML_ASSERT(start % QK == 0);
    if (type == GGML_TYPE_F16) {
        ggml_fp32_to_fp16_row(src + start, (ggml_fp16_t *)dst + start, n);
        return (size_t)n * 2;
    }
    if (type == GGML_TYPE_F32) {
        memcpy((float *)dst + start, src + start, (size_t)n * 4);
        return (size_t)n * 4;
    }
    uint8_t *out = (uint8_t *)dst + (size_t)(start / QK) * TYPE_INFO[type].size;
    return quantize_rows(type, src + start, out, n, QK, hist);
}

ggml_type_traits_t ggml_internal_get_type_traits(enum ggml_type i) {
    // Only the vec_dot_type column is meaningful here: the dot kernels live on the device and take
    // activations re-quantized to this type, exactly like the CPU table (lib.rs:2900-2906).
    ggml_type_traits_t t;
    memset(&t, 0, sizeof(t));
    switch (i) {
        case GGML_TYPE_Q4_0:
        case GGML_TYPE_Q5_0:
        case GGML_TYPE_Q8_0: t.vec_dot_type = GGML_TYPE_Q8_0; break;
        case GGML_TYPE_Q4_1:
        case GGML_TYPE_Q5_1: t.vec_dot_type = GGML_TYPE_Q8_1; break;
        case GGML_TYPE_Q2_K:
        case GGML_TYPE_Q3_K:
        case GGML_TYPE_Q4_K:
        case GGML_TYPE_Q5_K:
        case GGML_TYPE_Q6_K: t.vec_dot_type = GGML_TYPE_Q8_K; break;
        case GGML_TYPE_F16: t.vec_dot_type = GGML_TYPE_F16; break;
        default: t.vec_dot_type = GGML_TYPE_F32; break;
    }
    return t;
}
