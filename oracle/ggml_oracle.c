/*
 * ggml_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the arithmetic that rustformers/llm's hot path executes below
 * `ggml_graph_compute` (reference call site: crates/ggml/src/lib.rs:374-376) for the LLaMA graph
 * built at crates/models/llama/src/lib.rs:166-362.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this file's shared object; the product library
 * (llm_amd/csrc → libggml_hip.so) never links, loads or calls it.
 *
 * PARITY UNPINNED.  The arithmetic itself lives in the git submodule
 * `crates/ggml/sys/llama-cpp` (ggerganov/llama.cpp: ggml.c, k_quants.c; .gitmodules:1-3;
 * compiled by crates/ggml/sys/build.rs:12-17), which is EMPTY in /root/reference and whose pinned
 * commit is unrecoverable (no .git; API surface dates it to 2023-07-29…2023-08-21, pre-GGUF,
 * see SURVEY.md F1/F2).  The reference holds no golden vector, known-answer test or fixture for
 * this path that works offline.  Every function below therefore restates the
 * *published* upstream algorithm of that window from memory — the scalar `*_reference` /
 * non-SIMD code paths — and is anchored on what IS in tree: type ids and block byte sizes
 * (crates/ggml/sys/src/lib.rs:51-69; sizing rule crates/ggml/src/format/loader.rs:122-124), the
 * vec_dot_type table shape (lib.rs:2900-2906), eps (sys/src/llama.rs:15), and the op wiring of
 * the LLaMA graph.  Each function cites the reference call site it serves.
 *
 * Three modes for every matmul-bearing function (0 and 2 are both "ggml CPU semantics": they differ only where
 * upstream's scalar code and its AVX2 intrinsics code differ; mode 2 = what crates/ggml/sys/build.rs:46-62
 * (-mavx2 -mfma -mf16c on x86-64) actually selects):
 *   mode 0 "exact": ggml CPU semantics — activations re-quantized to the weight type's
 *           vec_dot_type (Q8_0 / Q8_1), integer block dot products, f32 accumulate across blocks;
 *           F16 matmuls round src1 to f16; softmax / SiLU through f16 rounding (ggml's lookup
 *           tables table_exp_f16 / table_silu_f16).
 *   mode 2 "avx2-order": mode 0 with (a) the activation quantizers of upstream's AVX2 branch — id = 127/amax
 *                  instead of 1/(amax/127), round-half-to-even (_mm256_round_ps NEAREST) instead of roundf's
 *                  half-away-from-zero; (b) the vec_dot accumulation of that branch — 8 f32 lanes, lane l =
 *                  fma(d, sum of elements 4l..4l+3 of the block, lane l), horizontal sum at the end
 *                  (hsum_float_8: (x[i+4]+x[i]), then (0+2),(1+3), then +), Q4_1/Q5_1's m*s term in a
 *                  separate scalar float; (c) ggml_vec_dot_f16's F16C branch for the two attention
 *                  products — 4 accumulators of 8 f32 lanes, fma, pairwise reduce.  Everything else (rms_norm,
 *                  rope, soft_max, silu table, adds) is scalar code in both builds.
 *   mode 3 "avx2":  mode 2 executed with the intrinsics themselves (_mm256_maddubs_epi16 / _mm256_madd_epi16 block dots,
 *                  _mm256_fmadd_ps lanes, _mm256_round_ps quantizers, F16C vec_dot_f16) — the code shape of upstream's
 *                  `#elif defined(__AVX2__)` branches, i.e. what the reference's build executes on this host.  Bit-identical
 *                  to mode 2 (tests/test_oracle.py); it exists to be TIMED (bench.py cpu_baseline, kind "port-avx2").
 *   mode 1 "math":  dequantized weights × f32 activations, f64 accumulation, exact expf — the
 *           yardstick that says how much of a difference is activation-quantization noise.
 *
 * Build: make -C oracle   (gcc -O3 -mavx2 -mfma -mf16c -fopenmp; flags of sys/build.rs:46-62)
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* sched_setaffinity: the CPU-baseline leg pins its OpenMP team (orc_pin_threads) */
#endif
#include <float.h>
#include <math.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(__AVX2__) && defined(__FMA__) && defined(__F16C__)
#include <immintrin.h>
#define ORC_HAVE_AVX2 1
#endif

#define QK 32
#define EXPORT __attribute__((visibility("default")))

typedef uint16_t fp16_t;

/* ---- fp16 <-> fp32: IEEE binary16, round-to-nearest-even (what F16C _cvtss_sh/_cvtsh_ss do,
 * which is what ggml uses on the x86 hosts build.rs targets with -mf16c) ----------------------- */
static inline float fp16_to_fp32(fp16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do {
                man <<= 1;
                e++;
            } while ((man & 0x400u) == 0);
            man &= 0x3FFu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static inline fp16_t fp32_to_fp16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t absx = x & 0x7FFFFFFFu;
    if (absx >= 0x7F800000u) { /* inf / nan */
        return (fp16_t)(sign | 0x7C00u | ((absx > 0x7F800000u) ? (0x200u | ((absx >> 13) & 0x3FFu)) : 0));
    }
    if (absx >= 0x477FF000u) { /* >= 65520 rounds to inf */
        return (fp16_t)(sign | 0x7C00u);
    }
    if (absx < 0x33000001u) { /* <= 2^-25: rounds to zero (tie at exactly 2^-25 goes to even = 0) */
        return (fp16_t)sign;
    }
    int32_t e = (int32_t)(absx >> 23) - 127;
    uint32_t man = (absx & 0x7FFFFFu) | 0x800000u; /* 24-bit significand */
    uint32_t shift;
    uint32_t hexp;
    if (e < -14) { /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        hexp = 0;
    } else {
        shift = 13;
        hexp = (uint32_t)(e + 15);
    }
    uint32_t hman = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hman & 1u))) hman++;
    /* hman carries the implicit bit for normals: combine by addition so mantissa overflow bumps exp */
    uint32_t out;
    if (hexp == 0) {
        out = hman; /* may carry into exp=1, which is correct */
    } else {
        out = ((hexp - 1) << 10) + hman; /* hman in [0x400,0x800] */
    }
    return (fp16_t)(sign | out);
}

EXPORT float orc_fp16_to_fp32(fp16_t h) { return fp16_to_fp32(h); }
EXPORT fp16_t orc_fp32_to_fp16(float f) { return fp32_to_fp16(f); }
EXPORT void orc_fp32_to_fp16_row(const float *x, fp16_t *y, int64_t n) {
    for (int64_t i = 0; i < n; i++) y[i] = fp32_to_fp16(x[i]);
}
EXPORT void orc_fp16_to_fp32_row(const fp16_t *x, float *y, int64_t n) {
    for (int64_t i = 0; i < n; i++) y[i] = fp16_to_fp32(x[i]);
}

/* ---- block formats (upstream ggml.c, restated; sizes cross-checked against the in-tree sizing
 * rule bytes = type_size*n/blck_size and the LLaMA-7B Q4_0 file size) --------- */
#pragma pack(push, 1)
typedef struct { fp16_t d; uint8_t qs[16]; } block_q4_0;                       /* 18 B */
typedef struct { fp16_t d; fp16_t m; uint8_t qs[16]; } block_q4_1;             /* 20 B */
typedef struct { fp16_t d; uint8_t qh[4]; uint8_t qs[16]; } block_q5_0;        /* 22 B */
typedef struct { fp16_t d; fp16_t m; uint8_t qh[4]; uint8_t qs[16]; } block_q5_1; /* 24 B */
typedef struct { fp16_t d; int8_t qs[32]; } block_q8_0;                        /* 34 B */
typedef struct { float d; float s; int8_t qs[32]; } block_q8_1;                /* 40 B */
#pragma pack(pop)
_Static_assert(sizeof(block_q4_0) == 18, "q4_0");
_Static_assert(sizeof(block_q4_1) == 20, "q4_1");
_Static_assert(sizeof(block_q5_0) == 22, "q5_0");
_Static_assert(sizeof(block_q5_1) == 24, "q5_1");
_Static_assert(sizeof(block_q8_0) == 34, "q8_0");
_Static_assert(sizeof(block_q8_1) == 40, "q8_1");

enum { T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q4_1 = 3, T_Q5_0 = 6, T_Q5_1 = 7, T_Q8_0 = 8, T_Q8_1 = 9,
       T_Q2_K = 10, T_Q3_K = 11, T_Q4_K = 12, T_Q5_K = 13, T_Q6_K = 14, T_Q8_K = 15 };

/* ---- K-quants (k_quants.c upstream): the checker of the product's Q4_K / Q6_K mat-vec and get_rows
 * (llm_amd/csrc/kernels/kquant.h, tests/test_kquant_gpu.py).  Struct layouts ARE in tree (bindgen: crates/ggml/sys/src/lib.rs:3103-3108,
 * 3240-3245, 3303-3307, sizes 144 / 210 / 292 asserted at :3115, :3252, :3314; QK_K = 256, K_SCALE_SIZE = 12 at
 * :31-32); the arithmetic (scale packing, dequantization, the q8_K dot products) is restated from memory of upstream
 * like everything else here.  The ENCODERS for Q4_K / Q6_K below are plain min/max and abs-max fits, NOT upstream's
 * iterative make_qkx1_quants / make_qx_quants search: any bytes that decode are valid weights for parity work. */
#define QK_K 256
#pragma pack(push, 1)
typedef struct { fp16_t d; fp16_t dmin; uint8_t scales[12]; uint8_t qs[QK_K / 2]; } block_q4_K;           /* 144 B */
typedef struct { uint8_t ql[QK_K / 2]; uint8_t qh[QK_K / 4]; int8_t scales[QK_K / 16]; fp16_t d; } block_q6_K; /* 210 B */
typedef struct { float d; int8_t qs[QK_K]; int16_t bsums[QK_K / 16]; } block_q8_K;                         /* 292 B */
/* crates/ggml/sys/src/lib.rs:2977-2982 (84 B asserted :2989), :3040-3045 (110 B, :3052), :3166-3172 (176 B, :3178) */
typedef struct { uint8_t scales[QK_K / 16]; uint8_t qs[QK_K / 4]; fp16_t d; fp16_t dmin; } block_q2_K;          /* 84 B */
typedef struct { uint8_t hmask[QK_K / 8]; uint8_t qs[QK_K / 4]; uint8_t scales[12]; fp16_t d; } block_q3_K;     /* 110 B */
typedef struct { fp16_t d; fp16_t dmin; uint8_t scales[12]; uint8_t qh[QK_K / 8]; uint8_t qs[QK_K / 2]; } block_q5_K; /* 176 B */
#pragma pack(pop)
_Static_assert(sizeof(block_q4_K) == 144, "q4_K");
_Static_assert(sizeof(block_q6_K) == 210, "q6_K");
_Static_assert(sizeof(block_q8_K) == 292, "q8_K");

EXPORT int orc_type_size(int type) {
    switch (type) {
        case T_F32: return 4;
        case T_F16: return 2;
        case T_Q4_0: return 18;
        case T_Q4_1: return 20;
        case T_Q5_0: return 22;
        case T_Q5_1: return 24;
        case T_Q8_0: return 34;
        case T_Q8_1: return 40;
        case T_Q2_K: return 84;
        case T_Q3_K: return 110;
        case T_Q5_K: return 176;
        case T_Q4_K: return 144;
        case T_Q6_K: return 210;
        case T_Q8_K: return 292;
    }
    return 0;
}
EXPORT int orc_blck_size(int type) {
    if (type == T_Q2_K || type == T_Q3_K || type == T_Q4_K || type == T_Q5_K || type == T_Q6_K || type == T_Q8_K) return QK_K;
    return (type == T_F32 || type == T_F16) ? 1 : QK;
}
/* vec_dot_type column of ggml's type_traits table (shape visible at sys/src/lib.rs:2900-2906) */
EXPORT int orc_vec_dot_type(int type) {
    switch (type) {
        case T_Q4_0: case T_Q5_0: case T_Q8_0: return T_Q8_0;
        case T_Q4_1: case T_Q5_1: return T_Q8_1;
        case T_Q2_K: case T_Q3_K: case T_Q4_K: case T_Q5_K: case T_Q6_K: return T_Q8_K;
        case T_F16: return T_F16;
    }
    return T_F32;
}

#define MIN(a, b) ((a) < (b) ? (a) : (b))

/* quantize_row_q*_reference (upstream ggml.c).  Serves ggml_quantize_q* (sys/src/lib.rs:2779-2822,
 * caller crates/llm-base/src/quantize.rs:363-379) and the activation re-quantization inside
 * ggml_compute_forward_mul_mat. */
static void quantize_row_q4_0(const float *x, block_q4_0 *y, int k) {
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        float amax = 0.0f, max = 0.0f;
        for (int j = 0; j < QK; j++) {
            const float v = x[i * QK + j];
            if (amax < fabsf(v)) { amax = fabsf(v); max = v; }
        }
        const float d = max / -8;
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = fp32_to_fp16(d);
        for (int j = 0; j < QK / 2; ++j) {
            const float x0 = x[i * QK + 0 + j] * id;
            const float x1 = x[i * QK + QK / 2 + j] * id;
            const uint8_t xi0 = MIN(15, (int8_t)(x0 + 8.5f));
            const uint8_t xi1 = MIN(15, (int8_t)(x1 + 8.5f));
            y[i].qs[j] = xi0 | (xi1 << 4);
        }
    }
}
static void quantize_row_q4_1(const float *x, block_q4_1 *y, int k) {
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        float min = FLT_MAX, max = -FLT_MAX;
        for (int j = 0; j < QK; j++) {
            const float v = x[i * QK + j];
            if (v < min) min = v;
            if (v > max) max = v;
        }
        const float d = (max - min) / ((1 << 4) - 1);
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = fp32_to_fp16(d);
        y[i].m = fp32_to_fp16(min);
        for (int j = 0; j < QK / 2; ++j) {
            const float x0 = (x[i * QK + 0 + j] - min) * id;
            const float x1 = (x[i * QK + QK / 2 + j] - min) * id;
            const uint8_t xi0 = MIN(15, (int8_t)(x0 + 0.5f));
            const uint8_t xi1 = MIN(15, (int8_t)(x1 + 0.5f));
            y[i].qs[j] = xi0 | (xi1 << 4);
        }
    }
}
static void quantize_row_q5_0(const float *x, block_q5_0 *y, int k) {
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        float amax = 0.0f, max = 0.0f;
        for (int j = 0; j < QK; j++) {
            const float v = x[i * QK + j];
            if (amax < fabsf(v)) { amax = fabsf(v); max = v; }
        }
        const float d = max / -16;
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = fp32_to_fp16(d);
        uint32_t qh = 0;
        for (int j = 0; j < QK / 2; ++j) {
            const float x0 = x[i * QK + 0 + j] * id;
            const float x1 = x[i * QK + QK / 2 + j] * id;
            const uint8_t xi0 = MIN(31, (int8_t)(x0 + 16.5f));
            const uint8_t xi1 = MIN(31, (int8_t)(x1 + 16.5f));
            y[i].qs[j] = (xi0 & 0x0F) | ((xi1 & 0x0F) << 4);
            qh |= ((xi0 & 0x10u) >> 4) << (j + 0);
            qh |= ((xi1 & 0x10u) >> 4) << (j + QK / 2);
        }
        memcpy(&y[i].qh, &qh, sizeof(qh));
    }
}
static void quantize_row_q5_1(const float *x, block_q5_1 *y, int k) {
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        float min = FLT_MAX, max = -FLT_MAX;
        for (int j = 0; j < QK; j++) {
            const float v = x[i * QK + j];
            if (v < min) min = v;
            if (v > max) max = v;
        }
        const float d = (max - min) / ((1 << 5) - 1);
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = fp32_to_fp16(d);
        y[i].m = fp32_to_fp16(min);
        uint32_t qh = 0;
        for (int j = 0; j < QK / 2; ++j) {
            const float x0 = (x[i * QK + 0 + j] - min) * id;
            const float x1 = (x[i * QK + QK / 2 + j] - min) * id;
            const uint8_t xi0 = (uint8_t)(x0 + 0.5f);
            const uint8_t xi1 = (uint8_t)(x1 + 0.5f);
            y[i].qs[j] = (xi0 & 0x0F) | ((xi1 & 0x0F) << 4);
            qh |= ((xi0 & 0x10u) >> 4) << (j + 0);
            qh |= ((xi1 & 0x10u) >> 4) << (j + QK / 2);
        }
        memcpy(&y[i].qh, &qh, sizeof(qh));
    }
}
static void quantize_row_q8_0(const float *x, block_q8_0 *y, int k) {
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < QK; j++) {
            const float v = x[i * QK + j];
            amax = amax > fabsf(v) ? amax : fabsf(v);
        }
        const float d = amax / ((1 << 7) - 1);
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = fp32_to_fp16(d);
        for (int j = 0; j < QK; ++j) {
            const float x0 = x[i * QK + j] * id;
            y[i].qs[j] = (int8_t)roundf(x0);
        }
    }
}
static void quantize_row_q8_1(const float *x, block_q8_1 *y, int k) {
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < QK; j++) {
            const float v = x[i * QK + j];
            amax = amax > fabsf(v) ? amax : fabsf(v);
        }
        const float d = amax / ((1 << 7) - 1);
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = d;
        int sum = 0;
        for (int j = 0; j < QK / 2; ++j) {
            const float v0 = x[i * QK + j] * id;
            const float v1 = x[i * QK + QK / 2 + j] * id;
            y[i].qs[j] = (int8_t)roundf(v0);
            y[i].qs[QK / 2 + j] = (int8_t)roundf(v1);
            sum += y[i].qs[j];
            sum += y[i].qs[QK / 2 + j];
        }
        y[i].s = sum * d;
    }
}

/* upstream quantize_row_q8_0 / q8_1, `#elif defined(__AVX2__)` branch: same amax and d, but the multiplier is
 * 127/amax (not 1/d) and the rounding is round-half-to-even (_mm256_round_ps(v, _MM_ROUND_NEAREST) followed by
 * cvtps_epi32); Q8_1's s = d * (int sum of the quants). */
static inline float rne_f32(float v) { return nearbyintf(v); } /* default rounding mode = to nearest even */
static void quantize_row_q8_0_avx2(const float *x, block_q8_0 *y, int k) {
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < QK; j++) {
            const float v = fabsf(x[i * QK + j]);
            amax = amax > v ? amax : v;
        }
        const float d = amax / 127.f;
        const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        y[i].d = fp32_to_fp16(d);
        for (int j = 0; j < QK; ++j) y[i].qs[j] = (int8_t)(int)rne_f32(x[i * QK + j] * id);
    }
}
static void quantize_row_q8_1_avx2(const float *x, block_q8_1 *y, int k) {
    const int nb = k / QK;
    for (int i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < QK; j++) {
            const float v = fabsf(x[i * QK + j]);
            amax = amax > v ? amax : v;
        }
        const float d = amax / 127.f;
        const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        y[i].d = d;
        int sum = 0;
        for (int j = 0; j < QK; ++j) {
            y[i].qs[j] = (int8_t)(int)rne_f32(x[i * QK + j] * id);
            sum += y[i].qs[j];
        }
        y[i].s = d * (float)sum;
    }
}

/* ---- K-quant codecs ---------------------------------------------------------------------------- */
static inline int nearest_int(float f) { return (int)lrintf(f); }
/* get_scale_min_k4 (k_quants.c): 8 six-bit (scale, min) pairs in 12 bytes */
static inline void get_scale_min_k4(int j, const uint8_t *q, uint8_t *d, uint8_t *m) {
    if (j < 4) {
        *d = q[j] & 63;
        *m = q[j + 4] & 63;
    } else {
        *d = (uint8_t)((q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4));
        *m = (uint8_t)((q[j + 4] >> 4) | ((q[j - 0] >> 6) << 4));
    }
}
static inline void set_scale_min_k4(int j, uint8_t *q, uint8_t ls, uint8_t lm) { /* inverse of the above */
    if (j < 4) {
        q[j] = (uint8_t)((q[j] & 0xC0) | ls);
        q[j + 4] = (uint8_t)((q[j + 4] & 0xC0) | lm);
    } else {
        q[j + 4] = (uint8_t)((ls & 0xF) | ((lm & 0xF) << 4));
        q[j - 4] = (uint8_t)((q[j - 4] & 0x3F) | ((ls >> 4) << 6));
        q[j - 0] = (uint8_t)((q[j - 0] & 0x3F) | ((lm >> 4) << 6));
    }
}
/* Q4_K: x = d*sc_j*q - dmin*m_j, 8 sub-blocks of 32, q in 0..15, (sc_j, m_j) six-bit.  Encoder: per sub-block min/max
 * fit (min clamped to <= 0 like upstream so that m_j >= 0), super-block scales from the largest sub-block values. */
static void quantize_row_q4_K(const float *x, block_q4_K *y, int k) {
    const int nb = k / QK_K;
    for (int i = 0; i < nb; i++) {
        float scales[8], mins[8], max_scale = 0.0f, max_min = 0.0f;
        for (int j = 0; j < 8; j++) {
            float lo = 0.0f, hi = 0.0f;
            for (int l = 0; l < 32; l++) {
                const float v = x[i * QK_K + 32 * j + l];
                if (v < lo) lo = v;
                if (v > hi) hi = v;
            }
            scales[j] = (hi - lo) / 15.0f;
            mins[j] = -lo;
            if (scales[j] > max_scale) max_scale = scales[j];
            if (mins[j] > max_min) max_min = mins[j];
        }
        const float inv_scale = max_scale > 0 ? 63.0f / max_scale : 0.0f, inv_min = max_min > 0 ? 63.0f / max_min : 0.0f;
        memset(y[i].scales, 0, 12);
        for (int j = 0; j < 8; j++) {
            const int ls = MIN(63, nearest_int(inv_scale * scales[j])), lm = MIN(63, nearest_int(inv_min * mins[j]));
            set_scale_min_k4(j, y[i].scales, (uint8_t)ls, (uint8_t)lm);
        }
        y[i].d = fp32_to_fp16(max_scale / 63.0f);
        y[i].dmin = fp32_to_fp16(max_min / 63.0f);
        uint8_t L[QK_K];
        for (int j = 0; j < 8; j++) {
            uint8_t sc, m;
            get_scale_min_k4(j, y[i].scales, &sc, &m);
            const float d = fp16_to_fp32(y[i].d) * sc, dm = fp16_to_fp32(y[i].dmin) * m;
            for (int l = 0; l < 32; l++) {
                int q = d != 0.0f ? nearest_int((x[i * QK_K + 32 * j + l] + dm) / d) : 0;
                L[32 * j + l] = (uint8_t)(q < 0 ? 0 : q > 15 ? 15 : q);
            }
        }
        uint8_t *q = y[i].qs;
        for (int j = 0; j < QK_K; j += 64) {
            for (int l = 0; l < 32; l++) q[l] = (uint8_t)(L[j + l] | (L[j + l + 32] << 4));
            q += 32;
        }
    }
}
static void dequantize_row_q4_K(const block_q4_K *x, float *y, int k) {
    const int nb = k / QK_K;
    for (int i = 0; i < nb; i++) {
        const uint8_t *q = x[i].qs;
        const float d = fp16_to_fp32(x[i].d), min = fp16_to_fp32(x[i].dmin);
        int is = 0;
        uint8_t sc, m;
        for (int j = 0; j < QK_K; j += 64) {
            get_scale_min_k4(is + 0, x[i].scales, &sc, &m);
            const float d1 = d * sc, m1 = min * m;
            get_scale_min_k4(is + 1, x[i].scales, &sc, &m);
            const float d2 = d * sc, m2 = min * m;
            for (int l = 0; l < 32; ++l) *y++ = d1 * (q[l] & 0xF) - m1;
            for (int l = 0; l < 32; ++l) *y++ = d2 * (q[l] >> 4) - m2;
            q += 32;
            is += 2;
        }
    }
}
/* Q6_K: x = d*sc_j*q, 16 sub-blocks of 16, q in -32..31 (low 4 bits in ql, high 2 in qh), sc_j int8.  Encoder: abs-max
 * fit per sub-block, super-block scale from the largest sub-block scale. */
static void quantize_row_q6_K(const float *x, block_q6_K *y, int k) {
    const int nb = k / QK_K;
    for (int i = 0; i < nb; i++) {
        float scales[16], max_abs_scale = 0.0f, max_scale = 0.0f;
        for (int ib = 0; ib < 16; ib++) {
            float amax = 0.0f, vmax = 0.0f;
            for (int l = 0; l < 16; l++) {
                const float v = x[i * QK_K + 16 * ib + l];
                if (fabsf(v) > amax) { amax = fabsf(v); vmax = v; }
            }
            scales[ib] = amax > 0 ? vmax / -32.0f : 0.0f; /* upstream make_qx_quants: the extreme value maps to -32 */
            if (fabsf(scales[ib]) > max_abs_scale) { max_abs_scale = fabsf(scales[ib]); max_scale = scales[ib]; }
        }
        if (max_abs_scale == 0.0f) {
            memset(&y[i], 0, sizeof(block_q6_K));
            continue;
        }
        const float iscale = -128.0f / max_scale;
        y[i].d = fp32_to_fp16(1.0f / iscale);
        for (int ib = 0; ib < 16; ib++) y[i].scales[ib] = (int8_t)MIN(127, nearest_int(iscale * scales[ib]));
        uint8_t L[QK_K];
        for (int j = 0; j < 16; j++) {
            const float d = fp16_to_fp32(y[i].d) * y[i].scales[j];
            for (int ii = 0; ii < 16; ii++) {
                int l = d != 0.0f ? nearest_int(x[i * QK_K + 16 * j + ii] / d) : 0;
                l = l < -32 ? -32 : l > 31 ? 31 : l;
                L[16 * j + ii] = (uint8_t)(l + 32);
            }
        }
        uint8_t *ql = y[i].ql, *qh = y[i].qh;
        for (int j = 0; j < QK_K; j += 128) {
            for (int l = 0; l < 32; l++) {
                const uint8_t q1 = L[j + l + 0] & 0xF, q2 = L[j + l + 32] & 0xF, q3 = L[j + l + 64] & 0xF, q4 = L[j + l + 96] & 0xF;
                ql[l + 0] = (uint8_t)(q1 | (q3 << 4));
                ql[l + 32] = (uint8_t)(q2 | (q4 << 4));
                qh[l] = (uint8_t)((L[j + l] >> 4) | ((L[j + l + 32] >> 4) << 2) | ((L[j + l + 64] >> 4) << 4) |
                                  ((L[j + l + 96] >> 4) << 6));
            }
            ql += 64;
            qh += 32;
        }
    }
}
static void dequantize_row_q6_K(const block_q6_K *x, float *y, int k) {
    const int nb = k / QK_K;
    for (int i = 0; i < nb; i++) {
        const float d = fp16_to_fp32(x[i].d);
        const uint8_t *ql = x[i].ql, *qh = x[i].qh;
        const int8_t *sc = x[i].scales;
        for (int n = 0; n < QK_K; n += 128) {
            for (int l = 0; l < 32; ++l) {
                const int is = l / 16;
                const int8_t q1 = (int8_t)((ql[l + 0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                const int8_t q2 = (int8_t)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                const int8_t q3 = (int8_t)((ql[l + 0] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                const int8_t q4 = (int8_t)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                y[l + 0] = d * sc[is + 0] * q1;
                y[l + 32] = d * sc[is + 2] * q2;
                y[l + 64] = d * sc[is + 4] * q3;
                y[l + 96] = d * sc[is + 6] * q4;
            }
            y += 128;
            ql += 64;
            qh += 32;
            sc += 8;
        }
    }
}
/* quantize_row_q8_K_reference (k_quants.c): the activation side of every K-quant dot product — f32 scale, the extreme
 * value maps to -128 (clamped to 127 on the other side), 16-element partial sums kept for the `min` terms */
static void quantize_row_q8_K(const float *x, block_q8_K *y, int k) {
    const int nb = k / QK_K;
    for (int i = 0; i < nb; i++) {
        float max = 0.0f, amax = 0.0f;
        for (int j = 0; j < QK_K; ++j) {
            const float ax = fabsf(x[j]);
            if (ax > amax) { amax = ax; max = x[j]; }
        }
        if (!amax) {
            y[i].d = 0;
            memset(y[i].qs, 0, QK_K);
            memset(y[i].bsums, 0, sizeof(y[i].bsums));
            x += QK_K;
            continue;
        }
        const float iscale = -128.f / max;
        for (int j = 0; j < QK_K; ++j) y[i].qs[j] = (int8_t)MIN(127, nearest_int(iscale * x[j]));
        for (int j = 0; j < QK_K / 16; ++j) {
            int sum = 0;
            for (int ii = 0; ii < 16; ++ii) sum += y[i].qs[j * 16 + ii];
            y[i].bsums[j] = (int16_t)sum;
        }
        y[i].d = 1 / iscale;
        x += QK_K;
    }
}
static void dequantize_row_q8_K(const block_q8_K *x, float *y, int k) {
    for (int i = 0; i < k / QK_K; i++)
        for (int j = 0; j < QK_K; ++j) *y++ = x[i].d * x[i].qs[j];
}

/* ---- Q2_K / Q3_K / Q5_K.  Decoders and dot products restate upstream k_quants.c (scalar branches);
 * the ENCODERS below are simple min/max (abs-max) fits that produce VALID blocks for tests — not upstream's iterative
 * make_qkx / make_q3 searches (oracle/SEMANTICS.md). ---------------------------------------------------------------- */
/* A second legal summation order of every block dot (orc_set_block_order(1)): the blocks / super-blocks of a row are walked
 * downwards instead of upwards.  ggml's scalar and SIMD branches differ from each other by re-associations of this size; the
 * difference between the two orders is the yardstick ("band") the model-level parity checks measure in the same run. */
static int g_rev = 0;
EXPORT void orc_set_block_order(int reverse) { g_rev = reverse; }
#define BLOCK_LOOP(i, nb) for (int ii_ = 0, i = g_rev ? (nb)-1 : 0; ii_ < (nb); ii_++, i += g_rev ? -1 : 1)

/* Q2_K: x = d*(sc&15)*q - dmin*(sc>>4), 16 sub-blocks of 16, q in 0..3.  Element e: n = e/128, j = (e%128)/32, h = (e%32)/16:
 * byte qs[32n + 16h + e%16], bits 2j..2j+1; scale byte scales[8n + 2j + h]. */
static void quantize_row_q2_K(const float *x, block_q2_K *y, int k) {
    const int nb = k / QK_K;
    for (int i = 0; i < nb; i++) {
        float sc[16], mn[16], max_sc = 0.0f, max_mn = 0.0f;
        for (int j = 0; j < 16; j++) {
            float lo = 0.0f, hi = 0.0f;
            for (int l = 0; l < 16; l++) {
                const float v = x[i * QK_K + 16 * j + l];
                if (v < lo) lo = v;
                if (v > hi) hi = v;
            }
            sc[j] = (hi - lo) / 3.0f;
            mn[j] = -lo;
            if (sc[j] > max_sc) max_sc = sc[j];
            if (mn[j] > max_mn) max_mn = mn[j];
        }
        const float isc = max_sc > 0 ? 15.0f / max_sc : 0.0f, imn = max_mn > 0 ? 15.0f / max_mn : 0.0f;
        y[i].d = fp32_to_fp16(max_sc / 15.0f);
        y[i].dmin = fp32_to_fp16(max_mn / 15.0f);
        uint8_t L[QK_K];
        for (int j = 0; j < 16; j++) {
            const int ls = MIN(15, nearest_int(isc * sc[j])), lm = MIN(15, nearest_int(imn * mn[j]));
            y[i].scales[j] = (uint8_t)(ls | (lm << 4));
            const float d = fp16_to_fp32(y[i].d) * ls, dm = fp16_to_fp32(y[i].dmin) * lm;
            for (int l = 0; l < 16; l++) {
                int q = d != 0.0f ? nearest_int((x[i * QK_K + 16 * j + l] + dm) / d) : 0;
                L[16 * j + l] = (uint8_t)(q < 0 ? 0 : q > 3 ? 3 : q);
            }
        }
        for (int j = 0; j < QK_K; j += 128)
            for (int l = 0; l < 32; l++)
                y[i].qs[j / 4 + l] = (uint8_t)(L[j + l] | (L[j + l + 32] << 2) | (L[j + l + 64] << 4) | (L[j + l + 96] << 6));
    }
}
static void dequantize_row_q2_K(const block_q2_K *x, float *y, int k) {
    const int nb = k / QK_K;
    for (int i = 0; i < nb; i++) {
        const float d = fp16_to_fp32(x[i].d), min = fp16_to_fp32(x[i].dmin);
        const uint8_t *q = x[i].qs;
        int is = 0;
        for (int n = 0; n < QK_K; n += 128) {
            int shift = 0;
            for (int j = 0; j < 4; ++j) {
                uint8_t sc = x[i].scales[is++];
                float dl = d * (sc & 0xF), ml = min * (sc >> 4);
                for (int l = 0; l < 16; ++l) *y++ = dl * ((int8_t)((q[l] >> shift) & 3)) - ml;
                sc = x[i].scales[is++];
                dl = d * (sc & 0xF);
                ml = min * (sc >> 4);
                for (int l = 0; l < 16; ++l) *y++ = dl * ((int8_t)((q[l + 16] >> shift) & 3)) - ml;
                shift += 2;
            }
            q += 32;
        }
    }
}
static float vec_dot_q2_K_q8_K(int n, const block_q2_K *x, const block_q8_K *y) {
    const int nb = n / QK_K;
    float sumf = 0.0f;
    BLOCK_LOOP(i, nb) {
        const uint8_t *q2 = x[i].qs;
        const int8_t *q8 = y[i].qs;
        const uint8_t *sc = x[i].scales;
        int summs = 0;
        for (int j = 0; j < 16; ++j) summs += y[i].bsums[j] * (sc[j] >> 4);
        const float dall = y[i].d * fp16_to_fp32(x[i].d), dmin = y[i].d * fp16_to_fp32(x[i].dmin);
        int isum = 0, is = 0;
        for (int kk = 0; kk < QK_K / 128; ++kk) {
            int shift = 0;
            for (int j = 0; j < 4; ++j) {
                int d = sc[is++] & 0xF, isuml = 0;
                for (int l = 0; l < 16; ++l) isuml += q8[l] * ((q2[l] >> shift) & 3);
                isum += d * isuml;
                d = sc[is++] & 0xF;
                isuml = 0;
                for (int l = 16; l < 32; ++l) isuml += q8[l] * ((q2[l] >> shift) & 3);
                isum += d * isuml;
                shift += 2;
                q8 += 32;
            }
            q2 += 32;
        }
        sumf += dall * isum - dmin * summs;
    }
    return sumf;
}
/* Q3_K: x = d*(sc_j - 32)*q, 16 sub-blocks of 16, q in -4..3: two low bits in qs (as Q2_K), the third in hmask (bit
 * 4n + j of hmask[16h + e%16]; bit CLEAR means subtract 4); sc_j: 6 bits packed in 12 bytes. */
static void q3_unpack_scales(const uint8_t *packed, int8_t *scales /*16*/) {
    const uint32_t kmask1 = 0x03030303, kmask2 = 0x0f0f0f0f;
    uint32_t aux[4];
    memcpy(aux, packed, 12);
    const uint32_t tmp = aux[2];
    aux[2] = ((aux[0] >> 4) & kmask2) | (((tmp >> 4) & kmask1) << 4);
    aux[3] = ((aux[1] >> 4) & kmask2) | (((tmp >> 6) & kmask1) << 4);
    aux[0] = (aux[0] & kmask2) | (((tmp >> 0) & kmask1) << 4);
    aux[1] = (aux[1] & kmask2) | (((tmp >> 2) & kmask1) << 4);
    memcpy(scales, aux, 16);
}
static void quantize_row_q3_K(const float *x, block_q3_K *y, int k) {
    const int nb = k / QK_K;
    for (int i = 0; i < nb; i++) {
        float sc[16], max_abs = 0.0f, max_sc = 0.0f;
        for (int j = 0; j < 16; j++) {
            float amax = 0.0f, vmax = 0.0f;
            for (int l = 0; l < 16; l++) {
                const float v = x[i * QK_K + 16 * j + l];
                if (fabsf(v) > amax) { amax = fabsf(v); vmax = v; }
            }
            sc[j] = amax > 0 ? vmax / -4.0f : 0.0f; /* the extreme value maps to -4 */
            if (fabsf(sc[j]) > max_abs) { max_abs = fabsf(sc[j]); max_sc = sc[j]; }
        }
        memset(&y[i], 0, sizeof(block_q3_K));
        if (max_abs == 0.0f) {
            for (int j = 0; j < 16; j++) { /* scales 32 -> (32 - 32) = 0 */
                const int l = 32;
                if (j < 8) y[i].scales[j] = l & 0xF; else y[i].scales[j - 8] |= ((l & 0xF) << 4);
                y[i].scales[j % 4 + 8] |= ((l >> 4) << (2 * (j / 4)));
            }
            for (int j = 0; j < QK_K / 8; j++) y[i].hmask[j] = 0xFF; /* q = 0 */
            continue;
        }
        const float iscale = -32.0f / max_sc;
        y[i].d = fp32_to_fp16(1.0f / iscale);
        int8_t ls[16];
        for (int j = 0; j < 16; j++) {
            int l = nearest_int(iscale * sc[j]);
            l = l < -32 ? -32 : l > 31 ? 31 : l;
            ls[j] = (int8_t)l;
            l += 32;
            if (j < 8) y[i].scales[j] = l & 0xF; else y[i].scales[j - 8] |= ((l & 0xF) << 4);
            y[i].scales[j % 4 + 8] |= ((l >> 4) << (2 * (j / 4)));
        }
        uint8_t L[QK_K]; /* q + 4 in 0..7 */
        for (int j = 0; j < 16; j++) {
            const float d = fp16_to_fp32(y[i].d) * ls[j];
            for (int l = 0; l < 16; l++) {
                int q = d != 0.0f ? nearest_int(x[i * QK_K + 16 * j + l] / d) : 0;
                q = q < -4 ? -4 : q > 3 ? 3 : q;
                L[16 * j + l] = (uint8_t)(q + 4);
            }
        }
        int m = 0;
        uint8_t hm = 1;
        for (int j = 0; j < QK_K; ++j) { /* upstream's order: bit hm of hmask[j % 32] for elements j = 32*bit .. */
            if (L[j] > 3) {
                y[i].hmask[m] |= hm;
                L[j] -= 4;
            }
            if (++m == QK_K / 8) {
                m = 0;
                hm <<= 1;
            }
        }
        for (int j = 0; j < QK_K; j += 128)
            for (int l = 0; l < 32; l++)
                y[i].qs[j / 4 + l] = (uint8_t)(L[j + l] | (L[j + l + 32] << 2) | (L[j + l + 64] << 4) | (L[j + l + 96] << 6));
    }
}
static void dequantize_row_q3_K(const block_q3_K *x, float *y, int k) {
    const int nb = k / QK_K;
    for (int i = 0; i < nb; i++) {
        const float d_all = fp16_to_fp32(x[i].d);
        const uint8_t *q = x[i].qs, *hm = x[i].hmask;
        uint8_t m = 1;
        int8_t scales[16];
        q3_unpack_scales(x[i].scales, scales);
        int is = 0;
        for (int n = 0; n < QK_K; n += 128) {
            int shift = 0;
            for (int j = 0; j < 4; ++j) {
                float dl = d_all * (scales[is++] - 32);
                for (int l = 0; l < 16; ++l) *y++ = dl * ((int8_t)((q[l + 0] >> shift) & 3) - ((hm[l + 0] & m) ? 0 : 4));
                dl = d_all * (scales[is++] - 32);
                for (int l = 0; l < 16; ++l) *y++ = dl * ((int8_t)((q[l + 16] >> shift) & 3) - ((hm[l + 16] & m) ? 0 : 4));
                shift += 2;
                m <<= 1;
            }
            q += 32;
        }
    }
}
/* upstream's scalar K dots keep EIGHT running f32 sums (sums[l], l = element index mod 8) and add them at the end */
static float vec_dot_q3_K_q8_K(int n, const block_q3_K *x, const block_q8_K *y) {
    const int nb = n / QK_K;
    int8_t aux8[QK_K];
    int16_t aux16[8];
    float sums[8];
    int32_t aux32[8];
    memset(sums, 0, sizeof(sums));
    float sumf = 0;
    BLOCK_LOOP(i, nb) {
        const uint8_t *q3 = x[i].qs, *hm = x[i].hmask;
        const int8_t *q8 = y[i].qs;
        memset(aux32, 0, sizeof(aux32));
        int8_t *a = aux8;
        uint8_t m = 1;
        for (int j = 0; j < QK_K; j += 128) {
            for (int sh = 0; sh < 8; sh += 2) {
                for (int l = 0; l < 32; ++l) a[l] = (int8_t)((q3[l] >> sh) & 3);
                for (int l = 0; l < 32; ++l) a[l] -= (hm[l] & m ? 0 : 4);
                a += 32;
                m <<= 1;
            }
            q3 += 32;
        }
        a = aux8;
        int8_t scales[16];
        q3_unpack_scales(x[i].scales, scales);
        for (int j = 0; j < QK_K / 16; ++j) {
            for (int half = 0; half < 2; half++) {
                for (int l = 0; l < 8; ++l) aux16[l] = (int16_t)(q8[l] * a[l]);
                for (int l = 0; l < 8; ++l) aux32[l] += (scales[j] - 32) * aux16[l];
                q8 += 8;
                a += 8;
            }
        }
        const float d = fp16_to_fp32(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}
/* Q5_K: Q4_K with a fifth bit: x = d*sc_j*q - dmin*m_j, q in 0..31; bit (2*(e/64) + (e%64)/32) of qh[e%32] */
static void quantize_row_q5_K(const float *x, block_q5_K *y, int k) {
    const int nb = k / QK_K;
    for (int i = 0; i < nb; i++) {
        float scales[8], mins[8], max_scale = 0.0f, max_min = 0.0f;
        for (int j = 0; j < 8; j++) {
            float lo = 0.0f, hi = 0.0f;
            for (int l = 0; l < 32; l++) {
                const float v = x[i * QK_K + 32 * j + l];
                if (v < lo) lo = v;
                if (v > hi) hi = v;
            }
            scales[j] = (hi - lo) / 31.0f;
            mins[j] = -lo;
            if (scales[j] > max_scale) max_scale = scales[j];
            if (mins[j] > max_min) max_min = mins[j];
        }
        const float inv_scale = max_scale > 0 ? 63.0f / max_scale : 0.0f, inv_min = max_min > 0 ? 63.0f / max_min : 0.0f;
        memset(y[i].scales, 0, 12);
        for (int j = 0; j < 8; j++) {
            const int ls = MIN(63, nearest_int(inv_scale * scales[j])), lm = MIN(63, nearest_int(inv_min * mins[j]));
            set_scale_min_k4(j, y[i].scales, (uint8_t)ls, (uint8_t)lm);
        }
        y[i].d = fp32_to_fp16(max_scale / 63.0f);
        y[i].dmin = fp32_to_fp16(max_min / 63.0f);
        uint8_t L[QK_K];
        for (int j = 0; j < 8; j++) {
            uint8_t sc, m;
            get_scale_min_k4(j, y[i].scales, &sc, &m);
            const float d = fp16_to_fp32(y[i].d) * sc, dm = fp16_to_fp32(y[i].dmin) * m;
            for (int l = 0; l < 32; l++) {
                int q = d != 0.0f ? nearest_int((x[i * QK_K + 32 * j + l] + dm) / d) : 0;
                L[32 * j + l] = (uint8_t)(q < 0 ? 0 : q > 31 ? 31 : q);
            }
        }
        uint8_t *qh = y[i].qh, *ql = y[i].qs;
        memset(qh, 0, QK_K / 8);
        uint8_t m1 = 1, m2 = 2;
        for (int n2 = 0; n2 < QK_K; n2 += 64) {
            for (int j = 0; j < 32; ++j) {
                int l1 = L[n2 + j];
                if (l1 > 15) { l1 -= 16; qh[j] |= m1; }
                int l2 = L[n2 + j + 32];
                if (l2 > 15) { l2 -= 16; qh[j] |= m2; }
                ql[j] = (uint8_t)(l1 | (l2 << 4));
            }
            m1 <<= 2;
            m2 <<= 2;
            ql += 32;
        }
    }
}
static void dequantize_row_q5_K(const block_q5_K *x, float *y, int k) {
    const int nb = k / QK_K;
    for (int i = 0; i < nb; i++) {
        const uint8_t *ql = x[i].qs, *qh = x[i].qh;
        const float d = fp16_to_fp32(x[i].d), min = fp16_to_fp32(x[i].dmin);
        int is = 0;
        uint8_t sc, m, u1 = 1, u2 = 2;
        for (int j = 0; j < QK_K; j += 64) {
            get_scale_min_k4(is + 0, x[i].scales, &sc, &m);
            const float d1 = d * sc, m1 = min * m;
            get_scale_min_k4(is + 1, x[i].scales, &sc, &m);
            const float d2 = d * sc, m2 = min * m;
            for (int l = 0; l < 32; ++l) *y++ = d1 * ((ql[l] & 0xF) + (qh[l] & u1 ? 16 : 0)) - m1;
            for (int l = 0; l < 32; ++l) *y++ = d2 * ((ql[l] >> 4) + (qh[l] & u2 ? 16 : 0)) - m2;
            ql += 32;
            is += 2;
            u1 <<= 2;
            u2 <<= 2;
        }
    }
}
static float vec_dot_q5_K_q8_K(int n, const block_q5_K *x, const block_q8_K *y) {
    const int nb = n / QK_K;
    int8_t aux8[QK_K];
    int16_t aux16[8];
    float sums[8];
    int32_t aux32[8];
    memset(sums, 0, sizeof(sums));
    float sumf = 0;
    BLOCK_LOOP(i, nb) {
        const uint8_t *q4 = x[i].qs, *hm = x[i].qh;
        const int8_t *q8 = y[i].qs;
        memset(aux32, 0, sizeof(aux32));
        int8_t *a = aux8;
        uint8_t m = 1;
        for (int j = 0; j < QK_K / 64; ++j) {
            for (int l = 0; l < 32; ++l) a[l] = (int8_t)(q4[l] & 0xF);
            for (int l = 0; l < 32; ++l) a[l] += (hm[l] & m ? 16 : 0);
            a += 32;
            m <<= 1;
            for (int l = 0; l < 32; ++l) a[l] = (int8_t)(q4[l] >> 4);
            for (int l = 0; l < 32; ++l) a[l] += (hm[l] & m ? 16 : 0);
            a += 32;
            m <<= 1;
            q4 += 32;
        }
        uint8_t scales[8], mins[8];
        for (int j = 0; j < 8; j++) get_scale_min_k4(j, x[i].scales, &scales[j], &mins[j]);
        int sumi = 0;
        for (int j = 0; j < QK_K / 16; ++j) sumi += y[i].bsums[j] * mins[j / 2];
        a = aux8;
        int is = 0;
        for (int j = 0; j < QK_K / 32; ++j) {
            const int32_t scale = scales[is++];
            for (int g = 0; g < 4; g++) {
                for (int l = 0; l < 8; ++l) aux16[l] = (int16_t)(q8[l] * a[l]);
                for (int l = 0; l < 8; ++l) aux32[l] += scale * aux16[l];
                q8 += 8;
                a += 8;
            }
        }
        const float d = fp16_to_fp32(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
        const float dmin = fp16_to_fp32(x[i].dmin) * y[i].d;
        sumf -= dmin * sumi;
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}
/* ggml_vec_dot_q4_K_q8_K, scalar branch: sumf = sum_i d8*d*(sum_j sc_j * <q4_j, q8_j>) - d8*dmin*(sum_j m_j * bsum_j), with the
 * integer products gathered in EIGHT lanes (element index mod 8) that each keep a running f32 sum, added at the end */
static float vec_dot_q4_K_q8_K(int n, const block_q4_K *x, const block_q8_K *y) {
    const int nb = n / QK_K;
    int8_t aux8[QK_K];
    int16_t aux16[8];
    float sums[8];
    int32_t aux32[8];
    memset(sums, 0, sizeof(sums));
    float sumf = 0;
    BLOCK_LOOP(i, nb) {
        const uint8_t *q4 = x[i].qs;
        const int8_t *q8 = y[i].qs;
        memset(aux32, 0, sizeof(aux32));
        int8_t *a = aux8;
        for (int j = 0; j < QK_K / 64; ++j) {
            for (int l = 0; l < 32; ++l) a[l] = (int8_t)(q4[l] & 0xF);
            a += 32;
            for (int l = 0; l < 32; ++l) a[l] = (int8_t)(q4[l] >> 4);
            a += 32;
            q4 += 32;
        }
        uint8_t scales[8], mins[8];
        for (int j = 0; j < 8; j++) get_scale_min_k4(j, x[i].scales, &scales[j], &mins[j]);
        int sumi = 0;
        for (int j = 0; j < QK_K / 16; ++j) sumi += y[i].bsums[j] * mins[j / 2];
        a = aux8;
        for (int j = 0; j < QK_K / 32; ++j) {
            const int32_t scale = scales[j];
            for (int g = 0; g < 4; g++) {
                for (int l = 0; l < 8; ++l) aux16[l] = (int16_t)(q8[l] * a[l]);
                for (int l = 0; l < 8; ++l) aux32[l] += scale * aux16[l];
                q8 += 8;
                a += 8;
            }
        }
        const float d = fp16_to_fp32(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
        const float dmin = fp16_to_fp32(x[i].dmin) * y[i].d;
        sumf -= dmin * sumi;
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}
/* ggml_vec_dot_q6_K_q8_K, scalar branch: sumf = sum_i d8*d * sum_{16-blocks} sc * <q6, q8>, eight lanes as above */
static float vec_dot_q6_K_q8_K(int n, const block_q6_K *x, const block_q8_K *y) {
    const int nb = n / QK_K;
    int8_t aux8[QK_K];
    int16_t aux16[8];
    float sums[8];
    int32_t aux32[8];
    memset(sums, 0, sizeof(sums));
    float sumf = 0;
    BLOCK_LOOP(i, nb) {
        const uint8_t *ql = x[i].ql, *qh = x[i].qh;
        const int8_t *q8 = y[i].qs;
        memset(aux32, 0, sizeof(aux32));
        int8_t *a = aux8;
        for (int j = 0; j < QK_K; j += 128) {
            for (int l = 0; l < 32; ++l) {
                a[l + 0] = (int8_t)((ql[l + 0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                a[l + 32] = (int8_t)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                a[l + 64] = (int8_t)((ql[l + 0] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                a[l + 96] = (int8_t)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
            }
            a += 128;
            ql += 64;
            qh += 32;
        }
        a = aux8;
        for (int j = 0; j < QK_K / 16; ++j) {
            const int scale = x[i].scales[j];
            for (int half = 0; half < 2; half++) {
                for (int l = 0; l < 8; ++l) aux16[l] = (int16_t)(q8[l] * a[l]);
                for (int l = 0; l < 8; ++l) aux32[l] += scale * aux16[l];
                q8 += 8;
                a += 8;
            }
        }
        const float d = fp16_to_fp32(x[i].d) * y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d * aux32[l];
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}

EXPORT void orc_quantize_row(int type, const float *x, void *y, int k) {
    switch (type) {
        case T_Q4_0: quantize_row_q4_0(x, (block_q4_0 *)y, k); break;
        case T_Q4_1: quantize_row_q4_1(x, (block_q4_1 *)y, k); break;
        case T_Q5_0: quantize_row_q5_0(x, (block_q5_0 *)y, k); break;
        case T_Q5_1: quantize_row_q5_1(x, (block_q5_1 *)y, k); break;
        case T_Q8_0: quantize_row_q8_0(x, (block_q8_0 *)y, k); break;
        case T_Q8_1: quantize_row_q8_1(x, (block_q8_1 *)y, k); break;
        case T_Q2_K: quantize_row_q2_K(x, (block_q2_K *)y, k); break;
        case T_Q3_K: quantize_row_q3_K(x, (block_q3_K *)y, k); break;
        case T_Q5_K: quantize_row_q5_K(x, (block_q5_K *)y, k); break;
        case T_Q4_K: quantize_row_q4_K(x, (block_q4_K *)y, k); break;
        case T_Q6_K: quantize_row_q6_K(x, (block_q6_K *)y, k); break;
        case T_Q8_K: quantize_row_q8_K(x, (block_q8_K *)y, k); break;
        case T_F16: orc_fp32_to_fp16_row(x, (fp16_t *)y, k); break;
        case T_F32: memcpy(y, x, (size_t)k * 4); break;
        default: fprintf(stderr, "orc_quantize_row: bad type %d\n", type); abort();
    }
}

/* dequantize_row_q* (upstream ggml.c).  Serves get_rows (models/llama/src/lib.rs:170) and the
 * "math" yardstick. */
EXPORT void orc_dequantize_row(int type, const void *vx, float *y, int k) {
    const int nb = k / QK;
    switch (type) {
        case T_Q2_K: dequantize_row_q2_K((const block_q2_K *)vx, y, k); return;
        case T_Q3_K: dequantize_row_q3_K((const block_q3_K *)vx, y, k); return;
        case T_Q5_K: dequantize_row_q5_K((const block_q5_K *)vx, y, k); return;
        case T_Q4_K: dequantize_row_q4_K((const block_q4_K *)vx, y, k); return;
        case T_Q6_K: dequantize_row_q6_K((const block_q6_K *)vx, y, k); return;
        case T_Q8_K: dequantize_row_q8_K((const block_q8_K *)vx, y, k); return;
        case T_Q4_0: {
            const block_q4_0 *x = (const block_q4_0 *)vx;
            for (int i = 0; i < nb; i++) {
                const float d = fp16_to_fp32(x[i].d);
                for (int j = 0; j < QK / 2; ++j) {
                    const int x0 = (x[i].qs[j] & 0x0F) - 8;
                    const int x1 = (x[i].qs[j] >> 4) - 8;
                    y[i * QK + j + 0] = x0 * d;
                    y[i * QK + j + QK / 2] = x1 * d;
                }
            }
        } break;
        case T_Q4_1: {
            const block_q4_1 *x = (const block_q4_1 *)vx;
            for (int i = 0; i < nb; i++) {
                const float d = fp16_to_fp32(x[i].d);
                const float m = fp16_to_fp32(x[i].m);
                for (int j = 0; j < QK / 2; ++j) {
                    const int x0 = (x[i].qs[j] & 0x0F);
                    const int x1 = (x[i].qs[j] >> 4);
                    y[i * QK + j + 0] = x0 * d + m;
                    y[i * QK + j + QK / 2] = x1 * d + m;
                }
            }
        } break;
        case T_Q5_0: {
            const block_q5_0 *x = (const block_q5_0 *)vx;
            for (int i = 0; i < nb; i++) {
                const float d = fp16_to_fp32(x[i].d);
                uint32_t qh;
                memcpy(&qh, x[i].qh, sizeof(qh));
                for (int j = 0; j < QK / 2; ++j) {
                    const uint8_t xh_0 = ((qh >> (j + 0)) << 4) & 0x10;
                    const uint8_t xh_1 = ((qh >> (j + 12))) & 0x10;
                    const int32_t x0 = ((x[i].qs[j] & 0x0F) | xh_0) - 16;
                    const int32_t x1 = ((x[i].qs[j] >> 4) | xh_1) - 16;
                    y[i * QK + j + 0] = x0 * d;
                    y[i * QK + j + QK / 2] = x1 * d;
                }
            }
        } break;
        case T_Q5_1: {
            const block_q5_1 *x = (const block_q5_1 *)vx;
            for (int i = 0; i < nb; i++) {
                const float d = fp16_to_fp32(x[i].d);
                const float m = fp16_to_fp32(x[i].m);
                uint32_t qh;
                memcpy(&qh, x[i].qh, sizeof(qh));
                for (int j = 0; j < QK / 2; ++j) {
                    const uint8_t xh_0 = ((qh >> (j + 0)) << 4) & 0x10;
                    const uint8_t xh_1 = ((qh >> (j + 12))) & 0x10;
                    const int x0 = (x[i].qs[j] & 0x0F) | xh_0;
                    const int x1 = (x[i].qs[j] >> 4) | xh_1;
                    y[i * QK + j + 0] = x0 * d + m;
                    y[i * QK + j + QK / 2] = x1 * d + m;
                }
            }
        } break;
        case T_Q8_0: {
            const block_q8_0 *x = (const block_q8_0 *)vx;
            for (int i = 0; i < nb; i++) {
                const float d = fp16_to_fp32(x[i].d);
                for (int j = 0; j < QK; ++j) y[i * QK + j] = x[i].qs[j] * d;
            }
        } break;
        case T_F16: orc_fp16_to_fp32_row((const fp16_t *)vx, y, k); break;
        case T_F32: memcpy(y, vx, (size_t)k * 4); break;
        default: fprintf(stderr, "orc_dequantize_row: bad type %d\n", type); abort();
    }
}

/* ggml_quantize_q*(src, dst, n, k, hist) — sys/src/lib.rs:2779-2822.  n elements total in rows of k. */
EXPORT size_t orc_quantize(int type, const float *src, void *dst, int n, int k, int64_t *hist) {
    const int nb = k / QK;
    const size_t bs = (size_t)orc_type_size(type);
    const int blck = orc_blck_size(type);
    for (int b = 0; b < n; b += k) {
        uint8_t *y = (uint8_t *)dst + (size_t)(b / blck) * bs;
        orc_quantize_row(type, src + b, y, k);
        if (!hist || blck != QK) continue; /* no histogram for the K-quant groundwork */
        for (int i = 0; i < nb; i++) {
            const uint8_t *blk = y + (size_t)i * bs;
            switch (type) {
                case T_Q4_0:
                case T_Q4_1: {
                    const uint8_t *qs = blk + (type == T_Q4_0 ? 2 : 4);
                    for (int j = 0; j < QK; j += 2) {
                        hist[qs[j / 2] & 0xF]++;
                        hist[qs[j / 2] >> 4]++;
                    }
                } break;
                case T_Q5_0:
                case T_Q5_1: {
                    const uint8_t *qhp = blk + (type == T_Q5_0 ? 2 : 4);
                    const uint8_t *qs = qhp + 4;
                    uint32_t qh;
                    memcpy(&qh, qhp, 4);
                    for (int j = 0; j < QK; j += 2) {
                        const uint8_t vh0 = ((qh & (1u << (j + 0))) >> (j + 0)) << 4;
                        /* upstream writes (1u << (j + 16)) >> (j + 12) with j up to 30; x86 wraps the count */
                        const uint8_t vh1 = ((qh & (1u << ((j + 16) & 31))) >> ((j + 12) & 31));
                        /* cast to 16 bins */
                        const uint8_t vi0 = ((qs[j / 2] & 0x0F) | vh0) / 2;
                        const uint8_t vi1 = ((qs[j / 2] >> 4) | vh1) / 2;
                        hist[vi0]++;
                        hist[vi1]++;
                    }
                } break;
                case T_Q8_0: {
                    const int8_t *qs = (const int8_t *)(blk + 2);
                    for (int j = 0; j < QK; ++j) hist[qs[j] / 16 + 8]++;
                } break;
            }
        }
    }
    return (size_t)(n / blck) * bs;
}

/* ---- vec_dot (upstream ggml_vec_dot_q*_q8_*, scalar branch).  The heart of
 * ggml_compute_forward_mul_mat for quantized src0. ------------------------- */
/* Order in which a row's blocks are added into sumf.  0 = ascending (ggml's scalar code).  1 = descending:
 * NOT a ggml mode — a yardstick for how much a legal re-association of the f32 block sum (which ggml's own
 * AVX2 path, 8 lanes + horizontal add, also performs) moves the results of a given model. */

static float vec_dot_q4_0_q8_0(int n, const block_q4_0 *x, const block_q8_0 *y) {
    const int nb = n / QK;
    float sumf = 0.0f;
    BLOCK_LOOP(i, nb) {
        int sumi = 0;
        for (int j = 0; j < QK / 2; ++j) {
            const int v0 = (x[i].qs[j] & 0x0F) - 8;
            const int v1 = (x[i].qs[j] >> 4) - 8;
            sumi += (v0 * y[i].qs[j]) + (v1 * y[i].qs[j + QK / 2]);
        }
        sumf += sumi * fp16_to_fp32(x[i].d) * fp16_to_fp32(y[i].d);
    }
    return sumf;
}
static float vec_dot_q4_1_q8_1(int n, const block_q4_1 *x, const block_q8_1 *y) {
    const int nb = n / QK;
    float sumf = 0.0f;
    BLOCK_LOOP(i, nb) {
        int sumi = 0;
        for (int j = 0; j < QK / 2; ++j) {
            const int v0 = (x[i].qs[j] & 0x0F);
            const int v1 = (x[i].qs[j] >> 4);
            sumi += (v0 * y[i].qs[j]) + (v1 * y[i].qs[j + QK / 2]);
        }
        sumf += (fp16_to_fp32(x[i].d) * y[i].d) * sumi + fp16_to_fp32(x[i].m) * y[i].s;
    }
    return sumf;
}
static float vec_dot_q5_0_q8_0(int n, const block_q5_0 *x, const block_q8_0 *y) {
    const int nb = n / QK;
    float sumf = 0.0f;
    BLOCK_LOOP(i, nb) {
        uint32_t qh;
        memcpy(&qh, x[i].qh, sizeof(qh));
        int sumi = 0;
        for (int j = 0; j < QK / 2; ++j) {
            const uint8_t xh_0 = ((qh & (1u << (j + 0))) >> (j + 0)) << 4;
            const uint8_t xh_1 = ((qh & (1u << (j + 16))) >> (j + 12));
            const int32_t x0 = ((x[i].qs[j] & 0x0F) | xh_0) - 16;
            const int32_t x1 = ((x[i].qs[j] >> 4) | xh_1) - 16;
            sumi += (x0 * y[i].qs[j]) + (x1 * y[i].qs[j + QK / 2]);
        }
        sumf += (fp16_to_fp32(x[i].d) * fp16_to_fp32(y[i].d)) * sumi;
    }
    return sumf;
}
static float vec_dot_q5_1_q8_1(int n, const block_q5_1 *x, const block_q8_1 *y) {
    const int nb = n / QK;
    float sumf = 0.0f;
    BLOCK_LOOP(i, nb) {
        uint32_t qh;
        memcpy(&qh, x[i].qh, sizeof(qh));
        int sumi = 0;
        for (int j = 0; j < QK / 2; ++j) {
            const uint8_t xh_0 = ((qh >> (j + 0)) << 4) & 0x10;
            const uint8_t xh_1 = ((qh >> (j + 12))) & 0x10;
            const int32_t x0 = (x[i].qs[j] & 0xF) | xh_0;
            const int32_t x1 = (x[i].qs[j] >> 4) | xh_1;
            sumi += (x0 * y[i].qs[j]) + (x1 * y[i].qs[j + QK / 2]);
        }
        sumf += (fp16_to_fp32(x[i].d) * y[i].d) * sumi + fp16_to_fp32(x[i].m) * y[i].s;
    }
    return sumf;
}
static float vec_dot_q8_0_q8_0(int n, const block_q8_0 *x, const block_q8_0 *y) {
    const int nb = n / QK;
    float sumf = 0.0f;
    BLOCK_LOOP(i, nb) {
        int sumi = 0;
        for (int j = 0; j < QK; j++) sumi += x[i].qs[j] * y[i].qs[j];
        sumf += sumi * (fp16_to_fp32(x[i].d) * fp16_to_fp32(y[i].d));
    }
    return sumf;
}
/* ---- the same dot products in the order of upstream's AVX2 branch (mode 2) ---------------------------------- */
static inline float hsum_float_8(const float *x) { /* upstream hsum_float_8 */
    const float r0 = x[4] + x[0], r1 = x[5] + x[1], r2 = x[6] + x[2], r3 = x[7] + x[3];
    const float s0 = r0 + r2, s1 = r1 + r3;
    return s0 + s1;
}
/* w[j], j = 0..31: the block's weights as integers in the order of y.qs (low nibbles 0..15, high nibbles 16..31) */
static inline void lanes_fma(const int *w, const int8_t *yq, float d, float *acc) {
    for (int l = 0; l < 8; l++) {
        int q = 0;
        for (int e = 0; e < 4; e++) q += w[4 * l + e] * yq[4 * l + e]; /* maddubs + madd: exact int */
        acc[l] = fmaf(d, (float)q, acc[l]);
    }
}
static float vec_dot_simd(int type, int n, const void *vx, const void *vy) {
    const int nb = n / QK;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float summs = 0.0f;
    int w[QK];
    BLOCK_LOOP(i, nb) { /* ascending like upstream; descending = the re-association yardstick (orc_set_block_order) */
        switch (type) {
            case T_Q4_0: {
                const block_q4_0 *x = (const block_q4_0 *)vx;
                const block_q8_0 *y = (const block_q8_0 *)vy;
                for (int j = 0; j < QK / 2; j++) {
                    w[j] = (x[i].qs[j] & 0x0F) - 8;
                    w[j + QK / 2] = (x[i].qs[j] >> 4) - 8;
                }
                lanes_fma(w, y[i].qs, fp16_to_fp32(x[i].d) * fp16_to_fp32(y[i].d), acc);
            } break;
            case T_Q4_1: {
                const block_q4_1 *x = (const block_q4_1 *)vx;
                const block_q8_1 *y = (const block_q8_1 *)vy;
                summs += fp16_to_fp32(x[i].m) * y[i].s;
                for (int j = 0; j < QK / 2; j++) {
                    w[j] = (x[i].qs[j] & 0x0F);
                    w[j + QK / 2] = (x[i].qs[j] >> 4);
                }
                lanes_fma(w, y[i].qs, fp16_to_fp32(x[i].d) * y[i].d, acc);
            } break;
            case T_Q5_0: {
                const block_q5_0 *x = (const block_q5_0 *)vx;
                const block_q8_0 *y = (const block_q8_0 *)vy;
                uint32_t qh;
                memcpy(&qh, x[i].qh, sizeof(qh));
                for (int j = 0; j < QK / 2; j++) {
                    w[j] = (int)((x[i].qs[j] & 0x0F) | (((qh >> j) & 1u) << 4)) - 16;
                    w[j + QK / 2] = (int)((x[i].qs[j] >> 4) | (((qh >> (j + 16)) & 1u) << 4)) - 16;
                }
                lanes_fma(w, y[i].qs, fp16_to_fp32(x[i].d) * fp16_to_fp32(y[i].d), acc);
            } break;
            case T_Q5_1: {
                const block_q5_1 *x = (const block_q5_1 *)vx;
                const block_q8_1 *y = (const block_q8_1 *)vy;
                uint32_t qh;
                memcpy(&qh, x[i].qh, sizeof(qh));
                summs += fp16_to_fp32(x[i].m) * y[i].s;
                for (int j = 0; j < QK / 2; j++) {
                    w[j] = (int)((x[i].qs[j] & 0x0F) | (((qh >> j) & 1u) << 4));
                    w[j + QK / 2] = (int)((x[i].qs[j] >> 4) | (((qh >> (j + 16)) & 1u) << 4));
                }
                lanes_fma(w, y[i].qs, fp16_to_fp32(x[i].d) * y[i].d, acc);
            } break;
            case T_Q8_0: {
                const block_q8_0 *x = (const block_q8_0 *)vx;
                const block_q8_0 *y = (const block_q8_0 *)vy;
                for (int j = 0; j < QK; j++) w[j] = x[i].qs[j];
                lanes_fma(w, y[i].qs, fp16_to_fp32(x[i].d) * fp16_to_fp32(y[i].d), acc);
            } break;
            default: fprintf(stderr, "vec_dot_simd: bad type %d\n", type); abort();
        }
    }
    return hsum_float_8(acc) + summs; /* summs == 0 for the Q8_0-activation types, x + 0.0f == x */
}
/* ggml_vec_dot_f16, GGML_SIMD branch with F16C/AVX2: GGML_F16_STEP = 32, four 8-lane f32 accumulators,
 * sum[j] = fma(ax[j], ay[j], sum[j]); GGML_F16_VEC_REDUCE (x0+x2, x1+x3, x0+x1, then lanes (l + l+4), hadd, hadd);
 * leftovers (n % 32) added in ggml_float (double). */
static float vec_dot_f16_simd(int64_t n, const float *x /* f16 values as f32 */, const float *y) {
    float sum[4][8];
    memset(sum, 0, sizeof(sum));
    const int64_t np = n & ~(int64_t)31;
    for (int64_t i = 0; i < np; i += 32)
        for (int j = 0; j < 4; j++)
            for (int l = 0; l < 8; l++) sum[j][l] = fmaf(x[i + j * 8 + l], y[i + j * 8 + l], sum[j][l]);
    float t[8];
    for (int l = 0; l < 8; l++) t[l] = (sum[0][l] + sum[2][l]) + (sum[1][l] + sum[3][l]);
    const float u0 = t[0] + t[4], u1 = t[1] + t[5], u2 = t[2] + t[6], u3 = t[3] + t[7];
    const float h0 = u0 + u1, h1 = u2 + u3; /* _mm_hadd_ps(t0, t0) */
    double sumf = (double)(h0 + h1);
    for (int64_t i = np; i < n; i++) sumf += (double)(x[i] * y[i]);
    return (float)sumf;
}

#ifdef ORC_HAVE_AVX2
/* ---- mode 3: upstream's AVX2 branch written with the intrinsics (helpers named as in ggml.c) ----------------------- */
static inline float hsum_float_8_avx(const __m256 x) {
    __m128 res = _mm256_extractf128_ps(x, 1);
    res = _mm_add_ps(res, _mm256_castps256_ps128(x));
    res = _mm_add_ps(res, _mm_movehl_ps(res, res));
    res = _mm_add_ss(res, _mm_movehdup_ps(res));
    return _mm_cvtss_f32(res);
}
static inline int hsum_i32_8_avx(const __m256i a) {
    const __m128i sum128 = _mm_add_epi32(_mm256_castsi256_si128(a), _mm256_extractf128_si256(a, 1));
    const __m128i hi64 = _mm_unpackhi_epi64(sum128, sum128);
    const __m128i sum64 = _mm_add_epi32(hi64, sum128);
    const __m128i hi32 = _mm_shuffle_epi32(sum64, _MM_SHUFFLE(2, 3, 0, 1));
    return _mm_cvtsi128_si32(_mm_add_epi32(sum64, hi32));
}
static inline __m256i bytes_from_nibbles_32(const uint8_t *rsi) {
    const __m128i tmp = _mm_loadu_si128((const __m128i *)rsi);
    const __m256i bytes = _mm256_set_m128i(_mm_srli_epi16(tmp, 4), tmp);
    return _mm256_and_si256(_mm256_set1_epi8(0xF), bytes);
}
static inline __m256i bytes_from_bits_32(const uint8_t *x) {
    uint32_t x32;
    memcpy(&x32, x, sizeof(uint32_t));
    const __m256i shuf_mask = _mm256_set_epi64x(0x0303030303030303, 0x0202020202020202, 0x0101010101010101, 0x0000000000000000);
    __m256i bytes = _mm256_shuffle_epi8(_mm256_set1_epi32((int)x32), shuf_mask);
    const __m256i bit_mask = _mm256_set1_epi64x(0x7fbfdfeff7fbfdfe);
    bytes = _mm256_or_si256(bytes, bit_mask);
    return _mm256_cmpeq_epi8(bytes, _mm256_set1_epi64x(-1));
}
static inline __m256 sum_i16_pairs_float(const __m256i x) {
    const __m256i summed_pairs = _mm256_madd_epi16(_mm256_set1_epi16(1), x);
    return _mm256_cvtepi32_ps(summed_pairs);
}
static inline __m256 mul_sum_us8_pairs_float(const __m256i ax, const __m256i sy) {
    return sum_i16_pairs_float(_mm256_maddubs_epi16(ax, sy));
}
static inline __m256 mul_sum_i8_pairs_float(const __m256i x, const __m256i y) {
    const __m256i ax = _mm256_sign_epi8(x, x);
    const __m256i sy = _mm256_sign_epi8(y, x);
    return mul_sum_us8_pairs_float(ax, sy);
}
/* GGML_FP16_TO_FP32 under -mf16c: _cvtsh_ss (same values as fp16_to_fp32: both are exact) */
static inline float f16c_to_f32(fp16_t h) { return _cvtsh_ss(h); }
static float vec_dot_avx2(int type, int n, const void *vx, const void *vy) {
    const int nb = n / QK;
    __m256 acc = _mm256_setzero_ps();
    float summs = 0.0f;
    switch (type) {
        case T_Q4_0: {
            const block_q4_0 *x = (const block_q4_0 *)vx;
            const block_q8_0 *y = (const block_q8_0 *)vy;
            for (int i = 0; i < nb; ++i) {
                const __m256 d = _mm256_set1_ps(f16c_to_f32(x[i].d) * f16c_to_f32(y[i].d));
                __m256i bx = bytes_from_nibbles_32(x[i].qs);
                bx = _mm256_sub_epi8(bx, _mm256_set1_epi8(8));
                const __m256i by = _mm256_loadu_si256((const __m256i *)y[i].qs);
                acc = _mm256_fmadd_ps(d, mul_sum_i8_pairs_float(bx, by), acc);
            }
        } break;
        case T_Q4_1: {
            const block_q4_1 *x = (const block_q4_1 *)vx;
            const block_q8_1 *y = (const block_q8_1 *)vy;
            for (int i = 0; i < nb; ++i) {
                const float d0 = f16c_to_f32(x[i].d), d1 = y[i].d;
                summs += f16c_to_f32(x[i].m) * y[i].s;
                const __m256 d0d1 = _mm256_mul_ps(_mm256_set1_ps(d0), _mm256_set1_ps(d1));
                const __m256i bx = bytes_from_nibbles_32(x[i].qs);
                const __m256i by = _mm256_loadu_si256((const __m256i *)y[i].qs);
                acc = _mm256_fmadd_ps(d0d1, mul_sum_us8_pairs_float(bx, by), acc);
            }
        } break;
        case T_Q5_0: {
            const block_q5_0 *x = (const block_q5_0 *)vx;
            const block_q8_0 *y = (const block_q8_0 *)vy;
            for (int i = 0; i < nb; ++i) {
                const __m256 d = _mm256_set1_ps(f16c_to_f32(x[i].d) * f16c_to_f32(y[i].d));
                __m256i bx = bytes_from_nibbles_32(x[i].qs);
                __m256i bxhi = bytes_from_bits_32(x[i].qh);
                bxhi = _mm256_andnot_si256(bxhi, _mm256_set1_epi8((char)0xF0));
                bx = _mm256_or_si256(bx, bxhi);
                const __m256i by = _mm256_loadu_si256((const __m256i *)y[i].qs);
                acc = _mm256_fmadd_ps(d, mul_sum_i8_pairs_float(bx, by), acc);
            }
        } break;
        case T_Q5_1: {
            const block_q5_1 *x = (const block_q5_1 *)vx;
            const block_q8_1 *y = (const block_q8_1 *)vy;
            for (int i = 0; i < nb; ++i) {
                const __m256 dx = _mm256_set1_ps(f16c_to_f32(x[i].d));
                summs += f16c_to_f32(x[i].m) * y[i].s;
                __m256i bx = bytes_from_nibbles_32(x[i].qs);
                __m256i bxhi = bytes_from_bits_32(x[i].qh);
                bxhi = _mm256_and_si256(bxhi, _mm256_set1_epi8(0x10));
                bx = _mm256_or_si256(bx, bxhi);
                const __m256 dy = _mm256_set1_ps(y[i].d);
                const __m256i by = _mm256_loadu_si256((const __m256i *)y[i].qs);
                acc = _mm256_fmadd_ps(mul_sum_us8_pairs_float(bx, by), _mm256_mul_ps(dx, dy), acc);
            }
        } break;
        case T_Q8_0: {
            const block_q8_0 *x = (const block_q8_0 *)vx;
            const block_q8_0 *y = (const block_q8_0 *)vy;
            for (int i = 0; i < nb; ++i) {
                const __m256 d = _mm256_set1_ps(f16c_to_f32(x[i].d) * f16c_to_f32(y[i].d));
                const __m256i bx = _mm256_loadu_si256((const __m256i *)x[i].qs);
                const __m256i by = _mm256_loadu_si256((const __m256i *)y[i].qs);
                acc = _mm256_fmadd_ps(d, mul_sum_i8_pairs_float(bx, by), acc);
            }
        } break;
        default: fprintf(stderr, "vec_dot_avx2: bad type %d\n", type); abort();
    }
    return hsum_float_8_avx(acc) + summs;
}
/* quantize_row_q8_0 / q8_1, AVX2 branch */
static void quantize_row_q8_avx2_intr(const float *x, void *vy, int k, int q81) {
    const int nb = k / QK;
    block_q8_0 *y0 = (block_q8_0 *)vy;
    block_q8_1 *y1 = (block_q8_1 *)vy;
    for (int i = 0; i < nb; i++) {
        __m256 v0 = _mm256_loadu_ps(x), v1 = _mm256_loadu_ps(x + 8), v2 = _mm256_loadu_ps(x + 16), v3 = _mm256_loadu_ps(x + 24);
        x += 32;
        const __m256 signBit = _mm256_set1_ps(-0.0f);
        __m256 maxAbs = _mm256_andnot_ps(signBit, v0);
        maxAbs = _mm256_max_ps(maxAbs, _mm256_andnot_ps(signBit, v1));
        maxAbs = _mm256_max_ps(maxAbs, _mm256_andnot_ps(signBit, v2));
        maxAbs = _mm256_max_ps(maxAbs, _mm256_andnot_ps(signBit, v3));
        __m128 max4 = _mm_max_ps(_mm256_extractf128_ps(maxAbs, 1), _mm256_castps256_ps128(maxAbs));
        max4 = _mm_max_ps(max4, _mm_movehl_ps(max4, max4));
        max4 = _mm_max_ss(max4, _mm_movehdup_ps(max4));
        const float maxScalar = _mm_cvtss_f32(max4);
        const float d = maxScalar / 127.f;
        const float id = (maxScalar != 0.0f) ? 127.f / maxScalar : 0.0f;
        const __m256 mul = _mm256_set1_ps(id);
        v0 = _mm256_round_ps(_mm256_mul_ps(v0, mul), _MM_ROUND_NEAREST);
        v1 = _mm256_round_ps(_mm256_mul_ps(v1, mul), _MM_ROUND_NEAREST);
        v2 = _mm256_round_ps(_mm256_mul_ps(v2, mul), _MM_ROUND_NEAREST);
        v3 = _mm256_round_ps(_mm256_mul_ps(v3, mul), _MM_ROUND_NEAREST);
        __m256i i0 = _mm256_cvtps_epi32(v0), i1 = _mm256_cvtps_epi32(v1), i2 = _mm256_cvtps_epi32(v2), i3 = _mm256_cvtps_epi32(v3);
        if (q81) {
            y1[i].d = d;
            y1[i].s = d * (float)hsum_i32_8_avx(_mm256_add_epi32(_mm256_add_epi32(i0, i1), _mm256_add_epi32(i2, i3)));
        } else {
            y0[i].d = fp32_to_fp16(d);
        }
        i0 = _mm256_packs_epi32(i0, i1);
        i2 = _mm256_packs_epi32(i2, i3);
        i0 = _mm256_packs_epi16(i0, i2);
        const __m256i perm = _mm256_setr_epi32(0, 4, 1, 5, 2, 6, 3, 7);
        i0 = _mm256_permutevar8x32_epi32(i0, perm);
        _mm256_storeu_si256((__m256i *)(q81 ? y1[i].qs : y0[i].qs), i0);
    }
}
/* ggml_vec_dot_f16, GGML_SIMD branch (F16C): both operands are fp16 arrays */
static float vec_dot_f16_avx2(int64_t n, const fp16_t *x, const fp16_t *y) {
    __m256 sum[4] = {_mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps()};
    const int64_t np = n & ~(int64_t)31;
    for (int64_t i = 0; i < np; i += 32)
        for (int j = 0; j < 4; j++) {
            const __m256 ax = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *)(x + i + j * 8)));
            const __m256 ay = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *)(y + i + j * 8)));
            sum[j] = _mm256_fmadd_ps(ax, ay, sum[j]);
        }
    sum[0] = _mm256_add_ps(sum[0], sum[2]);
    sum[1] = _mm256_add_ps(sum[1], sum[3]);
    sum[0] = _mm256_add_ps(sum[0], sum[1]);
    const __m128 t0 = _mm_add_ps(_mm256_castps256_ps128(sum[0]), _mm256_extractf128_ps(sum[0], 1));
    const __m128 t1 = _mm_hadd_ps(t0, t0);
    double sumf = (double)_mm_cvtss_f32(_mm_hadd_ps(t1, t1));
    for (int64_t i = np; i < n; i++) sumf += (double)(fp16_to_fp32(x[i]) * fp16_to_fp32(y[i]));
    return (float)sumf;
}
#endif /* ORC_HAVE_AVX2 */
EXPORT int orc_have_avx2(void) {
#ifdef ORC_HAVE_AVX2
    return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma") && __builtin_cpu_supports("f16c");
#else
    return 0;
#endif
}
EXPORT float orc_vec_dot_simd(int type, int n, const void *x, const void *y) { return vec_dot_simd(type, n, x, y); }
EXPORT float orc_vec_dot_avx2(int type, int n, const void *x, const void *y) {
#ifdef ORC_HAVE_AVX2
    return vec_dot_avx2(type, n, x, y);
#else
    return vec_dot_simd(type, n, x, y);
#endif
}
EXPORT void orc_quantize_row(int type, const float *x, void *y, int k);
EXPORT void orc_quantize_row_simd(int type, const float *x, void *y, int k) {
    if (type == T_Q8_0) quantize_row_q8_0_avx2(x, (block_q8_0 *)y, k);
    else if (type == T_Q8_1) quantize_row_q8_1_avx2(x, (block_q8_1 *)y, k);
    else orc_quantize_row(type, x, y, k);
}
EXPORT void orc_quantize_row_avx2(int type, const float *x, void *y, int k) {
#ifdef ORC_HAVE_AVX2
    if (type == T_Q8_0 || type == T_Q8_1) {
        quantize_row_q8_avx2_intr(x, y, k, type == T_Q8_1);
        return;
    }
#endif
    orc_quantize_row_simd(type, x, y, k);
}

EXPORT float orc_vec_dot(int type, int n, const void *x, const void *y) {
    switch (type) {
        case T_Q4_0: return vec_dot_q4_0_q8_0(n, (const block_q4_0 *)x, (const block_q8_0 *)y);
        case T_Q4_1: return vec_dot_q4_1_q8_1(n, (const block_q4_1 *)x, (const block_q8_1 *)y);
        case T_Q5_0: return vec_dot_q5_0_q8_0(n, (const block_q5_0 *)x, (const block_q8_0 *)y);
        case T_Q5_1: return vec_dot_q5_1_q8_1(n, (const block_q5_1 *)x, (const block_q8_1 *)y);
        case T_Q8_0: return vec_dot_q8_0_q8_0(n, (const block_q8_0 *)x, (const block_q8_0 *)y);
        case T_Q2_K: return vec_dot_q2_K_q8_K(n, (const block_q2_K *)x, (const block_q8_K *)y);
        case T_Q3_K: return vec_dot_q3_K_q8_K(n, (const block_q3_K *)x, (const block_q8_K *)y);
        case T_Q5_K: return vec_dot_q5_K_q8_K(n, (const block_q5_K *)x, (const block_q8_K *)y);
        case T_Q4_K: return vec_dot_q4_K_q8_K(n, (const block_q4_K *)x, (const block_q8_K *)y);
        case T_Q6_K: return vec_dot_q6_K_q8_K(n, (const block_q6_K *)x, (const block_q8_K *)y);
    }
    fprintf(stderr, "orc_vec_dot: bad type %d\n", type);
    abort();
}

/* ---- mul_mat: dst[n][m] = sum_k A[m][k] * B[n][k]   (crates/ggml/src/context.rs:314-324 doc;
 * call sites models/llama/src/lib.rs:194,208,223,310,323,325,332,352).
 * A: M rows of K (type), contiguous rows.  B: N rows of K f32 (row stride ldb floats).
 * dst: N rows of M f32. -------------------------------------------------------------------------- */
EXPORT void orc_mul_mat(int type, const void *A, int64_t M, int64_t K, const float *B, int64_t N, int64_t ldb,
                        float *dst, int mode) {
    const size_t row_bytes = (size_t)(K / orc_blck_size(type)) * (size_t)orc_type_size(type);
    if (mode != 1 && type != T_F32) {
        const int simd = (mode == 2 || mode == 3) && type != T_F16 && !(type >= T_Q2_K && type <= T_Q6_K);
        const int intr = simd && mode == 3;
        const int vdt = orc_vec_dot_type(type);
        const size_t qrow = (size_t)(K / orc_blck_size(vdt)) * (size_t)orc_type_size(vdt);
        uint8_t *wdata = (uint8_t *)malloc(qrow * (size_t)N); /* ggml: cplan.work_data, INIT phase */
        for (int64_t n = 0; n < N; n++)
            (intr ? orc_quantize_row_avx2 : simd ? orc_quantize_row_simd : orc_quantize_row)(vdt, B + n * ldb, wdata + (size_t)n * qrow, (int)K);
#pragma omp parallel for schedule(static)
        for (int64_t m = 0; m < M; m++) {
            const uint8_t *a = (const uint8_t *)A + (size_t)m * row_bytes;
            for (int64_t n = 0; n < N; n++) {
                const uint8_t *b = wdata + (size_t)n * qrow;
                float r;
                if (type == T_F16) {
                    /* ggml_vec_dot_f16, scalar branch: f32 products, ggml_float (double) sum */
                    const fp16_t *x = (const fp16_t *)a, *y = (const fp16_t *)b;
                    double s = 0.0;
                    for (int64_t k = 0; k < K; k++) s += (double)(fp16_to_fp32(x[k]) * fp16_to_fp32(y[k]));
                    r = (float)s;
                } else {
                    /* the intrinsics walk the blocks upwards only: the reversed-order yardstick takes their scalar restatement */
                    r = intr && !g_rev ? orc_vec_dot_avx2(type, (int)K, a, b) : simd ? vec_dot_simd(type, (int)K, a, b) : orc_vec_dot(type, (int)K, a, b);
                }
                dst[n * M + m] = r;
            }
        }
        free(wdata);
        return;
    }
    /* math mode (and f32 weights): dequantize the row, accumulate in f64 */
#pragma omp parallel
    {
        float *arow = (float *)malloc((size_t)K * 4);
#pragma omp for schedule(static)
        for (int64_t m = 0; m < M; m++) {
            orc_dequantize_row(type, (const uint8_t *)A + (size_t)m * row_bytes, arow, (int)K);
            for (int64_t n = 0; n < N; n++) {
                const float *b = B + n * ldb;
                double s = 0.0;
                for (int64_t k = 0; k < K; k++) s += (double)arow[k] * (double)b[k];
                dst[n * M + m] = (float)s;
            }
        }
        free(arow);
    }
}

/* ---- small ops -------------------------------------------------------------------------------- */
/* ggml_compute_forward_rms_norm_f32 (upstream): models/llama/src/lib.rs:183,318,343; eps from
 * crates/ggml/src/lib.rs:131-132.  x,y: nrows rows of ne0 floats. */
EXPORT void orc_rms_norm(const float *x, float *y, int64_t ne0, int64_t nrows, float eps) {
    for (int64_t r = 0; r < nrows; r++) {
        const float *xr = x + r * ne0;
        float *yr = y + r * ne0;
        double sum = 0.0;
        for (int64_t i = 0; i < ne0; i++) sum += (double)(xr[i] * xr[i]);
        const float mean = (float)(sum / (double)ne0);
        const float scale = 1.0f / sqrtf(mean + eps);
        for (int64_t i = 0; i < ne0; i++) yr[i] = xr[i] * scale;
    }
}
/* ggml_mul with src1 broadcast along rows (models/llama/src/lib.rs:186,321,346) */
EXPORT void orc_mul_rows(const float *x, const float *w, float *y, int64_t ne0, int64_t nrows) {
    for (int64_t r = 0; r < nrows; r++)
        for (int64_t i = 0; i < ne0; i++) y[r * ne0 + i] = x[r * ne0 + i] * w[i];
}
EXPORT void orc_add(const float *a, const float *b, float *y, int64_t n) {
    for (int64_t i = 0; i < n; i++) y[i] = a[i] + b[i];
}
EXPORT void orc_mul(const float *a, const float *b, float *y, int64_t n) {
    for (int64_t i = 0; i < n; i++) y[i] = a[i] * b[i];
}
/* ggml_vec_silu_f32 with GGML_SILU_FP16 (upstream default): silu through the f16 table,
 * table_silu_f16[i] = f16(silu_f32(f32(i))), silu_f32(x) = x/(1+expf(-x)).  models/llama:328 */
EXPORT void orc_silu(const float *x, float *y, int64_t n, int mode) {
    for (int64_t i = 0; i < n; i++) {
        if (mode != 1) {
            const float xf = fp16_to_fp32(fp32_to_fp16(x[i]));
            y[i] = fp16_to_fp32(fp32_to_fp16(xf / (1.0f + expf(-xf))));
        } else {
            y[i] = (float)((double)x[i] / (1.0 + exp(-(double)x[i])));
        }
    }
}
/* ggml_compute_forward_gelu_f32 with GGML_GELU_FP16 (upstream default): table_gelu_f16, tanh form.
 * Used only by the GPT-2 plumbing config (models/gpt2/src/lib.rs). */
static inline float gelu_f32(float x) {
    const float GELU_COEF_A = 0.044715f, SQRT_2_OVER_PI = 0.79788456080286535587989211986876f;
    return 0.5f * x * (1.0f + tanhf(SQRT_2_OVER_PI * x * (1.0f + GELU_COEF_A * x * x)));
}
EXPORT void orc_gelu(const float *x, float *y, int64_t n, int mode) {
    for (int64_t i = 0; i < n; i++) {
        if (mode != 1) {
            const float xf = fp16_to_fp32(fp32_to_fp16(x[i]));
            y[i] = fp16_to_fp32(fp32_to_fp16(gelu_f32(xf)));
        } else {
            y[i] = gelu_f32(x[i]);
        }
    }
}
/* ggml_compute_forward_norm_f32 (LayerNorm without affine; eps 1e-5 hard-coded upstream in that
 * window).  GPT-2 plumbing only. */
EXPORT void orc_norm(const float *x, float *y, int64_t ne0, int64_t nrows) {
    const float eps = 1e-5f;
    for (int64_t r = 0; r < nrows; r++) {
        const float *xr = x + r * ne0;
        float *yr = y + r * ne0;
        double sum = 0.0;
        for (int64_t i = 0; i < ne0; i++) sum += (double)xr[i];
        const float mean = (float)(sum / (double)ne0);
        double sum2 = 0.0;
        for (int64_t i = 0; i < ne0; i++) {
            const float v = xr[i] - mean;
            yr[i] = v;
            sum2 += (double)(v * v);
        }
        const float variance = (float)(sum2 / (double)ne0);
        const float scale = 1.0f / sqrtf(variance + eps);
        for (int64_t i = 0; i < ne0; i++) yr[i] *= scale;
    }
}
/* ggml_compute_forward_rope_f32, mode 0 (adjacent pairs), upstream window: theta is an iterated f32
 * product over the WHOLE row (ne0), n_dims only sets theta_scale.  x: [ne0, n_head, N] contiguous,
 * in place.  models/llama/src/lib.rs:191-218; builder crates/ggml/src/context.rs:557-590. */
EXPORT void orc_rope(float *x, int64_t ne0, int64_t n_head, int64_t N, int n_past, int n_dims, float freq_base,
                     float freq_scale) {
    const float theta_scale = powf(freq_base, -2.0f / n_dims);
    for (int64_t i2 = 0; i2 < N; i2++) {
        const int64_t p = n_past + i2;
        for (int64_t i1 = 0; i1 < n_head; i1++) {
            float theta = freq_scale * (float)p;
            float *row = x + (i2 * n_head + i1) * ne0;
            for (int64_t i0 = 0; i0 < ne0; i0 += 2) {
                const float cos_theta = cosf(theta);
                const float sin_theta = sinf(theta);
                theta *= theta_scale;
                const float x0 = row[i0], x1 = row[i0 + 1];
                row[i0] = x0 * cos_theta - x1 * sin_theta;
                row[i0 + 1] = x0 * sin_theta + x1 * cos_theta;
            }
        }
    }
}
/* scale → diag_mask_inf(n_past) → soft_max on rows (models/llama/src/lib.rs:268-281).
 * x: [nc, nr_per_head(N), n_head]; row j of a head masks columns i > n_past + j.
 * soft_max (upstream): max; exp through table_exp_f16 on f16(x-max); ggml_float sum; scale by 1/sum. */
EXPORT void orc_scale_mask_softmax(float *x, int64_t nc, int64_t N, int64_t n_head, float scale, int n_past,
                                   int mode) {
    for (int64_t h = 0; h < n_head; h++) {
        for (int64_t j = 0; j < N; j++) {
            float *row = x + (h * N + j) * nc;
            for (int64_t i = 0; i < nc; i++) {
                row[i] = row[i] * scale;
                if (i > n_past + j) row[i] = -INFINITY;
            }
            float max = -INFINITY;
            for (int64_t i = 0; i < nc; i++) max = row[i] > max ? row[i] : max;
            double sum = 0.0;
            for (int64_t i = 0; i < nc; i++) {
                if (row[i] == -INFINITY) {
                    row[i] = 0.0f;
                } else if (mode != 1) {
                    const fp16_t s = fp32_to_fp16(row[i] - max);
                    const float val = fp16_to_fp32(fp32_to_fp16(expf(fp16_to_fp32(s))));
                    sum += (double)val;
                    row[i] = val;
                } else {
                    const float val = (float)exp((double)row[i] - (double)max);
                    sum += (double)val;
                    row[i] = val;
                }
            }
            sum = 1.0 / sum;
            for (int64_t i = 0; i < nc; i++) row[i] *= (float)sum;
        }
    }
}
/* soft_max alone (rows of nc) */
EXPORT void orc_soft_max(float *x, int64_t nc, int64_t nrows, int mode) {
    for (int64_t r = 0; r < nrows; r++) {
        float *row = x + r * nc;
        float max = -INFINITY;
        for (int64_t i = 0; i < nc; i++) max = row[i] > max ? row[i] : max;
        double sum = 0.0;
        for (int64_t i = 0; i < nc; i++) {
            if (row[i] == -INFINITY) {
                row[i] = 0.0f;
            } else if (mode != 1) {
                const fp16_t s = fp32_to_fp16(row[i] - max);
                const float val = fp16_to_fp32(fp32_to_fp16(expf(fp16_to_fp32(s))));
                sum += (double)val;
                row[i] = val;
            } else {
                const float val = (float)exp((double)row[i] - (double)max);
                sum += (double)val;
                row[i] = val;
            }
        }
        sum = 1.0 / sum;
        for (int64_t i = 0; i < nc; i++) row[i] *= (float)sum;
    }
}

/* ---- whole LLaMA forward, node for node as crates/models/llama/src/lib.rs:166-362 ------------- */
typedef struct {
    int32_t n_vocab, n_embd, n_head, n_head_kv, n_layer, n_rot, n_ff, n_ctx;
    int32_t wtype;       /* ggml type of the 2-D weights */
    float rms_eps;       /* LLAMA_DEFAULT_RMS_EPS */
    float freq_base, freq_scale;
    const void *tok_embeddings; /* [n_embd, n_vocab] wtype */
    const float *norm;          /* [n_embd] f32 */
    const void *output;         /* [n_embd, n_vocab] wtype */
    /* per layer arrays of pointers */
    const float **attention_norm;
    const void **wq, **wk, **wv, **wo;
    const float **ffn_norm;
    const void **w1, **w2, **w3;
    /* session state: f16 KV, layouts of inference_session.rs:155-160 + llama lib.rs:228-244:
     * K element (layer il, pos p, chan c) at (il*n_ctx + p)*E_gqa + c ;
     * V element (layer il, chan c, pos p) at il*n_ctx*E_gqa + c*n_ctx + p  (transposed) */
    fp16_t *memory_k, *memory_v;
} orc_llama;

/* Optional taps on interior nodes for tensor-by-tensor parity tests. */
typedef struct {
    float *inpL0;       /* get_rows output [E,N] */
    float *layer0_attn_norm; /* after rms_norm*weight, layer 0 [E,N] */
    float *layer0_q;    /* Qcur after rope, layer 0 [D,H,N] */
    float *layer0_kq;   /* KQ_soft_max, layer 0 [P+N, N, H] */
    float *layer0_out;  /* layer 0 output residual [E,N] */
    float *final_norm;  /* embedding_result [E,N] */
    float *layer_out_all; /* output residual of EVERY layer [L][N][E]: layer il's input is inpL0 (il = 0) or slab il-1 —
                             the inputs and expected outputs of a teacher-forced per-layer check */
    const float *inp_override; /* if set: replaces get_rows' output [E,N] (a layer fed with a given residual) */
} orc_taps;

EXPORT void orc_llama_eval(const orc_llama *m, const int32_t *tokens, int N, int n_past, float *logits /* [V,N] */,
                           int mode, const orc_taps *taps) {
    const int64_t E = m->n_embd, H = m->n_head, Hkv = m->n_head_kv, D = E / H, L = m->n_layer, F = m->n_ff;
    const int64_t V = m->n_vocab, C = m->n_ctx, Egqa = E / (H / Hkv), P = n_past, T = P + N;
    const size_t erow = (size_t)(E / orc_blck_size(m->wtype)) * (size_t)orc_type_size(m->wtype);

    float *inpL = (float *)malloc((size_t)E * N * 4);
    float *cur = (float *)malloc((size_t)E * N * 4);
    float *q = (float *)malloc((size_t)E * N * 4);
    float *k = (float *)malloc((size_t)Egqa * N * 4);
    float *v = (float *)malloc((size_t)Egqa * N * 4);
    float *kq = (float *)malloc((size_t)T * N * H * 4);
    float *kqv = (float *)malloc((size_t)E * N * 4);
    float *att = (float *)malloc((size_t)E * N * 4);
    float *inpFF = (float *)malloc((size_t)E * N * 4);
    float *t1 = (float *)malloc((size_t)F * N * 4);
    float *t3 = (float *)malloc((size_t)F * N * 4);

    /* get_rows(wte, embd): dequantize_row (llama lib.rs:170) */
    for (int n = 0; n < N; n++)
        orc_dequantize_row(m->wtype, (const uint8_t *)m->tok_embeddings + (size_t)tokens[n] * erow, inpL + (size_t)n * E,
                           (int)E);
    if (taps && taps->inp_override) memcpy(inpL, taps->inp_override, (size_t)E * N * 4);
    if (taps && taps->inpL0) memcpy(taps->inpL0, inpL, (size_t)E * N * 4);

    for (int64_t il = 0; il < L; il++) {
        /* attention norm (:183-186) */
        orc_rms_norm(inpL, cur, E, N, m->rms_eps);
        orc_mul_rows(cur, m->attention_norm[il], cur, E, N);
        if (il == 0 && taps && taps->layer0_attn_norm) memcpy(taps->layer0_attn_norm, cur, (size_t)E * N * 4);
        /* Q, K (+RoPE), V (:191-226) */
        orc_mul_mat(m->wtype, m->wq[il], E, E, cur, N, E, q, mode);
        orc_rope(q, D, H, N, n_past, m->n_rot, m->freq_base, m->freq_scale);
        orc_mul_mat(m->wtype, m->wk[il], Egqa, E, cur, N, E, k, mode);
        orc_rope(k, D, Hkv, N, n_past, m->n_rot, m->freq_base, m->freq_scale);
        orc_mul_mat(m->wtype, m->wv[il], Egqa, E, cur, N, E, v, mode);
        if (il == 0 && taps && taps->layer0_q) memcpy(taps->layer0_q, q, (size_t)E * N * 4);
        /* KV store: CPY f32 -> f16 (:228-244); V scatter-transposed */
        for (int n = 0; n < N; n++) {
            for (int64_t c = 0; c < Egqa; c++) {
                m->memory_k[((size_t)il * C + P + n) * Egqa + c] = fp32_to_fp16(k[(size_t)n * Egqa + c]);
                m->memory_v[(size_t)il * C * Egqa + (size_t)c * C + (P + n)] = fp32_to_fp16(v[(size_t)n * Egqa + c]);
            }
        }
        /* KQ = mul_mat(K f16 [D,T,Hkv], Q f32 [D,N,H]) -> [T,N,H] (:246-265) */
#pragma omp parallel for collapse(2) schedule(static)
        for (int64_t h = 0; h < H; h++) {
            for (int64_t n = 0; n < N; n++) {
                const int64_t hk = h / (H / Hkv);
                const float *qrow = q + ((size_t)n * H + h) * D;
                float qh[512]; /* D <= 512 */
                for (int64_t d = 0; d < D; d++) qh[d] = mode != 1 ? fp16_to_fp32(fp32_to_fp16(qrow[d])) : qrow[d];
#ifdef ORC_HAVE_AVX2
                fp16_t q16[512];
                if (mode == 3)
                    for (int64_t d = 0; d < D; d++) q16[d] = fp32_to_fp16(qrow[d]);
#endif
                for (int64_t t = 0; t < T; t++) {
                    const fp16_t *krow = m->memory_k + ((size_t)il * C + t) * Egqa + hk * D;
                    double s = 0.0;
#ifdef ORC_HAVE_AVX2
                    if (mode == 3) {
                        kq[((size_t)h * N + n) * T + t] = vec_dot_f16_avx2(D, krow, q16);
                        continue;
                    }
#endif
                    if (mode == 2 || mode == 3) {
                        float kf[512];
                        for (int64_t d = 0; d < D; d++) kf[d] = fp16_to_fp32(krow[d]);
                        s = (double)vec_dot_f16_simd(D, kf, qh);
                    } else if (mode != 1) {
                        for (int64_t d = 0; d < D; d++) s += (double)(fp16_to_fp32(krow[d]) * qh[d]);
                    } else {
                        for (int64_t d = 0; d < D; d++) s += (double)fp16_to_fp32(krow[d]) * (double)qh[d];
                    }
                    kq[((size_t)h * N + n) * T + t] = (float)s;
                }
            }
        }
        /* scale, mask, softmax (:268-281) */
        orc_scale_mask_softmax(kq, T, N, H, 1.0f / sqrtf((float)E / (float)H), n_past, mode);
        if (il == 0 && taps && taps->layer0_kq) memcpy(taps->layer0_kq, kq, (size_t)T * N * H * 4);
        /* KQV = mul_mat(V f16 [T,D,Hkv], probs [T,N,H]) -> [D,N,H]; permute -> [D,H,N]; cpy contiguous (:284-307) */
#pragma omp parallel for collapse(2) schedule(static)
        for (int64_t h = 0; h < H; h++) {
            for (int64_t n = 0; n < N; n++) {
                const int64_t hk = h / (H / Hkv);
                const float *prow = kq + ((size_t)h * N + n) * T;
#ifdef ORC_HAVE_AVX2
                fp16_t *p16 = NULL;
                if (mode == 3) {
                    p16 = (fp16_t *)malloc((size_t)T * 2 + 64);
                    for (int64_t t = 0; t < T; t++) p16[t] = fp32_to_fp16(prow[t]);
                }
#endif
                for (int64_t d = 0; d < D; d++) {
                    const fp16_t *vrow = m->memory_v + (size_t)il * C * Egqa + (size_t)(hk * D + d) * C;
                    double s = 0.0;
#ifdef ORC_HAVE_AVX2
                    if (mode == 3) {
                        kqv[((size_t)n * H + h) * D + d] = vec_dot_f16_avx2(T, vrow, p16);
                        continue;
                    }
#endif
                    if (mode == 2 || mode == 3) {
                        float *vf = (float *)malloc((size_t)T * 8), *pf = vf + T;
                        for (int64_t t = 0; t < T; t++) {
                            vf[t] = fp16_to_fp32(vrow[t]);
                            pf[t] = fp16_to_fp32(fp32_to_fp16(prow[t]));
                        }
                        s = (double)vec_dot_f16_simd(T, vf, pf);
                        free(vf);
                    } else if (mode != 1) {
                        for (int64_t t = 0; t < T; t++)
                            s += (double)(fp16_to_fp32(vrow[t]) * fp16_to_fp32(fp32_to_fp16(prow[t])));
                    } else {
                        for (int64_t t = 0; t < T; t++) s += (double)fp16_to_fp32(vrow[t]) * (double)prow[t];
                    }
                    kqv[((size_t)n * H + h) * D + d] = (float)s;
                }
#ifdef ORC_HAVE_AVX2
                free(p16);
#endif
            }
        }
        /* out proj + residual (:310-314) */
        orc_mul_mat(m->wtype, m->wo[il], E, E, kqv, N, E, att, mode);
        orc_add(att, inpL, inpFF, E * N);
        /* FFN (:318-334) */
        orc_rms_norm(inpFF, cur, E, N, m->rms_eps);
        orc_mul_rows(cur, m->ffn_norm[il], cur, E, N);
        orc_mul_mat(m->wtype, m->w3[il], F, E, cur, N, E, t3, mode);
        orc_mul_mat(m->wtype, m->w1[il], F, E, cur, N, E, t1, mode);
        orc_silu(t1, t1, F * N, mode);
        orc_mul(t1, t3, t1, F * N);
        orc_mul_mat(m->wtype, m->w2[il], E, F, t1, N, F, cur, mode);
        orc_add(cur, inpFF, inpL, E * N);
        if (il == 0 && taps && taps->layer0_out) memcpy(taps->layer0_out, inpL, (size_t)E * N * 4);
        if (taps && taps->layer_out_all) memcpy(taps->layer_out_all + (size_t)il * E * N, inpL, (size_t)E * N * 4);
    }
    /* final norm + lm_head (:343-352) */
    orc_rms_norm(inpL, cur, E, N, m->rms_eps);
    orc_mul_rows(cur, m->norm, cur, E, N);
    if (taps && taps->final_norm) memcpy(taps->final_norm, cur, (size_t)E * N * 4);
    orc_mul_mat(m->wtype, m->output, V, E, cur, N, E, logits, mode);

    free(inpL); free(cur); free(q); free(k); free(v); free(kq); free(kqv); free(att); free(inpFF); free(t1); free(t3);
}

/* bench.py's cpu_baseline leg only: a copy of a weight matrix whose pages are FIRST TOUCHED by the threads that will read them
 * (the static row partition of orc_mul_mat's row loop, same team size), so that on a multi-socket host every thread streams
 * its rows from its own NUMA node — what ggml's mmap'd weights settle into after the first tokens.  dst: untouched memory. */
EXPORT void orc_first_touch_copy(void *dst, const void *src, int64_t rows, int64_t row_bytes) {
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < rows; m++) memcpy((uint8_t *)dst + (size_t)m * (size_t)row_bytes, (const uint8_t *)src + (size_t)m * (size_t)row_bytes, (size_t)row_bytes);
}
/* bench.py's cpu_baseline leg only: pins the threads of the current OpenMP team to distinct CPUs spread evenly over the CPUs
 * the process may use (both sockets of a two-socket host), so that the first-touch placement above stays valid for the timed
 * tokens; orc_unpin_threads gives every team thread (the caller included) its original mask back.  Returns the CPUs found. */
static cpu_set_t g_orig_mask;
static int g_orig_saved = 0;
EXPORT int orc_pin_threads(void) {
    if (!g_orig_saved) {
        CPU_ZERO(&g_orig_mask);
        if (sched_getaffinity(0, sizeof g_orig_mask, &g_orig_mask) != 0) return 0;
        g_orig_saved = 1;
    }
    static int cpus[CPU_SETSIZE];
    int n = 0;
    for (int c = 0; c < CPU_SETSIZE; c++)
        if (CPU_ISSET(c, &g_orig_mask)) cpus[n++] = c;
    if (n == 0) return 0;
#pragma omp parallel
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
#else
        const int t = 0, T = 1;
#endif
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(cpus[(int)((int64_t)t * n / T)], &one);
        (void)sched_setaffinity(0, sizeof one, &one);
    }
    return n;
}
EXPORT void orc_unpin_threads(void) {
    if (!g_orig_saved) return;
#pragma omp parallel
    { (void)sched_setaffinity(0, sizeof g_orig_mask, &g_orig_mask); }
}
EXPORT int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
EXPORT void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
