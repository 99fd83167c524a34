// ops.h — the non-matvec kernels of the LLaMA graph (SURVEY.md §8a rows a3–a8), generic over ggml
// strides so that `ggml_graph_compute` can execute any graph the reference's builders produce.
// Each kernel restates the reference CPU semantics of its op (f64 row sums, f16-rounded exp / SiLU,
// iterated-product RoPE angles) so results track the reference, not just "the math".
#pragma once
#include "common.h"

// ---- RMSNorm (ggml_rms_norm: context.rs:295-300; llama lib.rs:183,318,343) -------------------------
// one workgroup per row; Σx² in f64 like ggml's ggml_float; y = x * 1/sqrtf(mean+eps).
// FUSE_W: additionally multiply by the broadcast weight row (the ggml_mul that always follows).
template <bool FUSE_W>
__global__ void __launch_bounds__(256) k_rms_norm(const TView x, const TView y, const float *__restrict__ w,
                                                  float eps) {
    __shared__ double s_part[4];
    const int64_t r = blockIdx.x;
    const int64_t i1 = r % x.ne[1], i2 = (r / x.ne[1]) % x.ne[2], i3 = r / (x.ne[1] * x.ne[2]);
    const float *xr = (const float *)(x.p + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
    float *yr = (float *)(y.p + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]);
    const int64_t n = x.ne[0];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float v = xr[i];
        s += (double)(v * v);
    }
    s = wave_sum_f64(s);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    const double tot = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    const float mean = (float)(tot / (double)n);
    const float scale = 1.0f / sqrtf(mean + eps);
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        float v = xr[i] * scale;
        if (FUSE_W) v = v * w[i];
        yr[i] = v;
    }
}

// ---- LayerNorm without affine (ggml_norm; GPT-2 plumbing config only) -------------------------------
__global__ void __launch_bounds__(256) k_norm(const TView x, const TView y, float eps) {
    __shared__ double s_part[4];
    __shared__ float s_mean;
    const int64_t r = blockIdx.x;
    const int64_t i1 = r % x.ne[1], i2 = (r / x.ne[1]) % x.ne[2], i3 = r / (x.ne[1] * x.ne[2]);
    const float *xr = (const float *)(x.p + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
    float *yr = (float *)(y.p + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]);
    const int64_t n = x.ne[0];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)xr[i];
    s = wave_sum_f64(s);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) s_mean = (float)(((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) / (double)n);
    __syncthreads();
    const float mean = s_mean;
    double s2 = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float v = xr[i] - mean;
        s2 += (double)(v * v);
    }
    s2 = wave_sum_f64(s2);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s2;
    __syncthreads();
    const float variance = (float)(((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) / (double)n);
    const float scale = 1.0f / sqrtf(variance + eps);
    for (int64_t i = threadIdx.x; i < n; i += 256) yr[i] = (xr[i] - mean) * scale;
}

// ---- broadcasting binary ops (ggml_add / ggml_mul / ggml_repeat) ------------------------------------
enum { BIN_ADD = 0, BIN_MUL = 1, BIN_REPEAT = 2 };
template <int OP>
__global__ void __launch_bounds__(256) k_bin(const TView a, const TView b, const TView d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t i0 = i % d.ne[0], i1 = (i / d.ne[0]) % d.ne[1], i2 = (i / (d.ne[0] * d.ne[1])) % d.ne[2],
                  i3 = i / (d.ne[0] * d.ne[1] * d.ne[2]);
    float *dp = (float *)(d.p + i0 * d.nb[0] + i1 * d.nb[1] + i2 * d.nb[2] + i3 * d.nb[3]);
    const float bv = *(const float *)(b.p + (i0 % b.ne[0]) * b.nb[0] + (i1 % b.ne[1]) * b.nb[1] +
                                      (i2 % b.ne[2]) * b.nb[2] + (i3 % b.ne[3]) * b.nb[3]);
    if (OP == BIN_REPEAT) {
        *dp = bv;
        return;
    }
    const float av = *(const float *)(a.p + i0 * a.nb[0] + i1 * a.nb[1] + i2 * a.nb[2] + i3 * a.nb[3]);
    *dp = OP == BIN_ADD ? av + bv : av * bv;
}

// same-shape contiguous operands (the residual adds and the gate multiply of a prompt batch: 8..22 MB per
// operand): float4 per thread, no index arithmetic
template <int OP>
__global__ void __launch_bounds__(256) k_bin4(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b, f32x4 *d,
                                              int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 x = a[i], y = b[i];
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = OP == BIN_ADD ? x[k] + y[k] : x[k] * y[k];
    d[i] = r;
}

// ---- unary ops through ggml's f16 lookup tables (ggml_silu / ggml_gelu) -----------------------------
// table_silu_f16[h] = f16(silu_f32(f32(h))), looked up at h = f16(x): silu(x) := f16(xf/(1+expf(-xf))).
enum { UN_SILU = 0, UN_GELU = 1 };
__device__ __forceinline__ float silu_table(float x) {
    const float xf = round_f16(x);
    return round_f16(xf / (1.0f + expf(-xf)));
}
__device__ __forceinline__ float gelu_table(float x) {
    const float xf = round_f16(x);
    const float g = 0.5f * xf * (1.0f + tanhf(0.79788456080286535587989211986876f * xf * (1.0f + 0.044715f * xf * xf)));
    return round_f16(g);
}
// MUL_B: fused `silu(a) * b` (the ggml_mul that follows silu in the FFN, llama lib.rs:328-330)
template <int OP, bool MUL_B>
__global__ void __launch_bounds__(256) k_unary(const float *a, const float *b, float *d,
                                               int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = OP == UN_SILU ? silu_table(a[i]) : gelu_table(a[i]);
    if (MUL_B) v = v * b[i];
    d[i] = v;
}

template <int OP, bool MUL_B>
__global__ void __launch_bounds__(256) k_unary4(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b, f32x4 *d,
                                                int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 x = a[i];
    f32x4 y = {1.0f, 1.0f, 1.0f, 1.0f};
    if (MUL_B) y = b[i];
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float v = OP == UN_SILU ? silu_table(x[k]) : gelu_table(x[k]);
        if (MUL_B) v = v * y[k];
        r[k] = v;
    }
    d[i] = r;
}

// ---- scale by a device-resident scalar (ggml_scale) --------------------------------------------------
__global__ void __launch_bounds__(256) k_scale(const float *a, const float *s, float *d,
                                               int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = a[i] * s[0];
}

// ---- causal mask (ggml_diag_mask_inf): x [nc, nr, nz] contiguous; col i > n_past + row j → -inf ------
__global__ void __launch_bounds__(256) k_diag_mask_inf(const float *a, float *d, int64_t nc, int64_t nr,
                                                       int64_t n, int n_past) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t c = i % nc, j = (i / nc) % nr;
    d[i] = c > n_past + j ? -INFINITY : a[i];
}

// ---- softmax over rows (ggml_soft_max): exp through f16 like table_exp_f16, f64 row sum --------------
// SCALE_MASK: fused scale → diag_mask_inf → soft_max (llama lib.rs:268-281); rows are [nc] with row
// index j = (row % nr) inside each head.
template <bool SCALE_MASK>
__global__ void __launch_bounds__(256) k_soft_max(const float *a, float *d, int64_t nc, int64_t nr,
                                                  const float *scale, int n_past) {
    __shared__ float s_max[4];
    __shared__ double s_sum[4];
    const int64_t row = blockIdx.x;
    const float *x = a + row * nc;
    float *y = d + row * nc;
    const float sc = SCALE_MASK ? scale[0] : 1.0f;
    const int64_t lim = SCALE_MASK ? (int64_t)n_past + (row % nr) : nc;  // columns > lim are masked
    float mx = -INFINITY;
    for (int64_t i = threadIdx.x; i < nc; i += 256) {
        float v = x[i];
        if (SCALE_MASK) v = i > lim ? -INFINITY : v * sc;
        mx = fmaxf(mx, v);
    }
    mx = wave_max_f32(mx);
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    double sum = 0.0;
    for (int64_t i = threadIdx.x; i < nc; i += 256) {
        float v = x[i];
        if (SCALE_MASK) v = i > lim ? -INFINITY : v * sc;
        float e = 0.0f;
        if (v != -INFINITY) {
            e = round_f16(expf(round_f16(v - mx)));
            sum += (double)e;
        }
        y[i] = e;
    }
    sum = wave_sum_f64(sum);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = sum;
    __syncthreads();
    const double tot = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    const float inv = (float)(1.0 / tot);
    for (int64_t i = threadIdx.x; i < nc; i += 256) y[i] *= inv;  // same thread wrote y[i]
}

// ---- RoPE mode 0 (ggml_rope_inplace / ggml_rope_custom_inplace: context.rs:557-590) ------------------
// x [ne0, ne1(heads), ne2(tokens)]; pair (i0,i0+1) of token i2 rotated by theta_k = freq_scale*p*s^k with
// the iterated f32 product of the reference (k sequential multiplies by theta_scale, p = n_past+i2).
__global__ void __launch_bounds__(256) k_rope(const TView x, const TView y, int n_past, float theta_scale,
                                              float freq_scale, int mode) {
    const int64_t npairs = x.ne[0] / 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = npairs * x.ne[1] * x.ne[2] * x.ne[3];
    if (idx >= total) return;
    const int64_t k = idx % npairs, i1 = (idx / npairs) % x.ne[1], i2 = (idx / (npairs * x.ne[1])) % x.ne[2],
                  i3 = idx / (npairs * x.ne[1] * x.ne[2]);
    const int64_t p = (mode & 1) == 0 ? n_past + i2 : i2;
    float theta = freq_scale * (float)p;
    for (int64_t j = 0; j < k; j++) theta *= theta_scale;
    const float c = cosf(theta), s = sinf(theta);
    const float *src = (const float *)(x.p + (2 * k) * x.nb[0] + i1 * x.nb[1] + i2 * x.nb[2] + i3 * x.nb[3]);
    float *dst = (float *)(y.p + (2 * k) * y.nb[0] + i1 * y.nb[1] + i2 * y.nb[2] + i3 * y.nb[3]);
    const float x0 = src[0], x1 = *(const float *)((const char *)src + x.nb[0]);
    dst[0] = x0 * c - x1 * s;
    *(float *)((char *)dst + y.nb[0]) = x0 * s + x1 * c;
}

// ---- generic strided copy / convert (ggml_cpy, ggml_cont, ggml_dup): the KV-cache store -------------
// Element i of the flattened logical index space is read at src's strides and written at dst's strides
// (shapes may differ, element counts match) — covers K (contiguous f16 run) and the V scatter-transpose
// (llama lib.rs:228-244), the merge-heads copy (:302-307), and cont.
template <typename TS, typename TD>
__global__ void __launch_bounds__(256) k_cpy(const TView s, const TView d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t s0 = i % s.ne[0], s1 = (i / s.ne[0]) % s.ne[1], s2 = (i / (s.ne[0] * s.ne[1])) % s.ne[2],
                  s3 = i / (s.ne[0] * s.ne[1] * s.ne[2]);
    const int64_t d0 = i % d.ne[0], d1 = (i / d.ne[0]) % d.ne[1], d2 = (i / (d.ne[0] * d.ne[1])) % d.ne[2],
                  d3 = i / (d.ne[0] * d.ne[1] * d.ne[2]);
    const TS v = *(const TS *)(s.p + s0 * s.nb[0] + s1 * s.nb[1] + s2 * s.nb[2] + s3 * s.nb[3]);
    TD *o = (TD *)(d.p + d0 * d.nb[0] + d1 * d.nb[1] + d2 * d.nb[2] + d3 * d.nb[3]);
    if constexpr (sizeof(TS) == 4 && sizeof(TD) == 2)
        *o = __float2half_rn(v);
    else if constexpr (sizeof(TS) == 2 && sizeof(TD) == 4)
        *o = __half2float(v);
    else
        *o = v;
}

// ---- get_rows for f32 / f16 tables ------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_get_rows(const char *__restrict__ tab, int64_t nb1, const int *ids,
                                                  float *dst, int64_t ne0) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ne0) return;
    const T *row = (const T *)(tab + (int64_t)ids[blockIdx.y] * nb1);
    if constexpr (sizeof(T) == 2)
        dst[(int64_t)blockIdx.y * ne0 + i] = __half2float(row[i]);
    else
        dst[(int64_t)blockIdx.y * ne0 + i] = row[i];
}

// ---- F16 × F32 mat-mul with ggml strides: the attention products K·Q and V·P (llama lib.rs:265,296) --
// src0 f16 [K, M, ne02, ne03] (rows contiguous in K), src1 f32 [K, N, ne12, ne13], dst f32 [M, N, ne12, ne13].
// ggml converts the src1 row to f16 and accumulates f32 products; same here (lanes along K, shuffle reduce).
// grid = (ceil(M/4), N, ne12*ne13); one wave per (m, n, batch).
__global__ void __launch_bounds__(256) k_mul_mat_f16(const TView a, const TView b, const TView d) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + wave;
    if (m >= a.ne[1]) return;
    const int64_t n = blockIdx.y;
    const int64_t i12 = blockIdx.z % b.ne[2], i13 = blockIdx.z / b.ne[2];
    const int64_t i02 = i12 / (b.ne[2] / a.ne[2]), i03 = i13 / (b.ne[3] / a.ne[3]);  // broadcast (GQA)
    const __half *ar = (const __half *)(a.p + m * a.nb[1] + i02 * a.nb[2] + i03 * a.nb[3]);
    const float *br = (const float *)(b.p + n * b.nb[1] + i12 * b.nb[2] + i13 * b.nb[3]);
    const int64_t K = a.ne[0];
    float s = 0.0f;
    for (int64_t k = lane; k < K; k += 64) s += __half2float(ar[k]) * round_f16(br[k]);
    s = wave_sum_f32(s);
    if (lane == 0) *(float *)(d.p + m * d.nb[0] + n * d.nb[1] + i12 * d.nb[2] + i13 * d.nb[3]) = s;
}

// ---- F32 × F32 mat-mul (norm weights etc. never go here; kept for completeness of GPT-2 biasless paths)
__global__ void __launch_bounds__(256) k_mul_mat_f32(const TView a, const TView b, const TView d) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + wave;
    if (m >= a.ne[1]) return;
    const int64_t n = blockIdx.y;
    const int64_t i12 = blockIdx.z % b.ne[2], i13 = blockIdx.z / b.ne[2];
    const int64_t i02 = i12 / (b.ne[2] / a.ne[2]), i03 = i13 / (b.ne[3] / a.ne[3]);
    const float *ar = (const float *)(a.p + m * a.nb[1] + i02 * a.nb[2] + i03 * a.nb[3]);
    const float *br = (const float *)(b.p + n * b.nb[1] + i12 * b.nb[2] + i13 * b.nb[3]);
    const int64_t K = a.ne[0];
    float s = 0.0f;
    for (int64_t k = lane; k < K; k += 64) s += ar[k] * br[k];
    s = wave_sum_f32(s);
    if (lane == 0) *(float *)(d.p + m * d.nb[0] + n * d.nb[1] + i12 * d.nb[2] + i13 * d.nb[3]) = s;
}
