// prompt.h — the element-wise kernels of the fused PROMPT plan (llama_plan.inc plan_launch_prompt): a prompt batch of
// N >= 32 tokens through the LLaMA graph of crates/models/llama/src/lib.rs:166-362 with the quantized GEMMs on the
// matrix cores (mmq_dma.h) and everything between two GEMMs in ONE launch instead of 2-4 generic ones (ops.h):
//
//   k_p_norm_quant     [residual add ->] rms_norm -> x weight -> Q8 re-quantization -> f16(d*q) in the GEMM's k order
//                      (replaces k_bin4<ADD>, k_rms_norm<true>, k_quant_act_f16; lib.rs:183-186, 310-320, 343-347)
//   k_p_qkv_post       RoPE of Q in place, RoPE of K -> f16 -> memory_k, V -> f16 -> memory_v transposed
//                      (replaces 2 x k_rope, 2 x k_cpy<float, half>; lib.rs:191-244)
//   k_p_silu_mul_quant silu(w1 x) * (w3 x) -> Q8 re-quantization -> f16(d*q)   (k_unary4 + k_quant_act_f16; lib.rs:322-330)
//   k_p_soft_max       scale -> causal mask -> softmax of a row of scores, one wave per row (k_soft_max<true>; lib.rs:268-281)
//
// Every kernel performs the generic kernels' f32/f64 operations in the same order on the same values, so the plan's
// results are bit-identical to the node-by-node executor's (tests/test_prompt_plan_gpu.py compares logits and K/V).
#pragma once
#include "mmq.h"
#include "ops.h"

// the Q8_0 / Q8_1 re-quantization of one value inside its 32-wide block, as k_quant_act_f16: lanes 32k..32k+31 hold a block
template <bool F16_D>
__device__ __forceinline__ _Float16 p_requant(float v) {
    float amax = fabsf(v);
    amax = g32_max_f32(amax);
    float d = amax / 127.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    const int q = (int)roundf(v * id);
    if (F16_D) d = round_f16(d);
    float r = d * (float)q;
    r = fminf(fmaxf(r, -65504.0f), 65504.0f);
    return (_Float16)r;
}

// one 256-thread workgroup per token row.  ADD: xs = x + r first (written to xsum: the residual stream of the layer).
// E % 32 == 0 (a block never straddles two rows of threads).
template <bool F16_D, bool ADD>
__global__ void __launch_bounds__(256) k_p_norm_quant(const float *__restrict__ x, const float *__restrict__ r, float *xsum,
                                                      const float *__restrict__ w, float eps, int E, float *y_f32 /*nullable*/,
                                                      _Float16 *__restrict__ out) {
    __shared__ double s_part[4];
    const int64_t row = blockIdx.x;
    const float *xr = x + row * E;
    const float *src = xr;
    if constexpr (ADD) {
        const float *rr = r + row * E;
        float *xs = xsum + row * E;
        double s = 0.0;
        for (int i = threadIdx.x; i < E; i += 256) {
            const float v = xr[i] + rr[i];
            xs[i] = v;
            s += (double)(v * v);
        }
        s = wave_sum_f64(s);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
        src = xs;  // re-read below by the thread that wrote it
    } else {
        double s = 0.0;
        for (int i = threadIdx.x; i < E; i += 256) {
            const float v = xr[i];
            s += (double)(v * v);
        }
        s = wave_sum_f64(s);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    }
    __syncthreads();
    const double tot = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    const float mean = (float)(tot / (double)E);
    const float scale = 1.0f / sqrtf(mean + eps);
    _Float16 *orow = out + row * E;
    for (int i0 = 0; i0 < E; i0 += 256) {  // uniform trip count: whole blocks of 32 lanes take part in the reduction
        const int i = i0 + threadIdx.x;
        float v = 0.0f;
        if (i < E) {
            v = src[i] * scale;
            v = v * w[i];
            if (y_f32) y_f32[row * E + i] = v;
        }
        const _Float16 h = p_requant<F16_D>(v);
        if (i < E) orow[(i & ~31) + mmq_kperm_inv(i & 31)] = h;
    }
}

// silu(a) * b (ggml's f16-table SiLU) -> re-quantization; 32 lanes per block, n = number of elements (multiple of 32)
template <bool F16_D>
__global__ void __launch_bounds__(256) k_p_silu_mul_quant(const float *__restrict__ a, const float *__restrict__ b, int64_t n,
                                                          _Float16 *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float v = 0.0f;
    if (i < n) {
        v = silu_table(a[i]);
        v = v * b[i];
    }
    const _Float16 h = p_requant<F16_D>(v);
    if (i < n) out[(i & ~(int64_t)31) + mmq_kperm_inv((int)(i & 31))] = h;
}

// RoPE + K/V store of a prompt batch.  q [N][E] f32 is rotated in place; kf [N][Egqa], vf [N][Egqa] f32 are the wk / wv
// GEMM outputs; tab [N][128] the (cos, sin) pairs of k_rope_table; mem_k / mem_v the layer's cache (K: [C][Egqa] f16,
// V: [Egqa][C] f16).  Blocks [0, nb_rope): one thread per (token, pair) of Q and K.  Blocks [nb_rope, ...): 64 x 64
// (token x channel) tiles of V transposed through LDS so that both the reads and the writes are contiguous.
struct PQkvPost {
    float *q;
    const float *kf, *vf, *tab;
    __half *mem_k, *mem_v;
    int N, E, Egqa, D, n_past;
    int64_t C;
    int nb_rope, vt_n, vt_m;  // rope blocks; V tiles along tokens / channels
};
__global__ void __launch_bounds__(256) k_p_qkv_post(const PQkvPost a) {
    __shared__ float s_t[64][65];
    const int b = blockIdx.x;
    if (b < a.nb_rope) {
        const int per_tok = (a.E + a.Egqa) >> 1;  // pairs per token: Q then K
        const int64_t idx = (int64_t)b * 256 + threadIdx.x;
        if (idx >= (int64_t)a.N * per_tok) return;
        const int n = (int)(idx / per_tok), pr = (int)(idx - (int64_t)n * per_tok);
        const bool is_k = pr >= (a.E >> 1);
        const int pi = is_k ? pr - (a.E >> 1) : pr;  // pair index inside the row
        const int kk = pi % (a.D >> 1);
        const f32x2 cs = *(const f32x2 *)(a.tab + (int64_t)n * 128 + 2 * kk);
        const float c = cs[0], s = cs[1];
        if (!is_k) {
            float *p = a.q + (int64_t)n * a.E + 2 * pi;
            const f32x2 v = *(const f32x2 *)p;
            f32x2 o;
            o[0] = v[0] * c - v[1] * s;
            o[1] = v[0] * s + v[1] * c;
            *(f32x2 *)p = o;
        } else {
            const f32x2 v = *(const f32x2 *)(a.kf + (int64_t)n * a.Egqa + 2 * pi);
            const float o0 = v[0] * c - v[1] * s, o1 = v[0] * s + v[1] * c;
            __half2 h;
            h.x = __float2half_rn(o0);
            h.y = __float2half_rn(o1);
            *(__half2 *)(a.mem_k + ((int64_t)a.n_past + n) * a.Egqa + 2 * pi) = h;
        }
        return;
    }
    const int t = b - a.nb_rope, tn = t % a.vt_n, tm = t / a.vt_n;
    const int n0 = tn * 64, m0 = tm * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int n = n0 + ly + 4 * i, m = m0 + lx;
        s_t[ly + 4 * i][lx] = (n < a.N && m < a.Egqa) ? a.vf[(int64_t)n * a.Egqa + m] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int m = m0 + ly + 4 * i, n = n0 + lx;
        if (n < a.N && m < a.Egqa) a.mem_v[(int64_t)m * a.C + a.n_past + n] = __float2half_rn(s_t[lx][ly + 4 * i]);
    }
}

// scale -> causal mask -> softmax, one WAVE per row of scores (k_soft_max<true> spends a 256-thread workgroup and two
// barriers on a row of a few hundred values).  Same operations: v = x*scale, row max, e = f16(expf(f16(v - max))), the
// sum in f64, y = e * (float)(1/sum).  The f64 sum is order-sensitive only through rounding of a double accumulation of
// f16-valued terms (<= 2^15 terms of 11 significant bits each fit 53 bits exactly), so any order gives the same bits.
// x: [rows][nc] f32, in place; row r of a head has query index j = r % nr and sees columns <= n_past + j.
__global__ void __launch_bounds__(256) k_p_soft_max(float *x, int64_t rows, int nc, int nr, float scale, int n_past) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float *p = x + row * nc;
    const int lim = n_past + (int)(row % nr);  // columns > lim are masked
    float mx = -INFINITY;
    for (int i = lane; i < nc; i += 64) {
        const float v = i > lim ? -INFINITY : p[i] * scale;
        mx = fmaxf(mx, v);
    }
    mx = wave_max_f32(mx);
    double sum = 0.0;
    for (int i = lane; i < nc; i += 64) {
        const float v = i > lim ? -INFINITY : p[i] * scale;
        float e = 0.0f;
        if (v != -INFINITY) {
            e = round_f16(expf(round_f16(v - mx)));
            sum += (double)e;
        }
        p[i] = e;
    }
    sum = wave_sum_f64(sum);
    const float inv = (float)(1.0 / sum);
    for (int i = lane; i < nc; i += 64) p[i] *= inv;  // the same lane wrote p[i]
}
