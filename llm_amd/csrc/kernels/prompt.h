// prompt.h — the element-wise kernels of the fused PROMPT plan (llama_plan.inc plan_launch_prompt): a prompt batch of
// N >= 32 tokens through the LLaMA graph of crates/models/llama/src/lib.rs:166-362 with the quantized GEMMs on the
// matrix cores (mmq_w16.h / mmq_dmap8.h) and everything between two GEMMs in ONE launch instead of 2-4 generic ones (ops.h):
//
//   k_p_norm_quant     [residual add ->] rms_norm -> x weight -> Q8 re-quantization -> f16(d*q) in the GEMM's k order
//                      (replaces k_bin4<ADD>, k_rms_norm<true>, k_quant_act_f16; lib.rs:183-186, 310-320, 343-347)
//   k_p_qkv_post       RoPE of Q in place, RoPE of K -> f16 -> memory_k, V -> f16 -> memory_v transposed
//                      (replaces 2 x k_rope, 2 x k_cpy<float, half>; lib.rs:191-244)
//   k_p_silu_mul_quant silu(w1 x) * (w3 x) -> Q8 re-quantization -> f16(d*q)   (k_unary4 + k_quant_act_f16; lib.rs:322-330)
//   k_p_quant4         merged attention output -> Q8 re-quantization -> f16(d*q)  (k_quant_act_f16, four values per lane)
//   k_p_soft_max       scale -> causal mask -> softmax of a row of scores, one wave per row, probabilities written as the
//                      f16 the V.P product converts them to anyway (k_soft_max<true>; lib.rs:268-281)
// A GEMM that splits K hands over two partial tiles; the kernel consuming them adds the two (what the atomics of the
// node-by-node path add), so neither a memset nor atomics are needed.
//
// Every kernel performs the generic kernels' f32/f64 operations in the same order on the same values, so the plan's
// results are bit-identical to the node-by-node executor's (tests/test_prompt_plan_gpu.py compares logits and K/V).
#pragma once
#include "mmq.h"
#include "ops.h"

template <bool F16_D>
__device__ __forceinline__ void p_requant4_store(const f32x4 v, int64_t i4, bool valid, _Float16 *__restrict__ out);
__device__ __forceinline__ void p_requant4_store_k(const f32x4 v, int64_t i4, bool valid, _Float16 *__restrict__ out);
// K8: the operand of a K-quant weight's GEMM — the row after its Q8_K round trip (k_quant_act_f16_k) instead of Q8_0 / Q8_1
template <bool F16_D, bool K8>
__device__ __forceinline__ void p_requant4(const f32x4 v, int64_t i4, bool valid, _Float16 *__restrict__ out) {
    if constexpr (K8)
        p_requant4_store_k(v, i4, valid, out);
    else
        p_requant4_store<F16_D>(v, i4, valid, out);
}

// one 256-thread workgroup per token row.  ADD: xs = (x [+ x2]) + r first (written to xsum: the residual stream of the
// layer; x2 = the second partial of a K-split GEMM, only with ADD).
// E % 32 == 0 (a block never straddles two rows of threads).
// Round 6: ONE pass over HBM.  The row (E <= 8192: 8 x f32x4 per thread) is loaded with 16-byte loads into registers, summed with its
// residual there, and stays there for the second half; the sum of squares keeps k_rms_norm's order — thread t adds elements
// t, t + 256, ... in f64 — by reading the row back from an LDS copy in that order (bank-conflict-free), so the result is bit for bit
// what the two-pass version (4-byte strided loads, then a re-read of the sums through L2: 10.9 us for 28 MB at 7B) gave.
// Dynamic LDS: E floats.
template <bool F16_D, bool ADD, bool K8 = false>
__global__ void __launch_bounds__(256) k_p_norm_quant(const float *__restrict__ x, const float *__restrict__ x2 /*nullable*/,
                                                      const float *__restrict__ r, float *xsum, const float *__restrict__ w,
                                                      float eps, int E, float *y_f32 /*nullable*/, _Float16 *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float s_row[];
    __shared__ double s_part[4];
    constexpr int MAXU = 8;  // rows up to 8192 wide (launcher)
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x, E4 = E >> 2;
    const f32x4 *xr = (const f32x4 *)(x + row * E);
    const f32x4 *x2r = x2 ? (const f32x4 *)(x2 + row * E) : nullptr;  // second partial of a K-split GEMM: x = x + x2 (what the atomics computed)
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 v[MAXU], wv[MAXU];
#pragma unroll
    for (int u = 0; u < MAXU; u++) {
        const int j = u * 256 + tid;
        v[u] = j < E4 ? xr[j] : zero4;
    }
    if constexpr (ADD) {
        const f32x4 *rr = (const f32x4 *)(r + row * E);
        f32x4 t[MAXU];
        if (x2r) {  // uniform
#pragma unroll
            for (int u = 0; u < MAXU; u++) {
                const int j = u * 256 + tid;
                t[u] = j < E4 ? x2r[j] : zero4;
            }
#pragma unroll
            for (int u = 0; u < MAXU; u++) v[u] = v[u] + t[u];
        }
#pragma unroll
        for (int u = 0; u < MAXU; u++) {
            const int j = u * 256 + tid;
            t[u] = j < E4 ? rr[j] : zero4;
        }
#pragma unroll
        for (int u = 0; u < MAXU; u++) v[u] = v[u] + t[u];
        f32x4 *xs = (f32x4 *)(xsum + row * E);
#pragma unroll
        for (int u = 0; u < MAXU; u++) {
            const int j = u * 256 + tid;
            if (j < E4) xs[j] = v[u];  // the residual stream of the layer
        }
    }
    // the norm weights depend on nothing: requested here (behind the row's own loads in the wave's queue), they land during the
    // two barriers and the sum instead of costing a second round trip behind them
#pragma unroll
    for (int u = 0; u < MAXU; u++) {
        const int j = u * 256 + tid;
        wv[u] = j < E4 ? ((const f32x4 *)w)[j] : zero4;
    }
#pragma unroll
    for (int u = 0; u < MAXU; u++) {
        const int j = u * 256 + tid;
        if (j < E4) ((f32x4 *)s_row)[j] = v[u];
    }
    __syncthreads();
    double s = 0.0;
    for (int i = tid; i < E; i += 256) s += (double)(s_row[i] * s_row[i]);  // thread t: elements t, t + 256, ... ascending (k_rms_norm)
    s = wave_sum_f64(s);
    if ((tid & 63) == 0) s_part[tid >> 6] = s;
    __syncthreads();
    const double tot = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    const float mean = (float)(tot / (double)E);
    const float scale = 1.0f / sqrtf(mean + eps);
    // element-wise apart from the block maximum (order-free): four consecutive values per lane, a block = 8 lanes, so the maximum
    // needs three DPP steps instead of an LDS round trip per value
    _Float16 *orow = out + row * E;
#pragma unroll
    for (int u = 0; u < MAXU; u++) {
        if (u * 256 >= E4) break;  // uniform: whole 8-lane groups take part in the reduction
        const int j = u * 256 + tid;
        const f32x4 ww = wv[u];
        f32x4 y;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float t = v[u][k] * scale;
            t = t * ww[k];
            y[k] = t;
        }
        if (y_f32 && j < E4) ((f32x4 *)(y_f32 + row * E))[j] = y;
        p_requant4<F16_D, K8>(y, j, j < E4, orow);
    }
}

// Four consecutive values of a block per lane (a block = 8 lanes): the same re-quantization, the block maximum taken over
// the lane's four values and then over the 8 lanes.  Element 4*l8 + i of the block goes to position
// 8*(l8 & 3) + 4*(l8 >> 2) + {0, 2, 1, 3}[i] of the GEMM's k order (mmq_kperm_inv): four adjacent f16, one 8-byte store.
template <bool F16_D>
__device__ __forceinline__ void p_requant4_store(const f32x4 v, int64_t i4 /* index of the lane's 4-vector */, bool valid,
                                                 _Float16 *__restrict__ out) {
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = g8_max_f32(amax);
    float d = amax / 127.0f;
    const bool sc = aq_scalar();
    const float id = act_id(amax, d, sc);
    const float dq = F16_D ? round_f16(d) : d;  // id comes from the unrounded d, as in k_quant_act_f16
    _Float16 h[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int q = act_q(v[k] * id, sc);
        float r = dq * (float)q;
        r = fminf(fmaxf(r, -65504.0f), 65504.0f);
        h[k] = (_Float16)r;
    }
    if (valid) {
        const int l8 = (int)(i4 & 7);
        const int64_t base = (i4 >> 3) * 32 + 8 * (l8 & 3) + 4 * (l8 >> 2);
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        const f16x4 o = {h[0], h[2], h[1], h[3]};
        *(f16x4 *)(out + base) = o;
    }
}

// The same for a K-quant weight's GEMM: the Q8_K round trip of k_quant_act_f16_k (quantize_row_q8_K: the FIRST value of largest
// magnitude gives iscale = -128 / max, q = min(127, nearest_int(iscale x)), value = (1 / iscale) q).  A super-block of 256 values is
// the four values of each of a wave's 64 lanes (i4 = 64 x super-block + lane: rows are multiples of 256 wide and every caller
// indexes 4-vectors by 256 x block + thread), so the extreme and its index are wave reductions (kquant_big.h q8k_wave_block).
__device__ __forceinline__ void p_requant4_store_k(const f32x4 v, int64_t i4, bool valid, _Float16 *__restrict__ out) {
    const int lane = (int)(threadIdx.x & 63);
    const float a0 = fabsf(v[0]), a1 = fabsf(v[1]), a2 = fabsf(v[2]), a3 = fabsf(v[3]);
    const float am = wave_max_f32(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)));
    int idx = a0 == am ? 4 * lane : a1 == am ? 4 * lane + 1 : a2 == am ? 4 * lane + 2 : a3 == am ? 4 * lane + 3 : 256;
    idx = wave_min_i32(idx);  // wave-uniform: the first index holding the extreme
    const int k = idx & 3;
    const float cand = k == 0 ? v[0] : k == 1 ? v[1] : k == 2 ? v[2] : v[3];
    const float mx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cand), (idx >> 2) & 63));
    _Float16 h[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float r = 0.0f;
        if (am != 0.0f) {  // uniform
            const float iscale = -128.0f / mx;
            const int q = min(127, __float2int_rn(iscale * v[i]));
            r = (1.0f / iscale) * (float)q;
        }
        r = fminf(fmaxf(r, -65504.0f), 65504.0f);
        h[i] = (_Float16)r;
    }
    if (valid) {
        const int l8 = (int)(i4 & 7);
        const int64_t base = (i4 >> 3) * 32 + 8 * (l8 & 3) + 4 * (l8 >> 2);
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        const f16x4 o = {h[0], h[2], h[1], h[3]};
        *(f16x4 *)(out + base) = o;
    }
}

// f32 rows (contiguous, n4 4-vectors in total) -> re-quantized f16 operand: k_quant_act_f16 with four values per lane
template <bool F16_D, bool K8 = false>
__global__ void __launch_bounds__(256) k_p_quant4(const f32x4 *__restrict__ a, int64_t n4, _Float16 *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n4;
    const f32x4 v = valid ? a[i] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    p_requant4<F16_D, K8>(v, i, valid, out);
}

// silu(a) * b (ggml's f16-table SiLU) -> re-quantization; four values per lane, n4 = number of 4-vectors (rows are
// multiples of 32 wide, so a block never straddles rows)
template <bool F16_D, bool K8 = false>
__global__ void __launch_bounds__(256) k_p_silu_mul_quant(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b, int64_t n4,
                                                          int64_t part4 /* != 0: second partials this many 4-vectors on */,
                                                          _Float16 *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n4;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (valid) {
        f32x4 x = a[i], y = b[i];
        if (part4) {
            x = x + a[i + part4];
            y = y + b[i + part4];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float t = silu_table(x[k]);
            t = t * y[k];
            v[k] = t;
        }
    }
    p_requant4<F16_D, K8>(v, i, valid, out);
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
    int nb_rope, vt_n, vt_m;  // rope blocks (two pairs per thread); V tiles along tokens / channels
    int64_t part;             // != 0: q / kf / vf are the first partials of a K-split GEMM, the second ones lie `part` floats on
    int skip_q;               // != 0: Q is left as the GEMM wrote it — the fused attention kernel applies RoPE while loading it
};
__global__ void __launch_bounds__(256) k_p_qkv_post(const PQkvPost a) {
    __shared__ float s_t[64][65];
    const int b = blockIdx.x;
    if (b < a.nb_rope) {
        const int q4 = a.skip_q ? 0 : (a.E >> 2);
        const int per_tok = q4 + (a.Egqa >> 2);  // 4-vectors (two pairs) per token: Q then K; D/2 is even, so are E/2, Egqa/2
        const int64_t idx = (int64_t)b * 256 + threadIdx.x;
        if (idx >= (int64_t)a.N * per_tok) return;
        const int n = (int)(idx / per_tok), pr = (int)(idx - (int64_t)n * per_tok);
        const bool is_k = pr >= q4;
        const int pi = 2 * (is_k ? pr - q4 : pr);  // first of the two pairs, index inside the row
        const int kk = pi % (a.D >> 1);
        const f32x4 cs = *(const f32x4 *)(a.tab + (int64_t)n * 128 + 2 * kk);  // cos, sin of pairs kk, kk + 1
        if (!is_k) {
            float *p = a.q + (int64_t)n * a.E + 2 * pi;
            f32x4 v = *(const f32x4 *)p;
            if (a.part) v = v + *(const f32x4 *)(p + a.part);
            f32x4 o;
            o[0] = v[0] * cs[0] - v[1] * cs[1];
            o[1] = v[0] * cs[1] + v[1] * cs[0];
            o[2] = v[2] * cs[2] - v[3] * cs[3];
            o[3] = v[2] * cs[3] + v[3] * cs[2];
            *(f32x4 *)p = o;
        } else {
            f32x4 v = *(const f32x4 *)(a.kf + (int64_t)n * a.Egqa + 2 * pi);
            if (a.part) v = v + *(const f32x4 *)(a.kf + a.part + (int64_t)n * a.Egqa + 2 * pi);
            const float o0 = v[0] * cs[0] - v[1] * cs[1], o1 = v[0] * cs[1] + v[1] * cs[0];
            const float o2 = v[2] * cs[2] - v[3] * cs[3], o3 = v[2] * cs[3] + v[3] * cs[2];
            __half2 h0, h1;
            h0.x = __float2half_rn(o0);
            h0.y = __float2half_rn(o1);
            h1.x = __float2half_rn(o2);
            h1.y = __float2half_rn(o3);
            __half2 *dst = (__half2 *)(a.mem_k + ((int64_t)a.n_past + n) * a.Egqa + 2 * pi);
            dst[0] = h0;
            dst[1] = h1;
        }
        return;
    }
    // V: 64 tokens x 64 channels per workgroup through LDS.  In: four 16-byte loads per thread (four channels of one token; Egqa is a
    // multiple of 4, so a vector lies inside the row or outside it).  Out: a lane holds two neighbouring tokens of one channel and
    // stores them as one 4-byte pair when the cache position of the tile's first token is even (an odd n_past: one f16 at a time).
    const int t = b - a.nb_rope, tn = t % a.vt_n, tm = t / a.vt_n;
    const int n0 = tn * 64, m0 = tm * 64;
    {
        const int c4 = threadIdx.x & 15, r = threadIdx.x >> 4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int n = n0 + r + 16 * i, m = m0 + 4 * c4;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (n < a.N && m < a.Egqa) {
                v = *(const f32x4 *)(a.vf + (int64_t)n * a.Egqa + m);
                if (a.part) v = v + *(const f32x4 *)(a.vf + a.part + (int64_t)n * a.Egqa + m);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) s_t[r + 16 * i][4 * c4 + k] = v[k];
        }
    }
    __syncthreads();
    if ((((int64_t)a.n_past + n0) & 1) == 0 && (a.C & 1) == 0) {  // uniform (every channel's row of the cache starts 4-byte aligned)
        const int np = threadIdx.x & 31, mg = threadIdx.x >> 5;
        const int n = n0 + 2 * np;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int ml = mg + 8 * i, m = m0 + ml;
            if (m < a.Egqa && n < a.N) {
                __half *dst = a.mem_v + (int64_t)m * a.C + a.n_past + n;
                const __half h0 = __float2half_rn(s_t[2 * np][ml]);
                if (n + 1 < a.N) {
                    __half2 h2;
                    h2.x = h0;
                    h2.y = __float2half_rn(s_t[2 * np + 1][ml]);
                    *(__half2 *)dst = h2;
                } else {
                    *dst = h0;
                }
            }
        }
    } else {
        const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int m = m0 + ly + 4 * i, n = n0 + lx;
            if (n < a.N && m < a.Egqa) a.mem_v[(int64_t)m * a.C + a.n_past + n] = __float2half_rn(s_t[lx][ly + 4 * i]);
        }
    }
}

// scale -> causal mask -> softmax, one WAVE per row of scores (k_soft_max<true> spends a 256-thread workgroup and two
// barriers on a row of a few hundred values).  Same operations: v = x*scale, row max, e = f16(expf(f16(v - max))), the
// sum in f64, y = e * (float)(1/sum).  The f64 sum is order-sensitive only through rounding of a double accumulation of
// f16-valued terms (<= 2^15 terms of 11 significant bits each fit 53 bits exactly), so any order gives the same bits.
// x: [rows][nc] f32, in place; row r of a head has query index j = r % nr and sees columns <= n_past + j.
// y16 != nullptr: the probabilities are written as f16 rows of ld16 elements instead of in place — the V.P product
// converts them to f16 anyway (ggml converts src1 of an F16 mat-mul; k_gemm_f16 does it while loading), so the bits
// that reach the matrix cores are the same and the row is written and read once at half the size.
__global__ void __launch_bounds__(256) k_p_soft_max(float *x, int64_t rows, int nc, int nr, float scale, int n_past,
                                                    _Float16 *y16, int64_t ld16) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float *p = x + row * nc;
    _Float16 *p16 = y16 ? y16 + row * ld16 : nullptr;
    const int lim = n_past + (int)(row % nr);  // columns > lim are masked
    if (nc <= 512) {  // the whole row in registers: one read, one write (8 independent loads per lane; pairing adjacent
                      // columns per lane for 4-byte f16 stores was measured slower: 27 vs 23 us).  The kernel is VALU-bound
                      // (expf, f16 round trips and f64 adds per element), and under the causal mask half of the 64-column
                      // chunks of a batch are masked entirely: those are skipped with a wave-uniform branch (zeros stored).
        const int ulim = __builtin_amdgcn_readfirstlane(lim);  // one row per wave: uniform
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = lane + 64 * u;
            v[u] = 0.0f;
            if (64 * u <= ulim) v[u] = (i < nc && i <= lim) ? p[i] : 0.0f;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = lane + 64 * u;
            v[u] = (i < nc && i <= lim) ? v[u] * scale : -INFINITY;
            mx = fmaxf(mx, v[u]);
        }
        mx = wave_max_f32(mx);
        double sum = 0.0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            float e = 0.0f;
            if (64 * u <= ulim) {
                if (v[u] != -INFINITY) {
                    e = round_f16(expf(round_f16(v[u] - mx)));
                    sum += (double)e;
                }
            }
            v[u] = e;
        }
        sum = wave_sum_f64(sum);
        const float inv = (float)(1.0 / sum);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = lane + 64 * u;
            if (i < nc) {
                if (p16)
                    p16[i] = (_Float16)(v[u] * inv);
                else
                    p[i] = v[u] * inv;
            }
        }
        return;
    }
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
    for (int i = lane; i < nc; i += 64) {  // the same lane wrote p[i]
        if (p16)
            p16[i] = (_Float16)(p[i] * inv);
        else
            p[i] *= inv;
    }
}
