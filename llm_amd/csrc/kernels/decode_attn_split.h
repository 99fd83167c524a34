// Decode attention for long contexts, split over the positions as well as the heads.
//
// k_attn_decode gives one workgroup per head.  That is latency-optimal for a few hundred positions, but the head's
// K/V slice (1 MB at 1900 positions) then enters through ONE CU at ≈40 GB/s: 25 us per layer at 1900 positions, a
// third of the token.  Here S = #CUs / n_head workgroups share a head, each a contiguous range of positions.
// ggml's softmax is exact about its rounding points — row maximum first, exp of the f16-rounded difference rounded
// to f16, f64 sum, probabilities rounded to f16 before the V matmul — so an online-softmax rescaling is not an
// option; the split costs three dependent launches instead:
//   1. k_attn_split_scores : s_t = K[t]·f16(q) * scale for the workgroup's range -> sc[h][t]; range maximum -> pmax[h][s]
//   2. k_attn_split_vp     : every workgroup of a head redoes the (cheap) softmax bookkeeping of the WHOLE row from
//                            sc — maximum of the S partial maxima, e_t, f64 sum (exact: ≤ 2^16 terms of 11-bit
//                            mantissa, so the order of the sum cannot matter) — then p_t for its own range and the
//                            partial o_d = Σ_range V[d][t] p_t -> part[h][s][d]
//   3. k_attn_split_out    : o_d = Σ_s part[h][s][d] (s ascending), re-quantized to Q8 for the wo mat-vec
// Each launch moves a range's K or V through 256 CUs instead of 32.  The plan switches to this path from
// ATTN_SPLIT_MIN positions on (a second hipGraph of the same plan); below that the single launch wins.
// Measured (LLaMA-7B Q4_0, one MI355X): 547 -> 579 tok/s at 1000 positions, 471 -> 556 tok/s at 1900.
#pragma once
#include "decode.h"

#define ATTN_SPLIT_MIN 512  // positions (n_past + 1) from which the split path is used (7B Q4_0 on MI355X: the split
                            // token costs 1.69 ms from 200 to 1000 positions and 1.80 at 1900; the single launch 1.54 at
                            // 200, 1.62 at 330, 1.70-1.74 at 460-700, 1.83 at 1000, 2.12 at 1900)

struct AttnSplitArgs {
    const float *q;
    const __half *mem_k, *mem_v;
    const DecParams *prm;
    float scale;
    int D, n_rep, n_head, S;  // S = workgroups per head
    int64_t Egqa, C;
    float *sc;    // [n_head][C] scores
    float *pmax;  // [n_head][S]
    float *part;  // [n_head][S][D]
    int8_t *lo, *hi;
    float *dq;
    int *sumq;
};

// the range of positions of split s: [t0, t1), multiples of 64 except at the end
__device__ __forceinline__ void attn_split_range(int T, int S, int s, int &t0, int &t1) {
    const int chunk = (((T + S - 1) / S) + 63) & ~63;
    t0 = s * chunk < T ? s * chunk : T;
    t1 = t0 + chunk < T ? t0 + chunk : T;
}

__global__ void __launch_bounds__(1024) k_attn_split_scores(const AttnSplitArgs a) {
    __shared__ float s_red[16];
    const int h = blockIdx.x, s = blockIdx.y, hk = h / a.n_rep;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int T = a.prm->n_past + 1;
    int t0, t1;
    attn_split_range(T, a.S, s, t0, t1);
    const int g = tid >> 4, gl = tid & 15, d0 = gl * 8;  // 64 groups of 16 lanes, lane gl owns dims d0..d0+7
    const bool act = d0 < a.D;
    f32x4 q0 = {0.0f, 0.0f, 0.0f, 0.0f}, q1 = q0;
    if (act) {
        const float *qh = a.q + (int64_t)h * a.D;
        q0 = *(const f32x4 *)(qh + d0);
        q1 = *(const f32x4 *)(qh + d0 + 4);
    }
    f16x2 qh2[4];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        qh2[j] = f16x2{(_Float16)q0[2 * j], (_Float16)q0[2 * j + 1]};
        qh2[2 + j] = f16x2{(_Float16)q1[2 * j], (_Float16)q1[2 * j + 1]};
    }
    const __half *kbase = a.mem_k + (int64_t)hk * a.D + d0;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    float mx = -INFINITY;
#pragma unroll 1
    for (int tb = t0 + g; tb < t1; tb += 256) {
        f16x8 kv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = tb + 64 * u;
            kv[u] = zero8;
            if (act && t < t1) kv[u] = *(const f16x8 *)(kbase + (int64_t)t * a.Egqa);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = tb + 64 * u;
            float v = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; j++) v = __builtin_amdgcn_fdot2(f16x2{kv[u][2 * j], kv[u][2 * j + 1]}, qh2[j], v, false);
            v = g16_sum_f32(v);
            if (t < t1) {
                v *= a.scale;
                if (gl == 0) a.sc[(int64_t)h * a.C + t] = v;
                mx = fmaxf(mx, v);
            }
        }
    }
    mx = wave_max_f32(mx);
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    if (tid == 0) {
        float m = s_red[0];
#pragma unroll
        for (int i = 1; i < 16; i++) m = fmaxf(m, s_red[i]);
        a.pmax[h * a.S + s] = m;
    }
}

__global__ void __launch_bounds__(1024) k_attn_split_vp(const AttnSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // chunk halves: the probabilities of the range
    __shared__ double s_redd[16];
    const int h = blockIdx.x, s = blockIdx.y, hk = h / a.n_rep;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int T = a.prm->n_past + 1;
    int t0, t1;
    attn_split_range(T, a.S, s, t0, t1);
    const int chunk = (((T + a.S - 1) / a.S) + 63) & ~63;
    _Float16 *s_p = (_Float16 *)smem;  // [chunk] probabilities of this range, zero padded
    // the row maximum and the f64 sum of the WHOLE row (every workgroup of the head computes the same two numbers)
    float mx = a.pmax[h * a.S];
    for (int i = 1; i < a.S; i++) mx = fmaxf(mx, a.pmax[h * a.S + i]);
    const float *sc = a.sc + (int64_t)h * a.C;
    double sum = 0.0;
    for (int t = tid; t < T; t += 1024) sum += (double)round_f16(expf(round_f16(sc[t] - mx)));
    sum = wave_sum_f64(sum);
    if (lane == 0) s_redd[wave] = sum;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) tot += s_redd[i];
    const float inv = (float)(1.0 / tot);
    for (int i = tid; i < chunk; i += 1024) {
        const int t = t0 + i;
        s_p[i] = t < t1 ? (_Float16)(round_f16(expf(round_f16(sc[t] - mx))) * inv) : (_Float16)0.0f;
    }
    __syncthreads();
    // V·P over the range: wave w owns channels 8w..8w+7, 8 lanes per channel, a lane covers 8 consecutive positions
    const int cv = wave * 8 + (lane >> 3), pj = (lane & 7) * 8;
    const bool vact = cv < a.D;
    const __half *vbase = a.mem_v + ((int64_t)hk * a.D + cv) * a.C + t0 + pj;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int n8 = (t1 - t0 + 7) & ~7;  // 8-position pieces past the context are never touched (C % 8 == 0, t0 % 64 == 0)
    float acc = 0.0f;
#pragma unroll 1
    for (int p0 = 0; p0 < n8; p0 += 256) {
        f16x8 vv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            vv[u] = zero8;
            if (vact && p0 + 64 * u + pj < n8) vv[u] = *(const f16x8 *)(vbase + p0 + 64 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int pos = p0 + 64 * u + pj;
            if (pos < n8) {
                const f16x8 pp = *(const f16x8 *)(s_p + pos);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc = __builtin_amdgcn_fdot2(f16x2{vv[u][2 * j], vv[u][2 * j + 1]}, f16x2{pp[2 * j], pp[2 * j + 1]}, acc, false);
            }
        }
    }
    acc = g8_sum_f32(acc);
    if ((lane & 7) == 0 && vact) a.part[((int64_t)h * a.S + s) * a.D + cv] = acc;
}

template <bool F16_D>
__global__ void __launch_bounds__(256) k_attn_split_out(const AttnSplitArgs a) {
    const int h = blockIdx.x, tid = threadIdx.x;
    const int nblk = a.D / 32, l = tid & 31, b = tid >> 5;
    if (b >= nblk) return;
    float v = 0.0f;
    for (int s = 0; s < a.S; s++) v += a.part[((int64_t)h * a.S + s) * a.D + b * 32 + l];
    float amax = fabsf(v);
    amax = g32_max_f32(amax);
    const float d = amax / 127.0f;
    const float id = act_id(amax, d, aq_scalar());
    const int qv = act_q(v * id, aq_scalar());
    int sq = qv;
    sq = g32_sum_i32(sq);
    const int64_t gb = (int64_t)h * nblk + b;
    (l < 16 ? a.lo : a.hi)[gb * 16 + (l & 15)] = (int8_t)qv;
    if (l == 0) {
        a.dq[gb] = F16_D ? round_f16(d) : d;
        a.sumq[gb] = sq;
    }
}
