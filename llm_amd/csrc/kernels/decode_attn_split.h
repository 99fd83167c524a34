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

#define ATTN_SPLIT_MIN 512        // positions (n_past + 1) from which the split path replaces k_attn_decode (round 2; the K plan)
#define FUSE_HEADS_MIN 576        // positions from which k_qkv_attn takes 2 / 4 attention workgroups per head (window 512 + one 64-slab streamed)
#define ATTN_SPLIT_MIN_FUSED 768  // ... from which it replaces k_qkv_attn.  LLaMA-7B Q4_0 on one MI355X, ms per token
                            // (tests/tools/ctx_sweep.py, profiles/r04_ctx_sweep.txt): k_qkv_attn (one workgroup per head, register
                            // window of 512 positions, later ones streamed) 1.35 up to 480 positions, 1.50 at 560, 1.53 at 700,
                            // 1.64 at 900; wq|wk|wv + k_attn_split_one 1.60-1.65 from 300 to 1900 positions: they cross near 850.
                            // (Round 2, k_attn_decode against three split launches: 512, see the top of this file.)

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

// ---------------------------------------------------------------------------------------------------
// Round 4: the same three phases as ONE launch (k_attn_split_one).  The dependencies between them become hand-offs inside
// the launch — every one of the n_head x S workgroups is resident (one 1024-thread workgroup per CU, the launcher checks
// n_head * S <= #CUs), so a workgroup may wait for its S - 1 peers:
//   * what the workgroups of a head owe each other is two numbers each, not the score row: the range's maximum, and — once
//     the row maximum is known — the range's f64 sum of the f16-rounded exps.  Both travel as 8-byte {tag, 32 bit} granules
//     (the hand-off form of decode_fused.h: the data is the flag — one aligned agent-scope store, agent-scope polling loads
//     by S lanes of ONE wave, no fence, no counter); the f64 as two of them.  The row sum is exact in any order (every term
//     is a multiple of 2^-24 not above 1, at most 2^16 of them: 40 bits), so S partial sums added up equal the
//     three-launch version's thread-strided sum bit for bit; the scores themselves never leave the workgroup (LDS);
//   * the head's V range of the first 256 positions is requested at kernel ENTRY (it depends on nothing), so the V.P that
//     follows the second hand-off runs out of registers;
//   * the S partial outputs travel as granules too; an arrival counter per head elects the LAST workgroup, which sums the
//     partials (s ascending) and re-quantizes for wo.  Nobody waits for it.
// tag = the token's epoch (all 32 bits; k_rope_table bumps it once per token) and every layer has its OWN hand-off buffers,
// so a launch never takes another launch's granules for current — neither another layer's of this token nor, in a replayed
// hipGraph, an older token's (a 19-bit epoch shared with 12 bits of layer would repeat every 2^19 tokens).  Same float operations in the same order as the three launches:
// BIT-IDENTICAL outputs (tests/test_fused_attn_gpu.py runs both).  Every spin is bounded (err is raised).
// ---------------------------------------------------------------------------------------------------
struct AttnSplitOneArgs {
    AttnSplitArgs a;            // q, caches, prm, scale, shapes, Q8 outputs (sc / pmax / part unused)
    unsigned long long *mx_g;   // THIS layer's [n_head][S] range maxima
    unsigned long long *sum_g;  // ... [n_head][S][2] range sums (f64 as hi, lo words)
    unsigned long long *part_g; // ... [n_head][S][D] partial-output granules
    unsigned *cnt;              // [n_head] arrival counters (zero between launches)
    const unsigned *epoch;      // the token's epoch (k_rope_table)
    int layer;                  // (informational)
    unsigned *err;
    float *out_f32;             // nullable: the merged heads as f32 [n_head * D] (the K plan re-quantizes them to Q8_K itself)
};

__device__ __forceinline__ unsigned long long attn_one_wait(const unsigned long long *gp, unsigned tag, unsigned *err) {
    unsigned long long x = gran_load(gp);
    if ((unsigned)(x >> 32) == tag) return x;
    for (int spin = 0;; spin++) {
        __builtin_amdgcn_s_sleep(1);
        x = gran_load(gp);
        if ((unsigned)(x >> 32) == tag) return x;
        if (spin > GRAN_SPIN_MAX) {  // a peer never arrived (see GRAN_SPIN_MAX, kernels/common.h)
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return x;
        }
    }
}

template <bool F16_D>
__global__ void __launch_bounds__(1024) k_attn_split_one(const AttnSplitOneArgs f) {
    const AttnSplitArgs &a = f.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // chunk floats (scores, then exps) + chunk halves (probabilities)
    __shared__ float s_red[16];
    __shared__ double s_redd[16];
    __shared__ float s_mx;
    __shared__ double s_tot;
    __shared__ int s_last;
    const int h = blockIdx.x, s = blockIdx.y, hk = h / a.n_rep;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int T = a.prm->n_past + 1;
    const unsigned tag = *f.epoch;  // the token's epoch, all 32 bits: the hand-off buffers are per layer (AttnSplitOneArgs)
    int t0, t1;
    attn_split_range(T, a.S, s, t0, t1);
    const int chunk = (((T + a.S - 1) / a.S) + 63) & ~63;
    float *s_sc = (float *)smem;
    _Float16 *s_p = (_Float16 *)(s_sc + chunk);
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- V of the range's first 256 positions: requested now, used after the second hand-off
    const int cv = wave * 8 + (lane >> 3), pj = (lane & 7) * 8;
    const bool vact = cv < a.D;
    const __half *vbase = a.mem_v + ((int64_t)hk * a.D + cv) * a.C + t0 + pj;
    const int n8 = (t1 - t0 + 7) & ~7;
    f16x8 vpre[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        vpre[u] = zero8;
        if (vact && 64 * u + pj < n8) vpre[u] = *(const f16x8 *)(vbase + 64 * u);
    }

    // ---- phase 1 (= k_attn_split_scores): s_t = K[t].f16(q) * scale for the range -> LDS; the range maximum -> peers
    float mx = -INFINITY;
    {
        const int g = tid >> 4, gl = tid & 15, d0 = gl * 8;
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
                    if (gl == 0) s_sc[t - t0] = v;
                    mx = fmaxf(mx, v);
                }
            }
        }
    }
    mx = wave_max_f32(mx);
    if (lane == 0) s_red[wave] = mx;
    for (int i = tid; i < chunk; i += 1024) s_p[i] = (_Float16)0.0f;
    __syncthreads();
    if (wave == 0) {
        float m = lane < 16 ? s_red[lane] : -INFINITY;
        m = wave_max_f32(m);
        if (lane == 0) gran_store(f.mx_g + (int64_t)h * a.S + s, tag, __float_as_uint(m));  // an empty range publishes -inf
        // the row maximum: lanes 0..S-1 wait for one peer each (S <= 16)
        float pm = -INFINITY;
        if (lane < a.S) pm = __uint_as_float((unsigned)attn_one_wait(f.mx_g + (int64_t)h * a.S + lane, tag, f.err));
        pm = wave_max_f32(pm);
        if (lane == 0) s_mx = pm;
    }
    __syncthreads();
    mx = s_mx;
    // ---- phase 2 (= k_attn_split_vp): the range's f16-rounded exps and their f64 sum -> peers; the row sum
    double sum = 0.0;
    for (int i = tid; i < t1 - t0; i += 1024) {
        const float e = round_f16(expf(round_f16(s_sc[i] - mx)));
        s_sc[i] = e;
        sum += (double)e;
    }
    sum = wave_sum_f64(sum);
    if (lane == 0) s_redd[wave] = sum;
    __syncthreads();
    if (wave == 0) {
        if (lane == 0) {
            double loc = 0.0;
#pragma unroll
            for (int i = 0; i < 16; i++) loc += s_redd[i];
            const unsigned long long bits = (unsigned long long)__double_as_longlong(loc);
            gran_store(f.sum_g + ((int64_t)h * a.S + s) * 2, tag, (unsigned)(bits >> 32));
            gran_store(f.sum_g + ((int64_t)h * a.S + s) * 2 + 1, tag, (unsigned)bits);
        }
        double part = 0.0;
        if (lane < a.S) {
            const unsigned hi = (unsigned)attn_one_wait(f.sum_g + ((int64_t)h * a.S + lane) * 2, tag, f.err);
            const unsigned lo = (unsigned)attn_one_wait(f.sum_g + ((int64_t)h * a.S + lane) * 2 + 1, tag, f.err);
            part = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        }
        part = wave_sum_f64(part);  // exact: see the header
        if (lane == 0) s_tot = part;
    }
    __syncthreads();
    const float inv = (float)(1.0 / s_tot);
    for (int i = tid; i < t1 - t0; i += 1024) s_p[i] = (_Float16)(s_sc[i] * inv);
    __syncthreads();
    // V.P over the range: wave w owns channels 8w..8w+7, 8 lanes per channel, a lane covers 8 consecutive positions
    float acc = 0.0f;
#pragma unroll 1
    for (int p0 = 0; p0 < n8; p0 += 256) {
        f16x8 vv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            vv[u] = zero8;
            if (p0 == 0)
                vv[u] = vpre[u];
            else if (vact && p0 + 64 * u + pj < n8)
                vv[u] = *(const f16x8 *)(vbase + p0 + 64 * u);
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
    if ((lane & 7) == 0 && vact) gran_store(f.part_g + ((int64_t)h * a.S + s) * a.D + cv, tag, __float_as_uint(acc));
    // ---- phase 3 (= k_attn_split_out) by the last workgroup of the head to arrive
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(f.cnt + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == (unsigned)(a.S - 1);
        if (s_last) __hip_atomic_store(f.cnt + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // all S arrived: ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    const int nblk = a.D / 32, l = tid & 31, b = tid >> 5;
    if (b >= nblk) return;
    float v = 0.0f;
    for (int s2 = 0; s2 < a.S; s2++)
        v += __uint_as_float((unsigned)attn_one_wait(f.part_g + ((int64_t)h * a.S + s2) * a.D + b * 32 + l, tag, f.err));
    if (f.out_f32) f.out_f32[(int64_t)h * a.D + b * 32 + l] = v;
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
