// kquant_big.h — the K plan's decode mat-vec as ONE wave of big workgroups (Q4_K and Q6_K: what the *_K_M / *_K_S files of
// crates/llm-base/src/loader.rs:80-93 are made of), with the activation's norm / SiLU·mul and Q8_K quantization done by each
// workgroup while it stages x — the structure of k_mmvq_big (decode_big.h) for super-blocks of 256.
//
// Why: the K plan of round 4 ran the node-by-node executor's k_mmvq_k (four rows per 256-thread workgroup, 2048 workgroups) and
// needed a helper launch in front of every mat-vec group to produce the Q8_K row: 10 launches per layer, 163 helper launches per
// token = 0.46 ms of a 2.2 ms token.  Folding the quantization into k_mmvq_k's staging does not pay (2048 workgroups would each
// re-quantize the row); with 256 workgroups of 1024 threads it does: 6 launches per layer
//     wq|wk|wv (norm + Q8_K staged) | rope + K/V store | attention | wo (Q8_K staged) + residual | w1|w3 (norm + Q8_K staged) |
//     w2 (silu·mul + Q8_K staged) + residual
// Arithmetic: the row dots are k_mmvq_k<KT, 1>'s (same per-lane chunks, same f32 expression per chunk, same step order, same DPP
// wave reduction), the staging is k_k_norm_quant / k_k_quant / k_k_silu_mul_quant's (kquant_plan.h): the f64 sum of squares in the
// 256-thread order of k_rms_norm, Q8_K's order-free extreme / rounding / 16-sums.  The plan's results are BIT-IDENTICAL to the
// helper-launch form (tests/test_kquant_plan_gpu.py runs both).
#pragma once
#include "kquant.h"
#include "kquant2.h"
#include "decode.h"
#include "decode_fused.h"  // attn_consumer: the attention workgroups of k_qkv_attn_k

enum { KX_Q8K = 0, KX_NORM = 1, KX_F32 = 2, KX_SILU_MUL = 3 };

enum { KE_ROW = 0, KE_GATE = 1, KE_QKV = 2 };
struct KBigArgs {
    MmvqKArgs m;       // weights (up to three matrices of ONE type), dst, res; m.x only for KX_Q8K
    const float *xf;   // KX_NORM / KX_F32: the f32 row;  KX_SILU_MUL: w1 x
    const float *xw;   // KX_NORM: the norm weight;        KX_SILU_MUL: w3 x
    float eps;
    float *y_out;      // KX_NORM, nullable: f32 copy of the normed row (final norm -> OutputRequest.embeddings), written by workgroup 0
    // KE_QKV: the launch's matrices are some of wq / wk / wv (seg_kind[i] = 0 / 1 / 2 for m.w / m.wb / m.wc); a unit = two adjacent
    // rows of one matrix; the epilogue is k_k_rope_store's: RoPE of the pair (Q: f32 in place of dst; K: -> f16 -> the cache row of
    // the token's position), V: f16 into the transposed cache
    int seg_kind[3];
    const float *rope;      // the token's (cos, sin) table (k_rope_table)
    const DecParams *prm;
    __half *mem_k, *mem_v;  // + layer offset
    int64_t Egqa, C;
    int D;
    int wdeal;  // waves of a workgroup that take rows / units (0 = all 16): chosen per launch so that the units deal evenly (big_waves)
    // KE_QKV inside k_qkv_attn_k (below): every finished row pair is also PUBLISHED to the attention workgroups of the same launch as
    // one {epoch, f16 x 2} granule (index = the pair's index over wq|wk|wv), exactly as k_mmvq_big does inside k_qkv_attn
    unsigned long long *gran;
    const unsigned *epoch;
    const void *hot;  // 256 bytes the dummy ring steps read (BigArgs::hot, decode_big.h); nullptr = the first bytes of the scales
};

__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, dpp_i32<DPP_QUAD_XOR1>(v));
    v = min(v, dpp_i32<DPP_QUAD_XOR2>(v));
    v = min(v, dpp_i32<DPP_ROW_HALF_MIRROR>(v));
    v = min(v, dpp_i32<DPP_ROW_MIRROR>(v));
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16), r2 = __builtin_amdgcn_readlane(v, 32),
              r3 = __builtin_amdgcn_readlane(v, 48);
    return min(min(r0, r1), min(r2, r3));
}

// one super-block (256 values: lane l holds 4l .. 4l + 3) -> Q8_K in LDS; quantize_row_q8_K's arithmetic (k_quant_q8k): the
// first value of largest magnitude, iscale = -128 / max, q = min(127, nearest_int(iscale x)), 16-sums, d = 1 / iscale
__device__ __forceinline__ void q8k_wave_block(const f32x4 v, const int lane, const int sb, int8_t *s_q, float *s_d, int *s_b) {
    const float a0 = fabsf(v[0]), a1 = fabsf(v[1]), a2 = fabsf(v[2]), a3 = fabsf(v[3]);
    const float am = wave_max_f32(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)));
    int idx = a0 == am ? 4 * lane : a1 == am ? 4 * lane + 1 : a2 == am ? 4 * lane + 2 : a3 == am ? 4 * lane + 3 : 256;
    idx = wave_min_i32(idx);  // wave-uniform
    const int k = idx & 3;
    const float cand = k == 0 ? v[0] : k == 1 ? v[1] : k == 2 ? v[2] : v[3];
    const float mx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cand), (idx >> 2) & 63));
    int q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    float dd = 0.0f;
    if (am != 0.0f) {  // uniform
        const float iscale = -128.0f / mx;
        q0 = min(127, __float2int_rn(iscale * v[0]));
        q1 = min(127, __float2int_rn(iscale * v[1]));
        q2 = min(127, __float2int_rn(iscale * v[2]));
        q3 = min(127, __float2int_rn(iscale * v[3]));
        dd = 1.0f / iscale;
    }
    ((int *)s_q)[sb * 64 + lane] = (q0 & 0xFF) | ((q1 & 0xFF) << 8) | ((q2 & 0xFF) << 16) | (int)((unsigned)q3 << 24);
    int s16 = (q0 + q1) + (q2 + q3);
    s16 += dpp_i32<DPP_QUAD_XOR1>(s16);
    s16 += dpp_i32<DPP_QUAD_XOR2>(s16);
    if ((lane & 3) == 0) s_b[sb * 16 + (lane >> 2)] = s16;
    if (lane == 0) s_d[sb] = dd;
}

// one step of one row: this lane's 16-byte chunk of super-block sb against the staged Q8_K row — k_mmvq_k<KT, 1>'s expression
template <int KT>
__device__ __forceinline__ float kbig_chunk(const KStep<KT> &cur, const int c, const int sb, const int8_t *s_q, const float *s_d,
                                            const int *s_b) {
    if constexpr (KT == KT_Q4_K) {
        const int j = c >> 1, half = c & 1;
        const uint32_t scw = j < 2 ? cur.sc[0] : cur.sc[1];
        const int sc_lo = (int)((scw >> ((j & 1) * 16)) & 0xFF), sc_hi = (int)((scw >> ((j & 1) * 16 + 8)) & 0xFF);
        const uint32_t mw = c < 4 ? cur.sc[2] : cur.sc[3];
        const int mc = (int)((mw >> ((c & 3) * 8)) & 0xFF);
        const float d = __half2float(__ushort_as_half((unsigned short)(cur.dm & 0xFFFF)));
        const float dmin = __half2float(__ushort_as_half((unsigned short)(cur.dm >> 16)));
        const u32x4 lo = cur.q & 0x0F0F0F0Fu, hi = (cur.q >> 4) & 0x0F0F0F0Fu;
        const int8_t *xq = s_q + sb * 256 + 64 * j + 16 * half;
        const i32x4 xl = *(const i32x4 *)xq, xh = *(const i32x4 *)(xq + 32);
        const int isum = sc_lo * dot16(lo, xl, 0) + sc_hi * dot16(hi, xh, 0);
        const int *bp = s_b + sb * 16 + 2 * c;
        const int msum = mc * (bp[0] + bp[1]);
        const float d8 = s_d[sb];
        return (d * d8) * (float)isum - (dmin * d8) * (float)msum;
    } else {
        const int n2 = c >> 2, o = 16 * (c & 3);
        const uint32_t wa = n2 ? cur.sc[2] : cur.sc[0], wb = n2 ? cur.sc[3] : cur.sc[1];
        const int sc_a = (int)(int8_t)((wa >> (8 * (c & 3))) & 0xFF), sc_b = (int)(int8_t)((wb >> (8 * (c & 3))) & 0xFF);
        const float d = __half2float(__ushort_as_half((unsigned short)(cur.dm >> ((sb & 1) * 16))));  // (the word holds super-blocks 2i, 2i + 1)
        u32x4 lo, hi;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            lo[k] = (cur.q[k] & 0x0F0F0F0Fu) | (((cur.hA >> (2 * k)) & 0x03030303u) << 4);
            hi[k] = ((cur.q[k] >> 4) & 0x0F0F0F0Fu) | (((cur.hB >> (2 * k)) & 0x03030303u) << 4);
        }
        const int8_t *xq = s_q + sb * 256 + 128 * n2 + o;
        const i32x4 xl = *(const i32x4 *)xq, xh = *(const i32x4 *)(xq + 64);
        const int *bp = s_b + sb * 16 + 8 * n2 + (c & 3);
        const int isum = sc_a * (dot16(lo, xl, 0) - 32 * bp[0]) + sc_b * (dot16(hi, xh, 0) - 32 * bp[4]);
        return (d * s_d[sb]) * (float)isum;
    }
}

// ---- the 16-lanes-per-super-block family (Q2_K, Q3_K, Q5_K: kquant2.h): a lane's RAW loads of one 16-element group (what
// k2_load reads, kept in the ring) and their decoding + dot at consume time — k2_load's and k_mmvq_k2<KT, 1>'s expressions
template <int KT>
struct K2Raw {
    u32x4 q, a;    // quant bytes; Q3_K: hmask half, Q5_K: qh half
    uint32_t sm;   // Q2_K / Q3_K: the group's scale byte; Q5_K: scale | min << 8
    uint32_t dm;   // d | dmin << 16 (f16 pair)
};
template <int KT>
__device__ __forceinline__ void k2_issue(K2Raw<KT> &r, const uint8_t *qs, const uint8_t *aux, const uint8_t *sc, const __half *d,
                                         const int sb, const int g) {
    if constexpr (KT == KT_Q2_K || KT == KT_Q3_K) {
        r.q = *(const u32x4 *)(qs + sb * 64 + (2 * (g >> 3) + (g & 1)) * 16);
        if constexpr (KT == KT_Q3_K) r.a = *(const u32x4 *)(aux + sb * 32 + (g & 1) * 16);
        r.sm = sc[sb * 16 + g];
    } else {
        const int j64 = g >> 2, half = g & 1;
        r.q = *(const u32x4 *)(qs + sb * 128 + j64 * 32 + half * 16);
        r.a = *(const u32x4 *)(aux + sb * 32 + half * 16);
        r.sm = (uint32_t)sc[sb * 16 + (g >> 1)] | ((uint32_t)sc[sb * 16 + 8 + (g >> 1)] << 8);
    }
    r.dm = *(const uint32_t *)(d + sb * 2);
}
template <int KT>
__device__ __forceinline__ float k2_chunk(const K2Raw<KT> &r, const int g, const int sb, const int8_t *s_q, const float *s_d,
                                          const int *s_b) {
    u32x4 w;
    int sc, mn = 0;
    if constexpr (KT == KT_Q2_K) {
        const int shift = 2 * ((g >> 1) & 3);
        w = (r.q >> shift) & 0x03030303u;
        sc = (int)(r.sm & 15);
        mn = (int)(r.sm >> 4);
    } else if constexpr (KT == KT_Q3_K) {
        const int shift = 2 * ((g >> 1) & 3), bit = g >> 1;
        w = ((r.q >> shift) & 0x03030303u) | (((r.a >> bit) & 0x01010101u) << 2);
        sc = (int)r.sm - 32;
    } else {
        const int j64 = g >> 2, hi = (g >> 1) & 1;
        const u32x4 nib = hi ? ((r.q >> 4) & 0x0F0F0F0Fu) : (r.q & 0x0F0F0F0Fu);
        w = nib | (((r.a >> (2 * j64 + hi)) & 0x01010101u) << 4);
        sc = (int)(r.sm & 0xFF);
        mn = (int)(r.sm >> 8);
    }
    const float d = __half2float(__ushort_as_half((unsigned short)(r.dm & 0xFFFF)));
    const float dmin = __half2float(__ushort_as_half((unsigned short)(r.dm >> 16)));
    const i32x4 x = *(const i32x4 *)(s_q + sb * 256 + 16 * g);
    const int isum = dot16(w, x, 0), bsum = s_b[sb * 16 + g];
    const float d8 = s_d[sb];
    if constexpr (KT == KT_Q3_K)
        return (d * d8) * (float)(sc * (isum - 4 * bsum));
    else
        return (d * d8) * (float)(sc * isum) - (dmin * d8) * (float)(mn * bsum);
}
template <int KT>
struct KBigRing {
    using T = KStep<KT>;
};
template <>
struct KBigRing<KT_Q2_K> {
    using T = K2Raw<KT_Q2_K>;
};
template <>
struct KBigRing<KT_Q3_K> {
    using T = K2Raw<KT_Q3_K>;
};
template <>
struct KBigRing<KT_Q5_K> {
    using T = K2Raw<KT_Q5_K>;
};

#define KBIG_T 1024
// steps (8 super-blocks of one row) a wave keeps requested ahead.  With counted waits the depth is real (round 5's 4 was drained in
// front of every step): LLaMA-7B, -DKBIG_PF builds on one box (gpurun_out/r6/run24), tok/s Q4_K | Q6_K: 2: 678 | 585, 3: 676 | 579,
// 4: 666 | 517-562, 6: 624 | 518 — the K dots are VALU-bound, two steps in flight per wave x 16 waves keep the stream going
#ifndef KBIG_PF
#define KBIG_PF 2
#endif

// EPI = KE_GATE: the launch computes silu(w1 x) * (w3 x) (a.w = w1, a.wb = w3, same shape and type; dst = the product row): a wave's
// unit = row m of w1 followed by row m of w3, the epilogue lane of the unit applies ggml's f16-table SiLU and the multiply — the
// operations of k_k_silu_mul_quant's first half, once per element instead of once per workgroup and element in w2's staging.
// EPI = KE_QKV: unit = rows 2u, 2u + 1 of the launch's concatenated matrices (each with an even row count), epilogue = k_k_rope_store.
template <int KT, int XSRC, int EPI>
__device__ __forceinline__ void kbig_body(const KBigArgs &ka, const int bid, const int G) {
    const MmvqKArgs &a = ka.m;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double s_part[4];
    const int nsb = (int)a.w.nsb, K = nsb * 256;
    int8_t *s_q = (int8_t *)smem;                          // [K]
    float *s_d = (float *)(smem + (size_t)K);              // [nsb]
    int *s_b = (int *)(s_d + ((nsb + 3) & ~3));            // [nsb * 16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int W = KBIG_T / 64;
    constexpr bool K2 = KT == KT_Q2_K || KT == KT_Q3_K || KT == KT_Q5_K;  // 16 lanes per super-block, 4 super-blocks per step
    constexpr int LPS = K2 ? 16 : 8, SBS = 64 / LPS;
    const int c = lane & (LPS - 1), sbl = lane / LPS;
    const int nsteps = (nsb + SBS - 1) / SBS;
    using RingT = typename KBigRing<KT>::T;
    constexpr bool GATE = EPI == KE_GATE, PAIR = EPI != KE_ROW;
    const int Mt = GATE ? (int)a.w.M : EPI == KE_QKV ? (int)(mmvq_k_rows(a) >> 1) : (int)mmvq_k_rows(a);
    // rows (GATE: units) of this wave: dealt wave-major like k_mmvq_big: row = wave * G + bid + G * W * i
    const int Wd = ka.wdeal > 0 ? ka.wdeal : W;
    const int r_first = wave * G + bid, r_stride = G * Wd;
    const int nrw = (wave < Wd && r_first < Mt) ? (Mt - r_first + r_stride - 1) / r_stride : 0;
    const int nhr = PAIR ? 2 * nrw : nrw;  // weight rows the wave walks (GATE: w1's and w3's row of every unit; QKV: the pair's two rows)
    const int S = nhr * nsteps;            // steps of this wave

    // The load cursor walks (row of the wave, step of the row); the row's plane bases are wave-uniform and change only at a row
    // switch (scalar registers): a step costs a clamp and four small offsets instead of a division, a matrix select and 64-bit
    // index arithmetic — with 16 waves per CU sharing the VALU with the dots that was a third of the launch.
    int li = 0, ls = 0;
    const uint8_t *l_qs = a.w.qs, *l_sc = a.w.sc;
    const uint32_t *l_aux = a.w.aux;
    const __half *l_d = a.w.d;
    auto set_row = [&](int i) {
        KWeight w;
        int64_t row, ldd_;
        float *dst_;
        if constexpr (GATE) {
            w = (i & 1) ? a.wb : a.w;
            row = (int64_t)(r_first + r_stride * (i >> 1));
        } else if constexpr (EPI == KE_QKV) {
            mmvq_k_select(a, (int64_t)(2 * (r_first + r_stride * (i >> 1)) + (i & 1)), w, row, dst_, ldd_);
        } else {
            mmvq_k_select(a, (int64_t)(r_first + r_stride * i), w, row, dst_, ldd_);
        }
        const int64_t g0 = row * nsb;
        l_sc = w.sc + g0 * 16;
        if constexpr (K2) {
            l_qs = w.qs + g0 * (KT == KT_Q5_K ? 128 : 64);
            l_aux = (const uint32_t *)((const uint8_t *)w.aux + g0 * 32);
            l_d = w.d + g0 * 2;
        } else {
            l_qs = w.qs + g0 * 128;
            l_aux = w.aux + g0 * 16;
            l_d = w.d + g0 * (KT == KT_Q4_K ? 2 : 1);
        }
    };
    // A step past the wave's last real one is a DUMMY (see k_mmvq_big's `issue`, decode_big.h): every slot of the ring is refilled
    // unconditionally, so the number of loads in flight is a compile-time constant at every wait.  Round 5's form — refills under
    // `if (k + PF < S)`, a 16-bit scale load, a store that might be pending (the embedding tap) — compiled to an
    // s_waitcnt vmcnt(0) in front of EVERY step: a wave never had a second step in flight (tests/tools/disasm.py, round 6).
    const uint8_t *const hotp = ka.hot ? (const uint8_t *)ka.hot : (const uint8_t *)a.w.sc;
    auto load = [&](RingT &st, const bool dummy) {
        int sb = ls * SBS + sbl;
        sb = dummy ? 0 : sb < nsb ? sb : nsb - 1;  // lanes past the row end re-read the last super-block and are masked below
        const int cc = dummy ? 0 : c;
        const uint8_t *p_qs = dummy ? hotp : l_qs, *p_sc = dummy ? hotp : l_sc;
        const uint32_t *p_aux = dummy ? (const uint32_t *)hotp : l_aux;
        const __half *p_d = dummy ? (const __half *)hotp : l_d;
        if constexpr (K2) {
            k2_issue<KT>(st, p_qs, (const uint8_t *)p_aux, p_sc, p_d, sb, cc);
        } else {
            st.q = __builtin_nontemporal_load((const u32x4 *)(p_qs + sb * 128 + cc * 16));
            st.sc = *(const u32x4 *)(p_sc + sb * 16);
            if constexpr (KT == KT_Q4_K) {
                st.dm = *(const uint32_t *)(p_d + sb * 2);
            } else {
                const u32x2 h = __builtin_nontemporal_load((const u32x2 *)(p_aux + (sb * 8 + cc) * 2));
                st.hA = h[0];
                st.hB = h[1];
                // the f16 scale as the aligned 32-bit word that holds it (super-blocks 2i, 2i + 1), picked by parity at its use
                st.dm = *(const uint32_t *)(p_d + (sb & ~1));
            }
        }
        if (!dummy && ++ls == nsteps) {
            ls = 0;
            if (++li < nhr) set_row(li);
        }
    };
    // ---- 1. the activation's loads go FIRST (a wave's loads return in order: issued behind the weight ring they would only
    //         become usable once the whole ring has landed — the lesson of k_mmvq_big): this wave's
    //         super-blocks w, w + 16, ... (at most KBIG_SBW of them), and for the norm the 256-thread strided sum's elements
    constexpr int KBIG_SBW = XSRC == KX_NORM ? 2 : 4;  // super-blocks a wave stages: the normed rows are E wide (<= 8192: nsb <= 32),
                                                       // the others up to 16384 (nsb <= 64) (launcher)
    constexpr int KBIG_SQ = 16;   // elements per thread and pass of the 256-thread sum of squares (two passes for rows beyond 4096)
    f32x4 xv[KBIG_SBW], xw4[XSRC == KX_NORM || XSRC == KX_SILU_MUL ? KBIG_SBW : 1];
    float sq[XSRC == KX_NORM ? KBIG_SQ : 1];
    if constexpr (XSRC != KX_Q8K) {
#pragma unroll
        for (int u = 0; u < KBIG_SBW; u++) {
            const int sb = wave + W * u;
            const int i4 = (sb < nsb ? sb : 0) * 64 + lane;
            xv[u] = ((const f32x4 *)ka.xf)[i4];
            if constexpr (XSRC == KX_NORM || XSRC == KX_SILU_MUL) xw4[u] = ((const f32x4 *)ka.xw)[i4];
        }
        if constexpr (XSRC == KX_NORM) {
#pragma unroll
            for (int u = 0; u < KBIG_SQ; u++) {
                const int i = (tid & 255) + 256 * u;
                sq[u] = ka.xf[i < K ? i : 0];
            }
        }
    }
    // ---- 2. the weight stream starts before the activation is staged
    RingT ring[KBIG_PF];
    if (nhr > 0) set_row(0);
#pragma unroll
    for (int k = 0; k < KBIG_PF; k++) load(ring[k], k >= S);

    // ---- 3. stage x as Q8_K: wave w takes super-blocks w, w + 16, ...; lane l the values 4l .. 4l + 3 of the super-block
    if constexpr (XSRC == KX_Q8K) {
        for (int i = tid; i < K / 16; i += KBIG_T) ((i32x4 *)s_q)[i] = ((const i32x4 *)a.x.q8)[i];
        for (int i = tid; i < nsb; i += KBIG_T) s_d[i] = a.x.d8[i];
        for (int i = tid; i < nsb * 16; i += KBIG_T) s_b[i] = (int)a.x.bs[i];
    } else {
        float scale = 1.0f;
        if constexpr (XSRC == KX_NORM) {
            // k_rms_norm's / k_k_norm_quant's order: thread t of 256 adds elements t, t + 256, ... in f64, waves by DPP, (s0 + s1) + (s2 + s3)
            if (tid < 256) {
                double s = 0.0;
#pragma unroll
                for (int u = 0; u < KBIG_SQ; u++)
                    if (tid + 256 * u < K) s += (double)(sq[u] * sq[u]);
                if (K > 256 * KBIG_SQ) {  // rows beyond 4096: the rest of the thread's elements, still in ascending order
#pragma unroll
                    for (int u = 0; u < KBIG_SQ; u++) {
                        const int i = tid + 256 * (KBIG_SQ + u);
                        sq[u] = ka.xf[i < K ? i : 0];
                    }
#pragma unroll
                    for (int u = 0; u < KBIG_SQ; u++)
                        if (tid + 256 * (KBIG_SQ + u) < K) s += (double)(sq[u] * sq[u]);
                }
                s = wave_sum_f64(s);
                if (lane == 0) s_part[wave] = s;
            }
            __syncthreads();
            const double tot = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
            const float mean = (float)(tot / (double)K);
            scale = 1.0f / sqrtf(mean + ka.eps);
        }
#pragma unroll
        for (int u = 0; u < KBIG_SBW; u++) {
            const int sb = wave + W * u;
            if (sb >= nsb) break;  // uniform
            const int i4 = sb * 64 + lane;
            f32x4 v = xv[u];
            if constexpr (XSRC == KX_NORM) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float t = v[k] * scale;
                    v[k] = t * xw4[u][k];
                }
                // (only the lm_head launch has the tap: a store that MAY be pending turns every later wait into vmcnt(0))
                if constexpr (EPI == KE_ROW)
                    if (ka.y_out && bid == 0) ((f32x4 *)ka.y_out)[i4] = v;
            } else if constexpr (XSRC == KX_SILU_MUL) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float t = silu_table(v[k]);
                    v[k] = t * xw4[u][k];
                }
            }
            q8k_wave_block(v, lane, sb, s_q, s_d, s_b);
        }
    }
    __syncthreads();

    // ---- rows
    float acc = 0.0f, myv = 0.0f, myv3 = 0.0f;
    int ri = 0, rs = 0;  // row index of the wave, step inside the row
    auto step = [&](const RingT &st) {
        const int sb = rs * SBS + sbl;
        if (sb < nsb) {
            if constexpr (K2)
                acc += k2_chunk<KT>(st, c, sb, s_q, s_d, s_b);
            else
                acc += kbig_chunk<KT>(st, c, sb, s_q, s_d, s_b);
        }
        if (++rs == nsteps) {
            rs = 0;
            const float v = wave_sum_f32(acc);
            if constexpr (PAIR) {
                if (ri & 1)
                    myv3 = lane == (ri >> 1) ? v : myv3;
                else
                    myv = lane == (ri >> 1) ? v : myv;
            } else {
                myv = lane == ri ? v : myv;
            }
            acc = 0.0f;
            ri++;
        }
    };
    // full passes over the ring while a pass still has a step to request (every slot: dots, then its refill, unconditionally),
    // then a pass that drains the last PF steps and requests nothing: static wait counts throughout (k_mmvq_big, decode_big.h)
    int k0 = 0;
    for (; k0 + KBIG_PF < S; k0 += KBIG_PF) {
#pragma unroll
        for (int u = 0; u < KBIG_PF; u++) {
            step(ring[u]);
            load(ring[u], k0 + u + KBIG_PF >= S);
        }
    }
#pragma unroll
    for (int u = 0; u < KBIG_PF; u++)
        if (k0 + u < S) step(ring[u]);  // uniform
    if constexpr (GATE) {
        if (lane < nrw) a.dst[r_first + r_stride * lane] = silu_table(myv) * myv3;
        return;
    }
    if constexpr (EPI == KE_QKV) {
        if (lane < nrw) {
            const int64_t grow = (int64_t)2 * (r_first + r_stride * lane);
            int seg = 0;
            int64_t m0 = grow;
            if (a.nseg > 1 && grow >= a.w.M) {
                seg = 1;
                m0 = grow - a.w.M;
                if (a.nseg > 2 && m0 >= a.wb.M) {
                    seg = 2;
                    m0 -= a.wb.M;
                }
            }
            const int kind = seg == 0 ? ka.seg_kind[0] : seg == 1 ? ka.seg_kind[1] : ka.seg_kind[2];  // (a run-time index would put the argument block in scratch)
            const int p = ka.prm->n_past;
            __half h0, h1;  // the pair as f16: what the cache holds, and what ggml's F16 mat-mul makes of Q
            if (kind == 2) {  // V: f16 into the transposed cache
                h0 = __float2half_rn(myv);
                h1 = __float2half_rn(myv3);
                ka.mem_v[m0 * ka.C + p] = h0;
                ka.mem_v[(m0 + 1) * ka.C + p] = h1;
            } else {
                const int kk = (int)(m0 % ka.D) >> 1;
                const float cs = ka.rope[2 * kk], sn = ka.rope[2 * kk + 1];
                const float r0 = myv * cs - myv3 * sn, r1 = myv * sn + myv3 * cs;
                h0 = __float2half_rn(r0);
                h1 = __float2half_rn(r1);
                if (kind == 0) {
                    float *q = seg == 0 ? a.dst : seg == 1 ? a.dst_b : a.dst_c;
                    q[m0] = r0;
                    q[m0 + 1] = r1;
                } else {
                    ka.mem_k[(int64_t)p * ka.Egqa + m0] = h0;
                    ka.mem_k[(int64_t)p * ka.Egqa + m0 + 1] = h1;
                }
            }
            if (ka.gran) {  // one aligned 8-byte agent-scope store: the data is the flag (kernels/common.h gran_store)
                const unsigned v2 = (unsigned)__half_as_ushort(h0) | ((unsigned)__half_as_ushort(h1) << 16);
                gran_store(ka.gran + (r_first + r_stride * lane), *ka.epoch, v2);
            }
        }
        return;
    }
    if (lane < nrw) {
        KWeight w_;
        int64_t lrow, ldd_;
        float *dst_;
        mmvq_k_select(a, (int64_t)(r_first + r_stride * lane), w_, lrow, dst_, ldd_);
        dst_[lrow] = a.res ? myv + a.res[lrow] : myv;
    }
}
template <int KT, int XSRC, int EPI = KE_ROW>
__global__ void __launch_bounds__(KBIG_T) k_mmvq_kbig(const KBigArgs ka) {
    kbig_body<KT, XSRC, EPI>(ka, (int)blockIdx.x, (int)gridDim.x);
}

// wq|wk|wv of a K-quant model and the attention of the token in ONE launch: k_qkv_attn's structure (kernels/decode_fused.h) with
// the K mat-vec as the producer — workgroups 0 .. n_head - 1 are the attention (attn_consumer: K / V of the context preloaded while
// the weights stream, the token's own Q / K / V rows received as epoch-tagged granules, output as f32 for wo's staging), the
// others run kbig_body<KE_QKV> and publish every finished row pair.  Same arithmetic as k_attn_decode: bit-identical to the
// two-launch form.  One attention workgroup per head (contexts below the split threshold); all three matrices of one type.
template <int KT>
__global__ void __launch_bounds__(KBIG_T) k_qkv_attn_k(const KBigArgs ka, const FusedAttnArgs fa) {
    const int H = fa.n_head;
    if ((int)blockIdx.x < H) {
        attn_consumer<true>(fa, (int)blockIdx.x);
        return;
    }
    kbig_body<KT, KX_NORM, KE_QKV>(ka, (int)blockIdx.x - H, (int)gridDim.x - H);
}
