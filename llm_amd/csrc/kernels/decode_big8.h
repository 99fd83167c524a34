// decode_big8.h — the big-workgroup mat-vec of decode_big.h for 2..8 activation columns: the prompt chunks of
// InferenceSession::feed_prompt at the reference's default n_batch = 8 (crates/llm-base/src/inference_session.rs:
// 315-316, :837).  Same grid (one 1024-thread workgroup per CU, units dealt round-robin), same weight ring and
// waits, same epilogues — but every weight block is dotted with up to 8 activation blocks held in LDS, so a chunk of
// 8 tokens streams the weights once instead of 8 times.  Activations arrive already re-quantized (Q8 planar rows
// from k_rmsnorm_quant / k_quant_row / k_attn_decode: XSRC_Q8 only); with 8 columns the kernel is VALU-bound
// (8 x 0.8 lane-ops per weight), not HBM-bound.
#pragma once
#include "decode_big.h"

struct Big8Args {
    DecMmvqArgs d;     // d.x: Q8 rows [ncols][nb] (planar lo / hi / d / sum), d.dst / d.res: row 0
    int ncols;         // 2..8 (1 works too)
    int64_t ldd, ldr;  // floats between consecutive rows of dst / res (and of the Q output for EPI_QKV)
    const float *rope; // EPI_QKV: (cos, sin) tables of the chunk's positions, 128 floats per token (k_rope_table)
};

template <int QT, int EPI>
__global__ void __launch_bounds__(BIG_T) k_mmvq_big8(const Big8Args ba) {
    const DecMmvqArgs &a = ba.d;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float s_rope[EPI == EPI_QKV ? 8 * 128 : 2];  // [column][pair] cos, sin interleaved: D <= 128
    constexpr int RU = EPI == EPI_QKV ? 2 : 1, NW = EPI == EPI_GATE ? 2 : 1, NR = RU * NW;
    constexpr int PF = NR == 2 ? ((QT == QT_Q5_0 || QT == QT_Q5_1) ? 3 : 4) : 6, PF0 = 2;  // 16 accumulators + ring <= 128 VGPRs
    constexpr int NC = 8;
    const int ncols = ba.ncols;
    const int nb = (int)a.nb;
    const int nbl = (nb + 63) >> 6;
    const int nbp = nbl * 64;
    // LDS: per column c the planar Q8 row: lo[nbp] | hi[nbp] | d[nbp] | sum[nbp]
    i32x4 *s_lo = (i32x4 *)smem;
    i32x4 *s_hi = s_lo + (size_t)NC * nbp;
    float *s_d = (float *)(s_hi + (size_t)NC * nbp);
    int *s_sum = (int *)(s_d + (size_t)NC * nbp);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = (int)blockDim.x, W = T >> 6;  // waves per workgroup: chosen per launch (launch_big8), >= 8

    // ---- 1. activation loads first (see decode_big.h): up to 4 blocks per thread
    int n_past = 0;
    if constexpr (EPI == EPI_QKV) n_past = a.prm->n_past;
    constexpr int MAXB = 4;  // ncols * nb <= 4 * T blocks (checked by the launcher)
    i32x4 xl[MAXB], xh[MAXB];
    float xdv[MAXB];
    int xsv[MAXB];
    const int nx = ncols * nb;  // the Q8 rows are contiguous: block index = c * nb + b
#pragma unroll
    for (int u = 0; u < MAXB; u++) {
        const int i = u * T + tid;
        const int ic = i < nx ? i : 0;
        xl[u] = a.x.lo[ic];
        xh[u] = a.x.hi[ic];
        xdv[u] = a.x.d[ic];
        xsv[u] = a.x.sum[ic];
    }
    // EPI_QKV: the last 512 threads fetch the RoPE tables of the chunk's positions (8 columns x 64 pairs)
    f32x2 rope_pre = {0.0f, 0.0f};
    if constexpr (EPI == EPI_QKV) {
        const int e = tid - (T - 512), c = e >> 6, kk = e & 63;
        rope_pre = ((const f32x2 *)ba.rope)[(e >= 0 && c < ncols && kk < (a.D >> 1)) ? c * 64 + kk : 0];
    }

    const int M0 = (int)a.w[0].M, M1 = EPI == EPI_QKV ? (int)a.w[1].M : 0, M2 = EPI == EPI_QKV ? (int)a.w[2].M : 0;
    const int Utot = (M0 + M1 + M2) / RU;
    const int u_first = (int)blockIdx.x * W + wave, u_stride = (int)gridDim.x * W;
    const int nu = u_first < Utot ? (Utot - u_first + u_stride - 1) / u_stride : 0;  // <= 64 (launcher)
    const int S = nu * nbl;
    // EPI_ADD: lane i preloads the residuals of unit i, one per column
    float res_pre[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) res_pre[c] = 0.0f;
    if constexpr (EPI == EPI_ADD) {
        const int m = u_first + u_stride * lane;
        const bool ok = lane < nu && m < Utot;
#pragma unroll
        for (int c = 0; c < NC; c++) res_pre[c] = a.res[(ok && c < ncols) ? (int64_t)c * ba.ldr + m : 0];
    }

    auto resolve = [&](int i, int &sg, int &m0) {
        int r = (u_first + u_stride * i) * RU;
        if (r >= Utot * RU) r = 0;
        sg = 0;
        m0 = r;
        if constexpr (EPI == EPI_QKV) {
            if (r >= M0 + M1) {
                sg = 2;
                m0 = r - M0 - M1;
            } else if (r >= M0) {
                sg = 1;
                m0 = r - M0;
            }
        }
    };
    auto issue = [&](BigStep<QT, NR> &st, int i, int j, bool dummy) {
        int sg, m0;
        resolve(i, sg, m0);
        const int b = lane + 64 * j;
        const int bc = dummy ? 0 : (b < nb ? b : nb - 1);
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const uint8_t *qs = a.w[0].qs, *qs2 = a.w[0].qs2;
            const uint32_t *qh = a.w[0].qh;
            const __half *wd = a.w[0].d, *wm = a.w[0].m;
            if constexpr (EPI == EPI_GATE) {
                if (k == 1) {
                    qs = a.w[1].qs; qs2 = a.w[1].qs2; qh = a.w[1].qh; wd = a.w[1].d; wm = a.w[1].m;
                }
            } else if constexpr (EPI == EPI_QKV) {
                qs = sg == 0 ? a.w[0].qs : sg == 1 ? a.w[1].qs : a.w[2].qs;
                qs2 = sg == 0 ? a.w[0].qs2 : sg == 1 ? a.w[1].qs2 : a.w[2].qs2;
                qh = sg == 0 ? a.w[0].qh : sg == 1 ? a.w[1].qh : a.w[2].qh;
                wd = sg == 0 ? a.w[0].d : sg == 1 ? a.w[1].d : a.w[2].d;
                wm = sg == 0 ? a.w[0].m : sg == 1 ? a.w[1].m : a.w[2].m;
            }
            const uint32_t o = (uint32_t)(m0 + (EPI == EPI_QKV ? k : 0)) * (uint32_t)nb + (uint32_t)bc;
            st.q[k] = __builtin_nontemporal_load((const u32x4 *)(qs + (size_t)o * 16));
            if constexpr (QT == QT_Q8_0) st.p[k] = __builtin_nontemporal_load((const u32x4 *)(qs2 + (size_t)o * 16));
            if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) st.h[k] = __builtin_nontemporal_load(qh + o);
            st.dw[k] = ((const unsigned short *)wd)[o];  // raw f16 bits, zero-extended (BigStep)
            if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1) st.mw[k] = ((const unsigned short *)wm)[o];
        }
    };

    // ---- 2. weight prologue
    BigStep<QT, NR> ring[PF];
    int pi = 0, pj = 0;
    auto advance = [&](int k) {
        if (k + 1 < S && ++pj == nbl) {
            pj = 0;
            pi++;
        }
    };
#pragma unroll
    for (int k = 0; k < PF0; k++) {
        issue(ring[k], pi, pj, k >= S);
        advance(k);
    }

    // ---- 3. activations -> LDS (zero padding up to nbp per column); RoPE tables of the chunk's positions
    if constexpr (EPI == EPI_QKV) {
        if (tid >= T - 512) {
            const int e = tid - (T - 512), c = e >> 6, kk = e & 63;
            if (c < ncols && kk < (a.D >> 1)) {
                s_rope[(c * 64 + kk) * 2] = rope_pre[0];
                s_rope[(c * 64 + kk) * 2 + 1] = rope_pre[1];
            }
        }
    }
    for (int i = tid; i < NC * nbp; i += T) {
        const int c = i / nbp, b = i - c * nbp;
        if (b >= nb || c >= ncols) {
            s_lo[i] = i32x4{0, 0, 0, 0};
            s_hi[i] = i32x4{0, 0, 0, 0};
            s_d[i] = 0.0f;
            s_sum[i] = 0;
        }
    }
#pragma unroll
    for (int u = 0; u < MAXB; u++) {
        const int i = u * T + tid;
        if (i < nx) {
            const int c = i / nb, b = i - c * nb;
            s_lo[c * nbp + b] = xl[u];
            s_hi[c * nbp + b] = xh[u];
            s_d[c * nbp + b] = xdv[u];
            s_sum[c * nbp + b] = xsv[u];
        }
    }
#pragma unroll
    for (int k = PF0; k < PF; k++) {
        issue(ring[k], pi, pj, k >= S);
        advance(k);
    }
    __syncthreads();

    // ---- 4. dots: every weight block against the ncols activation blocks of its column
    float acc[NR][NC], myv[NR][NC];
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
        for (int c = 0; c < NC; c++) acc[r][c] = myv[r][c] = 0.0f;
    int ci = 0, cj = 0;
    for (int s = 0; s < S; s += PF) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            if (s + k < S) {
                const int b = lane + 64 * cj;
                const BigStep<QT, NR> &st = ring[k];
                // unpack each weight block once, then one v_dot4 chain + scale per activation column
                uint32_t wl[NR][4], wh[NR][4];
                float dwf[NR], mwf[NR];
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    u32x4 p2 = st.q[r];
                    uint32_t hh = 0;
                    mwf[r] = 0.0f;
                    if constexpr (QT == QT_Q8_0) p2 = st.p[r];
                    if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) hh = st.h[r];
                    if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1) mwf[r] = big_h2f(st.mw[r]);
                    dwf[r] = big_h2f(st.dw[r]);
                    block_unpack<QT>(st.q[r], p2, hh, wl[r], wh[r]);
                }
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    if (c < ncols) {  // uniform
                        const i32x4 lo = s_lo[c * nbp + b], hi = s_hi[c * nbp + b];
                        const float xd = s_d[c * nbp + b];
                        const int xs = s_sum[c * nbp + b];
#pragma unroll
                        for (int r = 0; r < NR; r++)
                            acc[r][c] += block_dot_codes<QT>(wl[r], wh[r], dwf[r], mwf[r], lo, hi, xd, xs);
                    }
                }
                if (++cj == nbl) {
                    cj = 0;
#pragma unroll
                    for (int r = 0; r < NR; r++)
#pragma unroll
                        for (int c = 0; c < NC; c++) {
                            const float v = wave_sum_f32(acc[r][c]);
                            myv[r][c] = lane == ci ? v : myv[r][c];
                            acc[r][c] = 0.0f;
                        }
                    ci++;
                }
                if (s + k + PF < S) {
                    issue(ring[k], pi, pj, false);
                    if (++pj == nbl) {
                        pj = 0;
                        pi++;
                    }
                }
            }
        }
    }

    // ---- 5. epilogues: lane i finishes unit i for every column
    if (lane < nu) {
        int sg, m0;
        resolve(lane, sg, m0);
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (c >= ncols) break;
            if constexpr (EPI == EPI_STORE) {
                a.dst[(int64_t)c * ba.ldd + m0] = myv[0][c];
            } else if constexpr (EPI == EPI_ADD) {
                a.dst[(int64_t)c * ba.ldd + m0] = myv[0][c] + res_pre[c];
            } else if constexpr (EPI == EPI_GATE) {
                a.dst[(int64_t)c * ba.ldd + m0] = silu_table(myv[0][c]) * myv[1][c];
            } else {
                const int p = n_past + c;  // position of this column's token
                if (sg == 2) {
                    a.mem_v[(int64_t)m0 * a.C + p] = __float2half_rn(myv[0][c]);
                    a.mem_v[(int64_t)(m0 + 1) * a.C + p] = __float2half_rn(myv[1][c]);
                } else {
                    const int kk = (m0 % a.D) >> 1;
                    const float cs = s_rope[(c * 64 + kk) * 2], sn = s_rope[(c * 64 + kk) * 2 + 1];
                    const float r0 = myv[0][c] * cs - myv[1][c] * sn, r1 = myv[0][c] * sn + myv[1][c] * cs;
                    if (sg == 0) {
                        a.dst[(int64_t)c * ba.ldd + m0] = r0;
                        a.dst[(int64_t)c * ba.ldd + m0 + 1] = r1;
                    } else {
                        a.mem_k[(int64_t)p * a.Egqa + m0] = __float2half_rn(r0);
                        a.mem_k[(int64_t)p * a.Egqa + m0 + 1] = __float2half_rn(r1);
                    }
                }
            }
        }
    }
}
