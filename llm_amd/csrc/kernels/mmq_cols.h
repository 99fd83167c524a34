// mmq_cols.h — quantized mat-mul for 2..8 activation columns on the INTEGER matrix cores: the prompt chunks of
// InferenceSession::feed_prompt at the reference's default n_batch = 8 (crates/llm-base/src/inference_session.rs:315-316,
// :837).  k_mmvq_big8 (decode_big8.h) dots every weight block with the 8 activation blocks on the VALU (8 x v_dot4 + a
// scale per column: ~100 lane-operations per block) and is VALU-bound: 29 us for a 7B w1|w3 that streams in 11.7 us with
// one column.  Here the integer block dots are one v_mfma_i32_16x16x64_i8 per 16 weight rows x 4 blocks x 4 columns:
//
//   B operand = weights: lane (row j = lane & 15, block group g = lane >> 4) holds the 16 codes of the low (then high)
//               half of block b+g of row j — from the very 16 bytes that lane loaded (SoA plane of common.h, no shuffles);
//   A operand = activations: MFMA row i = 4 g' + c carries column c's block b+g' in k group g' and ZEROS in the other
//               three k groups, so one instruction yields, for every (row j, block b+g), the four integer dots with
//               columns c = 0..3 — in the registers of the lane that holds that block's scale d_w;
//   the lane then applies ggml's per-block formula (block_dot_codes of mmvq.h: zero point, d_w * d_x, m_w * s_x) in f32
//   and accumulates per column.  Columns 4..7 take a second pair of instructions on the same weight registers.
//
// ggml's contract is kept: activations re-quantized to Q8_0 / Q8_1, exact int32 block dots, f32 accumulation across
// blocks (only the order of that f32 sum differs: 2e-5 * scale, the mat-vec bound of tests/test_ops_gpu.py).
//
// Work split: the workgroup (8 waves) owns a contiguous range of 16-row groups; a unit = (group, chunk of `kc` blocks of
// K) goes to wave (unit % 8); partial sums of a row's chunks meet in LDS in chunk order (deterministic), then the
// epilogues of decode_big8.h run (store / +residual / silu(w1 x) * w3 x / RoPE + K,V store per token).
#pragma once
#include "decode_big8.h"

struct ColsArgs {
    DecMmvqArgs d;     // d.x: Q8 rows [ncols][nb] (planar), d.dst / d.res: row 0
    int ncols;         // 2..8
    int64_t ldd, ldr;  // floats between consecutive rows of dst / res
    const float *rope; // EPI_QKV: (cos, sin) tables of the chunk's positions, 128 floats per token (k_rope_table)
    int kc;            // blocks per K chunk (multiple of 4)
    int ngroups;       // 16-row groups (EPI_GATE: pairs of a w1 and a w3 group) in the launch
};

typedef int i32x4v __attribute__((ext_vector_type(4)));

#define COLS_T 512
#define COLS_W 8
#define COLS_MAX_UNITS 96  /* (sub-group, chunk) units of a workgroup: 48 KB of partial sums */

template <int QT>
struct ColsStep {
    u32x4 q, p;   // quants (p: Q8_0's second plane)
    uint32_t h;   // Q5: fifth bits
    __half dw, mw;
};

template <int QT>
__device__ __forceinline__ float cols_scale(int sumi, float dw, float mw, float xd, int xs) {  // = block_dot_codes after its dots
    if constexpr (QT == QT_Q4_0) {
        return ((float)(sumi - 8 * xs) * dw) * xd;
    } else if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1) {
        return (dw * xd) * (float)sumi + mw * ((float)xs * xd);
    } else if constexpr (QT == QT_Q5_0) {
        return (dw * xd) * (float)(sumi - 16 * xs);
    } else {
        return (float)sumi * (dw * xd);
    }
}

template <int QT, int EPI>
__global__ void __launch_bounds__(COLS_T) k_mmq_cols(const ColsArgs ca) {
    const DecMmvqArgs &a = ca.d;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NSUB = EPI == EPI_GATE ? 2 : 1;  // weight matrices a group draws rows from (w1 and w3)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncols = ca.ncols, nb = (int)a.nb, kc = ca.kc;
    const int nk = (nb + kc - 1) / kc;
    // LDS: activation planes [8][nb] x 16 B (low / high halves), per-block scales and quant sums [nb][8],
    // the partial sums [unit][16 rows][8 columns]
    i32x4 *s_lo = (i32x4 *)smem;
    i32x4 *s_hi = s_lo + 8 * nb;
    float *s_dx = (float *)(s_hi + 8 * nb);
    int *s_sx = (int *)(s_dx + 8 * nb);
    float *s_part = (float *)(s_sx + 8 * nb);

    // ---- this workgroup's groups
    const int G = (int)gridDim.x, bid = (int)blockIdx.x;
    const int g_begin = (int)(((int64_t)ca.ngroups * bid) / G), g_end = (int)(((int64_t)ca.ngroups * (bid + 1)) / G);
    const int nlg = g_end - g_begin;             // groups here
    const int nunits = nlg * NSUB * nk;          // <= COLS_MAX_UNITS (launcher)
    // group -> (matrix, first row)
    const int M0 = (int)a.w[0].M, M1 = EPI == EPI_QKV ? (int)a.w[1].M : 0;
    auto group_rows = [&](int g, int sub, int &sg, int &m0) {
        int r = g * 16;
        sg = 0;
        if constexpr (EPI == EPI_QKV) {
            if (r >= M0 + M1) {
                sg = 2;
                r -= M0 + M1;
            } else if (r >= M0) {
                sg = 1;
                r -= M0;
            }
        } else if constexpr (EPI == EPI_GATE) {
            sg = sub;
        }
        m0 = r;
    };

    // ---- activations -> LDS
    {
        const int nx = ncols * nb;
        for (int i = tid; i < 8 * nb; i += COLS_T) {
            const int c = i / nb, b = i - c * nb;
            i32x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
            float d = 0.0f;
            int s = 0;
            if (i < nx) {
                lo = a.x.lo[i];
                hi = a.x.hi[i];
                d = a.x.d[i];
                s = a.x.sum[i];
            }
            s_lo[i] = lo;
            s_hi[i] = hi;
            s_dx[b * 8 + c] = d;
            s_sx[b * 8 + c] = s;
        }
    }

    // ---- the wave's units, a weight ring across them
    const int jrow = lane & 15, bg = lane >> 4;
    const bool act = ((lane & 15) >> 2) == bg;  // A operand: this lane's row carries a column of its own k group
    const int acol = lane & 3;
    struct Cursor {
        int u;     // unit index (local): u = (lg * NSUB + sub) * nk + chunk
        int st;    // super-tile (4 blocks) within the chunk
        int nst;   // super-tiles of this chunk
        const uint8_t *qs, *qs2;
        const uint32_t *qh;
        const __half *wd, *wm;
        uint32_t row_blk;  // (row of this lane) * nb
        int b0;            // first block of the chunk
    };
    auto open_unit = [&](Cursor &c, int u) {
        c.u = u;
        c.st = 0;
        if (u >= nunits) {
            c.nst = 0;
            return;
        }
        const int chunk = u % nk, gs = u / nk, sub = gs % NSUB, lg = gs / NSUB;
        int sg, m0;
        group_rows(g_begin + lg, sub, sg, m0);
        // scalar selects (sg is wave-uniform): a kernarg array indexed by a "divergent" id is fetched with vector loads
        c.qs = sg == 0 ? a.w[0].qs : sg == 1 ? a.w[1].qs : a.w[2].qs;
        c.qs2 = sg == 0 ? a.w[0].qs2 : sg == 1 ? a.w[1].qs2 : a.w[2].qs2;
        c.qh = sg == 0 ? a.w[0].qh : sg == 1 ? a.w[1].qh : a.w[2].qh;
        c.wd = sg == 0 ? a.w[0].d : sg == 1 ? a.w[1].d : a.w[2].d;
        c.wm = sg == 0 ? a.w[0].m : sg == 1 ? a.w[1].m : a.w[2].m;
        c.row_blk = (uint32_t)(m0 + jrow) * (uint32_t)nb;
        c.b0 = chunk * kc;
        c.nst = (min(nb, c.b0 + kc) - c.b0) >> 2;
    };
    auto issue = [&](ColsStep<QT> &s, const Cursor &c) {
        const uint32_t o = c.row_blk + (uint32_t)(c.b0 + 4 * c.st + bg);
        s.q = __builtin_nontemporal_load((const u32x4 *)(c.qs + (size_t)o * 16));
        if constexpr (QT == QT_Q8_0) s.p = __builtin_nontemporal_load((const u32x4 *)(c.qs2 + (size_t)o * 16));
        if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) s.h = __builtin_nontemporal_load(c.qh + o);
        s.dw = c.wd[o];
        if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1) s.mw = c.wm[o];
    };
    auto advance = [&](Cursor &c) {
        if (++c.st >= c.nst) open_unit(c, c.u + COLS_W);
    };
    constexpr int PF = QT == QT_Q8_0 ? 4 : 6;
    ColsStep<QT> ring[PF];
    Cursor pc;  // producer
    open_unit(pc, wave);
#pragma unroll
    for (int k = 0; k < PF; k++) {
        if (pc.nst > 0) {
            issue(ring[k], pc);
            advance(pc);
        }
    }
    __syncthreads();  // activations staged

    Cursor cc;  // consumer
    open_unit(cc, wave);
    float f[8];
#pragma unroll
    for (int c = 0; c < 8; c++) f[c] = 0.0f;
    while (cc.nst > 0) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            if (cc.nst > 0) {  // wave-uniform
                const ColsStep<QT> st = ring[k];
                if (pc.nst > 0) {
                    issue(ring[k], pc);
                    advance(pc);
                }
                const int bl = cc.b0 + 4 * cc.st + bg;  // this lane's block (weights, scales, outputs)
                uint32_t wl[4], wh[4];
                block_unpack<QT>(st.q, st.p, st.h, wl, wh);
                const i32x4v bwl = {(int)wl[0], (int)wl[1], (int)wl[2], (int)wl[3]}, bwh = {(int)wh[0], (int)wh[1], (int)wh[2], (int)wh[3]};
                const float dw = __half2float(st.dw);
                float mw = 0.0f;
                if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1) mw = __half2float(st.mw);
                const i32x4v zero4 = {0, 0, 0, 0};
                {
                    const i32x4v alo = act ? *(const i32x4v *)((const char *)s_lo + (acol * nb + bl) * 16) : zero4;
                    const i32x4v ahi = act ? *(const i32x4v *)((const char *)s_hi + (acol * nb + bl) * 16) : zero4;
                    i32x4v acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(alo, bwl, zero4, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(ahi, bwh, acc, 0, 0, 0);
                    const f32x4 xd = *(const f32x4 *)(s_dx + bl * 8);
                    const i32x4v xs = *(const i32x4v *)(s_sx + bl * 8);
#pragma unroll
                    for (int c = 0; c < 4; c++) f[c] += cols_scale<QT>(acc[c], dw, mw, xd[c], xs[c]);
                }
                if (ncols > 4) {  // uniform
                    const i32x4v alo = act ? *(const i32x4v *)((const char *)s_lo + ((acol + 4) * nb + bl) * 16) : zero4;
                    const i32x4v ahi = act ? *(const i32x4v *)((const char *)s_hi + ((acol + 4) * nb + bl) * 16) : zero4;
                    i32x4v acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(alo, bwl, zero4, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(ahi, bwh, acc, 0, 0, 0);
                    const f32x4 xd = *(const f32x4 *)(s_dx + bl * 8 + 4);
                    const i32x4v xs = *(const i32x4v *)(s_sx + bl * 8 + 4);
#pragma unroll
                    for (int c = 0; c < 4; c++) f[4 + c] += cols_scale<QT>(acc[c], dw, mw, xd[c], xs[c]);
                }
                if (cc.st + 1 >= cc.nst) {  // the unit is complete: add the four block groups, park the 16 x 8 sums
#pragma unroll
                    for (int c = 0; c < 8; c++) {
                        float v = f[c];
                        v += __shfl_xor(v, 16, 64);
                        v += __shfl_xor(v, 32, 64);
                        f[c] = v;
                    }
                    if (lane < 16) {
                        float *o = s_part + ((size_t)cc.u * 16 + lane) * 8;
                        *(f32x4 *)o = f32x4{f[0], f[1], f[2], f[3]};
                        *(f32x4 *)(o + 4) = f32x4{f[4], f[5], f[6], f[7]};
                    }
#pragma unroll
                    for (int c = 0; c < 8; c++) f[c] = 0.0f;
                }
                advance(cc);
            }
        }
    }
    __syncthreads();

    // ---- epilogues: thread per (group, row or row pair, column); chunk partials added in chunk order
    auto total = [&](int lg, int sub, int row, int c) {
        const float *p = s_part + ((size_t)((lg * NSUB + sub) * nk) * 16 + row) * 8 + c;
        float v = 0.0f;
        for (int k = 0; k < nk; k++) v += p[(size_t)k * 128];
        return v;
    };
    if constexpr (EPI == EPI_QKV) {
        const int n_past = a.prm->n_past;
        for (int i = tid; i < nlg * 8 * ncols; i += COLS_T) {
            const int c = i % ncols, pr = (i / ncols) & 7, lg = i / (ncols * 8);
            int sg, m0;
            group_rows(g_begin + lg, 0, sg, m0);
            const int m = m0 + 2 * pr;
            const float v0 = total(lg, 0, 2 * pr, c), v1 = total(lg, 0, 2 * pr + 1, c);
            const int p = n_past + c;
            if (sg == 2) {
                a.mem_v[(int64_t)m * a.C + p] = __float2half_rn(v0);
                a.mem_v[(int64_t)(m + 1) * a.C + p] = __float2half_rn(v1);
            } else {
                const int kk = (m % a.D) >> 1;
                const float cs = ca.rope[(c * 64 + kk) * 2], sn = ca.rope[(c * 64 + kk) * 2 + 1];
                const float r0 = v0 * cs - v1 * sn, r1 = v0 * sn + v1 * cs;
                if (sg == 0) {
                    a.dst[(int64_t)c * ca.ldd + m] = r0;
                    a.dst[(int64_t)c * ca.ldd + m + 1] = r1;
                } else {
                    a.mem_k[(int64_t)p * a.Egqa + m] = __float2half_rn(r0);
                    a.mem_k[(int64_t)p * a.Egqa + m + 1] = __float2half_rn(r1);
                }
            }
        }
    } else {
        for (int i = tid; i < nlg * 16 * ncols; i += COLS_T) {
            const int row = i & 15, c = (i >> 4) % ncols, lg = i / (16 * ncols);  // consecutive threads: consecutive rows
            int sg, m0;
            group_rows(g_begin + lg, 0, sg, m0);
            const int m = m0 + row;
            const float v = total(lg, 0, row, c);
            if constexpr (EPI == EPI_STORE) {
                a.dst[(int64_t)c * ca.ldd + m] = v;
            } else if constexpr (EPI == EPI_ADD) {
                a.dst[(int64_t)c * ca.ldd + m] = v + a.res[(int64_t)c * ca.ldr + m];
            } else {
                a.dst[(int64_t)c * ca.ldd + m] = silu_table(v) * total(lg, 1, row, c);
            }
        }
    }
}
