// mmq_plain.h — k_mmq: the prompt GEMM for an ODD number of 32-wide blocks per row (K/32 odd: e.g. the 352-wide feed-forward
// of the test models).  The persistent kernels (mmq_dmap8.h, mmq_w16.h, mmq_w16_256.h) advance K two blocks per stage and
// fetch a row's scales as aligned pairs, so they need K/32 even — every LLaMA size has that; this register-staged kernel
// (one workgroup per 128 x 128 tile, in-LDS dequant, same tile arithmetic and k order) takes the rest.  No other launch
// uses it.  One workgroup per CU-half was its round-1 tuning (__launch_bounds__(256, 2) — and scratch spills); as a
// fallback it simply takes the registers it needs.
#pragma once
#include "mmq.h"

// registers holding one stage of global data in flight: the weight block (5..10 VGPRs) and the 4 activation chunks
// (16 VGPRs); two ring slots each (stages s+1 and s+2)
template <int QT>
struct MmqW {
    u32x4 q, q2;
    uint32_t qh;
    _Float16 d, m;
};
struct MmqX {
    u32x4 xa[4];
};

// Branch-free so that the whole k-stage is ONE basic block: addresses are clamped to the last valid block, and a
// block past the end of K gets d = m = 0 (its dequantized weights are exactly 0, which also cancels the — finite —
// activations loaded for it).
struct MmqSel {  // the launch's matrix (of up to three) this workgroup's tile belongs to: scalar selects, no copy of MmqArgs
    QWeight w;
    const _Float16 *x;
    float *dst;
    int64_t ldd, M, N, nb, split_stride;
};
template <int QT>
__device__ __forceinline__ void mmq_load_w(MmqW<QT> &s, const MmqSel &a, int64_t wrow, int64_t kb /*first block*/, int wj) {
    const bool kvalid = kb + wj < a.nb;
    const int64_t blk = wrow * a.nb + (kvalid ? kb + wj : a.nb - 1);
    s.q = __builtin_nontemporal_load((const u32x4 *)(a.w.qs) + blk);
    if constexpr (QT == QT_Q8_0) s.q2 = __builtin_nontemporal_load((const u32x4 *)(a.w.qs2) + blk);
    if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) s.qh = a.w.qh[blk];
    const _Float16 d = ((const _Float16 *)a.w.d)[blk];
    s.d = kvalid ? d : (_Float16)0.0f;
    if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1) {
        const _Float16 m = ((const _Float16 *)a.w.m)[blk];
        s.m = kvalid ? m : (_Float16)0.0f;
    }
}
__device__ __forceinline__ void mmq_load_x(MmqX &s, const MmqSel &a, int64_t kb, const _Float16 *xrow[4], int xc) {
    int64_t kel = kb * 32 + xc * 8;  // chunk column xc (0..7) of rows xrow[i]
    kel = kel < a.nb * 32 ? kel : a.nb * 32 - 8;
#pragma unroll
    for (int i = 0; i < 4; i++) s.xa[i] = *(const u32x4 *)(xrow[i] + kel);
}

// Software pipeline, one barrier per 64-wide k stage:
//     iteration s:   barrier | global loads of stage s+2 -> registers
//                            | MFMA on LDS[s & 1]   (matrix pipe)
//                            | dequantize stage s+1 -> LDS[(s+1) & 1]   (VALU + LDS-write pipe, independent of the MFMAs)
// so the dequant + ds_write of the next stage (~300 VALU + ~400 LDS-write cycles) hide under the 512 MFMA cycles of the
// current one instead of preceding them (first version: write -> barrier -> MFMA, 12 % of the f16 peak).
// gridDim.y = number of K splits (1 or 2): with 2 the partial tiles are combined with f32 atomic adds into a zeroed
// dst — two addends commute, so the result does not depend on arrival order.
template <int QT>
__global__ void __launch_bounds__(256) k_mmq(const MmqArgs a_in) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;

    // workgroup -> tile: the workgroups of one XCD walk consecutive tiles (token tiles of one weight slab back to back)
    const int t = xcd_tile_id(blockIdx.x, gridDim.x);
    int tm = t / a_in.tiles_n;
    const int tn = t % a_in.tiles_n;
    const int seg = a_in.nseg > 2 && tm >= a_in.tile_end[1] ? 2 : a_in.nseg > 1 && tm >= a_in.tile_end[0] ? 1 : 0;
    if (seg) tm -= a_in.tile_end[seg - 1];
    MmqSel a;
    a.w = seg == 2 ? a_in.wc : seg == 1 ? a_in.wb : a_in.w;
    a.dst = seg == 2 ? a_in.dst_c : seg == 1 ? a_in.dst_b : a_in.dst;
    a.ldd = seg == 2 ? a_in.ldd_c : seg == 1 ? a_in.ldd_b : a_in.ldd;
    a.x = a_in.x;
    a.M = a.w.M;
    a.N = a_in.N;
    a.nb = a_in.nb;
    a.split_stride = a_in.split_stride;
    const int64_t m0 = (int64_t)tm * MMQ_TM, n0 = (int64_t)tn * MMQ_TN;

    // staging assignment
    const int wr = tid >> 1, wj = tid & 1;  // weight row / block-in-stage
    const int64_t wrow = min(m0 + wr, a.M - 1);
    const int xc = tid & 7;  // 16-byte chunk column of the activation tile
    const _Float16 *xrow[4];
#pragma unroll
    for (int i = 0; i < 4; i++) xrow[i] = a.x + min(n0 + (tid >> 3) + 32 * i, a.N - 1) * (a.nb * 32);

    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[j][i][r] = 0.0f;

    // this workgroup's stages [s_begin, s_end) of the K loop
    const int nstage_all = (int)((a.nb + 1) >> 1);
    const int per = (nstage_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int s_begin = (int)blockIdx.y * per, s_end = min(nstage_all, s_begin + per);
    const int nstage = s_end - s_begin;

    MmqW<QT> wr_[2];  // weight ring: stage s+1 (being dequantized), s+2 (in flight)
    MmqX xr_[2];      // activation ring: stage s+1, s+2
#pragma unroll
    for (int u = 0; u < 2; u++) {
        wr_[u].q2 = u32x4{0, 0, 0, 0};
        wr_[u].qh = 0;
        wr_[u].m = (_Float16)0.0f;
    }
    auto kb_of = [&](int s) { return (int64_t)(s_begin + min(s, nstage - 1)) * 2; };  // clamped: see mmq_load_w
    const int frag_off = (lane & 31) * MMQ_ROWB + (lane >> 5) * 16;
    const int xoff = (tid >> 3) * MMQ_ROWB + xc * 16, woff = wr * MMQ_ROWB + wj * 64;

    // One k-stage, hand-interleaved: after EACH of the 16 MFMAs (32 cycles on the matrix pipe, ~8 issue slots)
    // comes one slice of the other work — a fragment read for the next k-step, one 32-bit slice of the dequant of
    // stage s+1 (5 VALU ops), a ds_write when a 16-byte word is complete — and a scheduling barrier that keeps
    // hipcc from clustering the MFMAs (it does, and then the VALU/LDS work runs with the matrix pipe idle).
    // Ring slots are compile-time: the caller unrolls by 2.  st/sx hold stage s+1; lw/lx receive stage s+3.
    auto stage = [&](int s, const MmqW<QT> &st, const MmqX &sx, MmqW<QT> &lw, MmqX &lx) {
        const char *W = lds + (s & 1) * 2 * MMQ_TILEB, *X = W + MMQ_TILEB;
        char *Wn = lds + ((s + 1) & 1) * 2 * MMQ_TILEB, *Xn = Wn + MMQ_TILEB;
        __syncthreads();
        const MmqW<QT> stc = st;  // slot st is re-filled below (lw may alias it)
        const MmqX sxc = sx;
        mmq_load_w<QT>(lw, a, wrow, kb_of(s + 3), wj);
        mmq_load_x(lx, a, kb_of(s + 3), xrow, xc);
        const f16x2 dd = {stc.d, stc.d}, mm = {stc.m, stc.m};
        f16x8 fa[2][2], fb[2][2];
#pragma unroll
        for (int j = 0; j < 2; j++) fa[0][j] = *(const f16x8 *)(X + (wn * 64 + j * 32) * MMQ_ROWB + frag_off);
#pragma unroll
        for (int i = 0; i < 2; i++) fb[0][i] = *(const f16x8 *)(W + (wm * 64 + i * 32) * MMQ_ROWB + frag_off);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            u32x4 o;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int j = t >> 1, i = t & 1, cb = ks & 1, nb2 = cb ^ 1;
                acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cb][j], fb[cb][i], acc[j][i], 0, 0, 0);
                if (ks < 3) {  // fragments of k-step ks+1
                    if (t < 2)
                        fa[nb2][t] = *(const f16x8 *)(X + (wn * 64 + t * 32) * MMQ_ROWB + frag_off + (ks + 1) * 32);
                    else
                        fb[nb2][t - 2] = *(const f16x8 *)(W + (wm * 64 + (t - 2) * 32) * MMQ_ROWB + frag_off + (ks + 1) * 32);
                }
                o[t] = mmq_dequant_slice<QT>(stc.q, stc.q2, stc.qh, ks, t, dd, mm);
                if (t == 1) *(u32x4 *)(Xn + xoff + 32 * ks * MMQ_ROWB) = sxc.xa[ks];
                if (t == 3) *(u32x4 *)(Wn + woff + ks * 16) = o;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    if (nstage > 0) {
        // stage 0 straight into LDS buffer 0; stages 1..2 into the rings
        MmqW<QT> w0 = wr_[0];
        MmqX x0;
        mmq_load_w<QT>(w0, a, wrow, kb_of(0), wj);
        mmq_load_x(x0, a, kb_of(0), xrow, xc);
#pragma unroll
        for (int u = 0; u < 2; u++) mmq_load_w<QT>(wr_[u], a, wrow, kb_of(1 + u), wj);  // slot u <-> stage u+1
#pragma unroll
        for (int u = 0; u < 2; u++) mmq_load_x(xr_[u], a, kb_of(1 + u), xrow, xc);
        u32x4 o[4];
        mmq_dequant<QT>(w0.q, w0.q2, w0.qh, w0.d, w0.m, o);
#pragma unroll
        for (int k = 0; k < 4; k++) *(u32x4 *)(lds + woff + k * 16) = o[k];
#pragma unroll
        for (int i = 0; i < 4; i++) *(u32x4 *)(lds + MMQ_TILEB + xoff + 32 * i * MMQ_ROWB) = x0.xa[i];
    }
    // iteration s consumes ring slot s % 2 (it holds stage s+1) and re-fills it with stage s+3.  The steady-state
    // loop body is branch-free.  (A deeper ring — 4 weight stages — was measured and bought nothing: hipcc's
    // s_waitcnt insertion waits for vmcnt <= 5 at the top of every stage whatever the ring depth, so the effective
    // prefetch distance stays ~1 stage and the kernel remains latency-bound; WAIT_ANY = 43 % of wave cycles in
    // profiles/r01_run23_prefill_mmq_pmc.txt.  The way out is LDS-DMA staging with hand-placed counted waits.)
    int s = 0;
    for (; s + 2 <= nstage; s += 2) {
        stage(s, wr_[0], xr_[0], wr_[0], xr_[0]);
        stage(s + 1, wr_[1], xr_[1], wr_[1], xr_[1]);
    }
    if (s < nstage) stage(s, wr_[0], xr_[0], wr_[0], xr_[0]);

    // C layout of the 32x32 MFMA: column (B index = weight row) = lane & 31,
    // row (A index = token) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    // K split in two: either f32 atomic adds into a zeroed dst (two addends commute), or — split_stride != 0 — each
    // half stores its partial tile to its own buffer (dst + blockIdx.y * split_stride) and the consumer adds them
    const bool split = gridDim.y > 1 && a.split_stride == 0;
    float *const dstp = a.dst + (int64_t)blockIdx.y * a.split_stride;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int64_t m = m0 + wm * 64 + i * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t n = n0 + wn * 64 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < a.M && n < a.N) {
                    if (split)
                        unsafeAtomicAdd(a.dst + n * a.ldd + m, acc[j][i][r]);
                    else
                        dstp[n * a.ldd + m] = acc[j][i][r];
                }
            }
        }
}
