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
// Work split (round 3, after the in-kernel timeline of tests/tools/cols_timeline.py): the workgroup (8 waves) owns a
// contiguous range of 16-row groups; wave w owns the K range [K w / 8, K (w + 1) / 8) (in steps of 4 blocks) of EVERY group
// of the workgroup: a unit = (group, matrix) costs every wave the same number of steps, nothing is divided at run time, and
// the eight partial sums of a row meet in LDS in wave order (deterministic).  Then the epilogues of decode_big8.h run
// (store / +residual / silu(w1 x) * w3 x / RoPE + K,V store per token).
// Staging: the activation planes and the [block][8] scale / sum tables (written in that layout by the producing kernels:
// ColsArgs::dxT, sxT) go to LDS by LDS-DMA in front of the weight ring; the first barrier waits for them only
// (s_waitcnt vmcnt(ring loads)), not for the ring (12.5 MB chip-wide: 2.7 us at HBM speed).
#pragma once
#include "decode_big8.h"
#include "mmq.h"

struct ColsArgs {
    DecMmvqArgs d;     // d.x: Q8 rows [ncols][nb] (planar), d.dst / d.res: row 0
    int ncols;         // 1..8 columns of this pass
    int col0;          // first column of the pass within the chunk (EPI_QKV: K / V positions are n_past + col0 + c)
    int64_t ldd, ldr;  // floats between consecutive rows of dst / res
    const float *rope; // EPI_QKV: (cos, sin) tables of the chunk's positions, 128 floats per token (k_rope_table)
    const float *dxT;  // activation scales transposed: [nb][8] (column c of block b at b * 8 + c; zeros for c >= ncols)
    const int *sxT;    // activation quant sums, same layout
    int ngroups;       // 16-row groups (EPI_GATE: pairs of a w1 and a w3 group) in the launch
    int gq, gr;        // ngroups / gridDim.x and ngroups % gridDim.x: workgroup b owns gq (+1 if b < gr) groups from b * gq + min(b, gr)
    long long *ts;     // INSTR build only (option "timeline"): 8 x int64 per sampled workgroup, as BigArgs::ts
    int ts_wgs;
    const void *hot;   // 256 bytes the dummy ring steps read (BigArgs::hot, decode_big.h); nullptr = the first bytes of the scales
};

typedef int i32x4v __attribute__((ext_vector_type(4)));

#define COLS_T 512
#define COLS_W 8
#define COLS_MAX_UNITS 12  /* (group, matrix) units of a workgroup: 8 x 512 B of partial sums each */

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

// INSTR: the measurement build (timeline stamps of wave 0 in sampled workgroups: tests/tools/cols_timeline.py), launched only
// while option "timeline" is set; the production instantiation carries none of it.
template <int QT, int EPI, bool INSTR = false>
__global__ void __launch_bounds__(COLS_T) k_mmq_cols(const ColsArgs ca) {
    const DecMmvqArgs &a = ca.d;
    long long t_in = 0, t_pre = 0, t_dma = 0, t_issued = 0, t_staged = 0, t_loop = 0, t_sync2 = 0;
    if constexpr (INSTR) t_in = big_now();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NSUB = EPI == EPI_GATE ? 2 : 1;  // weight matrices a group draws rows from (w1 and w3)
    constexpr int LPS = QT == QT_Q4_0 ? 2 : QT == QT_Q5_1 ? 4 : 3;  // VMEM loads of one ring step (issue below)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncols = ca.ncols, nb = (int)a.nb;
    // LDS: activation planes [8][nb] x 16 B (low / high halves), per-block scales and quant sums [nb][8],
    // the partial sums [unit][wave][16 rows][8 columns]
    i32x4 *s_lo = (i32x4 *)smem;
    i32x4 *s_hi = s_lo + 8 * nb;
    float *s_dx = (float *)(s_hi + 8 * nb);
    int *s_sx = (int *)(s_dx + 8 * nb);
    float *s_part = (float *)(s_sx + 8 * nb);

    // ---- this workgroup's groups, this wave's K range
    const int bid = (int)blockIdx.x;
    const int g_begin = bid * ca.gq + min(bid, ca.gr);
    const int nlg = ca.gq + (bid < ca.gr ? 1 : 0);  // groups here (no run-time division: the prologue is a serial scalar chain)
    const int nu = nlg * NSUB;         // units: <= COLS_MAX_UNITS (launcher)
    const int nst_all = nb >> 2;       // steps of 4 blocks in a row
    const int st0 = (nst_all * wave) >> 3, st1 = (nst_all * (wave + 1)) >> 3;
    const int nst = st1 - st0;         // >= 1 (launcher: nb >= 32)
    // the weight records in registers up front (one batch of kernarg loads instead of a load + wait inside every unit switch)
    // (plain scalars: a copy of the QWeight structs lands in scratch)
    constexpr int I1 = EPI == EPI_GATE ? 1 : 0;
    const uint8_t *const qsA = a.w[0].qs, *const qsB = a.w[I1].qs;
    const uint8_t *const qs2A = a.w[0].qs2, *const qs2B = a.w[I1].qs2;
    const uint32_t *const qhA = a.w[0].qh, *const qhB = a.w[I1].qh;
    const __half *const wdA = a.w[0].d, *const wdB = a.w[I1].d;
    const __half *const wmA = a.w[0].m, *const wmB = a.w[I1].m;
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

    // every kernel argument the prologue needs, fetched as ONE batch of scalar loads: left to itself the compiler loads each
    // next to its first use, and the prologue becomes a chain of four ~0.35 us kernarg round trips (in-kernel timeline)
    {
        const void *p0 = a.x.lo, *p1 = a.x.hi, *p2 = ca.dxT, *p3 = ca.sxT;
        asm volatile("" ::"s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(nb), "s"(ncols), "s"(ca.gq), "s"(ca.gr), "s"(M0), "s"(M1));
        if constexpr (EPI != EPI_QKV) {
            const void *p4 = qsA, *p5 = qsB, *p6 = wdA, *p7 = wdB, *p8 = a.dst, *p9 = a.res;
            asm volatile("" ::"s"(p4), "s"(p5), "s"(p6), "s"(p7), "s"(p8), "s"(p9), "s"(ca.ldd), "s"(ca.ldr));
            if constexpr (QT == QT_Q8_0) asm volatile("" ::"s"(qs2A), "s"(qs2B));
            if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) asm volatile("" ::"s"(qhA), "s"(qhB));
            if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1) asm volatile("" ::"s"(wmA), "s"(wmB));
        }
    }
    if constexpr (INSTR) t_pre = big_now();
    // ---- activations -> LDS by DMA: every plane has its LDS layout in global memory, one instruction per 1 KB
    {
        const int nq = ncols * nb;  // 16-byte items of a quant plane (columns >= ncols stay as they are: never stored)
        for (int i0 = wave * 64; i0 < nq; i0 += COLS_T) {  // wave-uniform
            if (i0 + lane < nq) {
                __builtin_amdgcn_global_load_lds((gptr_t)(a.x.lo + i0 + lane), (lptr_t)(s_lo + i0), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(a.x.hi + i0 + lane), (lptr_t)(s_hi + i0), 16, 0, 0);
            }
        }
        const int nt = 2 * nb;  // 16-byte items of a [nb][8] table
        for (int i0 = wave * 64; i0 < nt; i0 += COLS_T) {
            if (i0 + lane < nt) {
                __builtin_amdgcn_global_load_lds((gptr_t)((const f32x4 *)ca.dxT + i0 + lane), (lptr_t)((f32x4 *)s_dx + i0), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)((const f32x4 *)ca.sxT + i0 + lane), (lptr_t)((f32x4 *)s_sx + i0), 16, 0, 0);
            }
        }
    }
    asm volatile("" ::: "memory");  // the ring's loads stay behind the DMA (the wait below counts on it)
    if constexpr (INSTR) t_dma = big_now();

    // ---- the wave's steps: unit q = 0 .. nu-1, within a unit the steps st0 .. st1-1; a weight ring across them
    const int jrow = lane & 15, bg = lane >> 4;
    const bool act = ((lane & 15) >> 2) == bg;  // A operand: this lane's row carries a column of its own k group
    const int acol = lane & 3;
    struct Cursor {
        int q, st;
        const uint8_t *qs, *qs2;
        const uint32_t *qh;
        const __half *wd, *wm;
        uint32_t row_blk;  // (row of this lane) * nb
    };
    auto open_unit = [&](Cursor &c, int q) {
        c.q = q;
        c.st = st0;
        if (q >= nu) return;
        const int sub = q & (NSUB - 1), lg = NSUB == 2 ? q >> 1 : q;
        int sg, m0;
        group_rows(g_begin + lg, sub, sg, m0);
        // scalar selects (sg is wave-uniform): a kernarg array indexed by a "divergent" id is fetched with vector loads
        if constexpr (EPI == EPI_QKV) {  // three matrices and the longest epilogue: hoisted copies overflow the SGPRs into scratch
            c.qs = sg == 0 ? a.w[0].qs : sg == 1 ? a.w[1].qs : a.w[2].qs;
            if constexpr (QT == QT_Q8_0) c.qs2 = sg == 0 ? a.w[0].qs2 : sg == 1 ? a.w[1].qs2 : a.w[2].qs2;
            if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) c.qh = sg == 0 ? a.w[0].qh : sg == 1 ? a.w[1].qh : a.w[2].qh;
            c.wd = sg == 0 ? a.w[0].d : sg == 1 ? a.w[1].d : a.w[2].d;
            if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1) c.wm = sg == 0 ? a.w[0].m : sg == 1 ? a.w[1].m : a.w[2].m;
        } else {
            c.qs = sg == 0 ? qsA : qsB;
            if constexpr (QT == QT_Q8_0) c.qs2 = sg == 0 ? qs2A : qs2B;
            if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) c.qh = sg == 0 ? qhA : qhB;
            c.wd = sg == 0 ? wdA : wdB;
            if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1) c.wm = sg == 0 ? wmA : wmB;
        }
        c.row_blk = (uint32_t)(m0 + jrow) * (uint32_t)nb;
    };
    // A step past the wave's last one is a DUMMY that reads ColsArgs::hot (one line, a cache hit): every ring slot is refilled
    // unconditionally, so the loads in flight are a compile-time constant at every wait (k_mmvq_big's `issue`, decode_big.h).
    // Round 3's form — refills under `if (pc.q < nu)` — compiled to ONE s_waitcnt vmcnt(0) at the head of every pass of PF
    // steps: the whole ring drained, then PF steps of arithmetic with the next batch in flight (tests/tools/disasm.py, round 6).
    const uint8_t *const hotp = ca.hot ? (const uint8_t *)ca.hot : (const uint8_t *)a.w[0].d;
    auto issue = [&](ColsStep<QT> &s, const Cursor &c, const bool dummy) {
        const uint32_t o = dummy ? 0u : c.row_blk + (uint32_t)(4 * c.st + bg);
        const uint8_t *pq = dummy ? hotp : c.qs;
        s.q = __builtin_nontemporal_load((const u32x4 *)(pq + (size_t)o * 16));
        if constexpr (QT == QT_Q8_0) s.p = __builtin_nontemporal_load((const u32x4 *)((dummy ? hotp : c.qs2) + (size_t)o * 16));
        if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) s.h = __builtin_nontemporal_load((dummy ? (const uint32_t *)hotp : c.qh) + o);
        s.dw = (dummy ? (const __half *)hotp : c.wd)[o];
        if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1) s.mw = (dummy ? (const __half *)hotp : c.wm)[o];
    };
    auto advance = [&](Cursor &c) {
        if (++c.st >= st1) open_unit(c, c.q + 1);
    };
    // ring depth: with counted waits the depth is real, and deeper is SLOWER here (8 waves x PF steps of 1 KB + the scales compete
    // for the CU's request slots with each other): LLaMA-7B Q4_0 prompt feed at n_batch = 8, -DCOLS_PF / -DCOLS_PF0 builds on one
    // box (gpurun_out/r6/run31, run32), tok/s: 12|2 2913, 10|2 2899, 8|2 3021, 6|2 3160, 5|2 3190, 4|2 3299, 4|1 3292, 3|2 3323,
    // 3|1 3358, 2|2 3249 (round 5: 6 steps, all requested in front of the barrier, drained at every pass: 3100)
#ifndef COLS_PF
#define COLS_PF 3
#endif
    constexpr int PF = QT == QT_Q8_0 ? (COLS_PF * 2 + 2) / 3 : COLS_PF;
    ColsStep<QT> ring[PF];
    Cursor pc;  // producer
    open_unit(pc, 0);
    // Only PF0 steps go out in front of the staging barrier.  A CU accepts only so many outstanding requests: with the whole ring
    // (6 steps x 8 waves, ~100 KB) requested first, the last waves sat in ISSUE until the first HBM lines came back and reached
    // the barrier 2 us after wave 0 had its activations (in-kernel probe, round 6: DMA landed 2.0 us after entry, barrier passed
    // at 4.0) — the lesson of k_mmvq_big's PF0 (decode_big.h), learnt again.
#ifndef COLS_PF0
#define COLS_PF0 1
#endif
    constexpr int PF0 = PF < COLS_PF0 ? PF : COLS_PF0;
#pragma unroll
    for (int k = 0; k < PF0; k++) {
        const bool more = pc.q < nu;
        issue(ring[k], pc, !more);
        if (more) advance(pc);
    }
    if constexpr (INSTR) t_issued = big_now();
    // the DMA is older than the ring's loads, and the ring steps requested so far are always PF0 (dummies where the wave has fewer):
    // "at most PF0 * LPS outstanding" means the DMA has landed
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF0 * LPS) : "memory");
    __syncthreads();  // activations staged
    if constexpr (INSTR) t_staged = big_now();
#pragma unroll
    for (int k = PF0; k < PF; k++) {
        const bool more = pc.q < nu;
        issue(ring[k], pc, !more);
        if (more) advance(pc);
    }

    int cq = 0, cst = st0;  // consumer
    float f[8];
#pragma unroll
    for (int c = 0; c < 8; c++) f[c] = 0.0f;
    auto consume = [&](const ColsStep<QT> &st) {
        const int bl = 4 * cst + bg;  // this lane's block (weights, scales, outputs)
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
        if (++cst >= st1) {  // the wave's part of the unit is complete: add the four block groups, park the 16 x 8 sums
#pragma unroll
            for (int c = 0; c < 8; c++) {
                float v = f[c];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                f[c] = v;
            }
            if (lane < 16) {
                float *o = s_part + ((size_t)(cq * COLS_W + wave) * 16 + lane) * 8;
                *(f32x4 *)o = f32x4{f[0], f[1], f[2], f[3]};
                *(f32x4 *)(o + 4) = f32x4{f[4], f[5], f[6], f[7]};
            }
#pragma unroll
            for (int c = 0; c < 8; c++) f[c] = 0.0f;
            cst = st0;
            cq++;
        }
    };
    // full passes over the ring while a pass still has a step to request (every slot: its refill goes out unconditionally, then its
    // arithmetic), then a pass that drains the last PF steps and requests nothing: static wait counts throughout
    const int Ttot = nu * nst;
    int tdone = 0;
    for (; tdone + PF < Ttot; tdone += PF) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            consume(ring[k]);  // (then the refill INTO the registers just consumed: a copy of the slot taken first makes the
            const bool more = pc.q < nu;  //  loop's back edge a register shuffle behind an s_waitcnt vmcnt(0))
            issue(ring[k], pc, !more);
            if (more) advance(pc);
        }
    }
#pragma unroll
    for (int k = 0; k < PF; k++)
        if (tdone + k < Ttot) consume(ring[k]);  // wave-uniform
    if constexpr (INSTR) t_loop = big_now();
    __syncthreads();
    if constexpr (INSTR) t_sync2 = big_now();

    // ---- epilogues: thread per (group, row or row pair, column); the eight waves' partials added in wave order
    auto total = [&](int lg, int sub, int row, int c) {
        const float *p = s_part + ((size_t)((lg * NSUB + sub) * COLS_W) * 16 + row) * 8 + c;
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < COLS_W; k++) v += p[(size_t)k * 128];
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
            const int p = n_past + ca.col0 + c;
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
    if constexpr (INSTR) {
        const int every = ca.ts_wgs > 0 ? max(1, (int)gridDim.x / ca.ts_wgs) : 0;
        if (ca.ts && every && tid == 0 && (int)blockIdx.x % every == 0 && (int)blockIdx.x / every < ca.ts_wgs) {
            long long *o = ca.ts + (size_t)((int)blockIdx.x / every) * 8;
            o[0] = t_in; o[1] = t_pre; o[2] = t_dma; o[3] = t_issued; o[4] = t_staged; o[5] = t_loop; o[6] = t_sync2; o[7] = big_now();
        }
    }
}
