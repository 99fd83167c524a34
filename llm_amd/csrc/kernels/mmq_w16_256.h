// mmq_w16_256.h — the prompt GEMM on resident f16 weight copies with a 256 x 256 x 64 workgroup tile.
//
// Why a bigger tile (profiles/r02_prefill_mmq_w16_pmc.txt): k_mmq_w16_p8 (128 x 128 tile) keeps the
// matrix pipes 31 % busy with its vector-memory front end stalled on outstanding requests half of the time — a CU pulls
// ~20 bytes per clock through its L1 whatever the schedule, and a 128 x 128 x 64 stage needs 32 KB for 2.1 MFLOP.  A
// 256 x 256 x 64 stage needs 64 KB for 8.4 MFLOP: half the bytes per flop, and half the LDS fragment traffic per MFMA
// (each wave owns 128 tokens x 64 weight rows: 24 ds_read_b128 feed 32 MFMAs instead of 12 feeding 8).
//
// Schedule (the 8-wave "ping-pong" of cdna_hip_programming.md section 5, written for this kernel's operands):
//  * 8 waves = 2 groups of 4 (one wave of each group per SIMD).  A k-stage is 4 phases of 8 MFMAs; every phase is
//    {LOAD: fragment reads + one 16 KB LDS-DMA region -> s_barrier -> COMPUTE: 8 x v_mfma_f32_32x32x16_f16 -> s_barrier}.
//    Group B runs one barrier behind group A, so while one wave of a SIMD multiplies, the other one reads and requests.
//  * Two 64 KB stage buffers are the CU's LDS.  A buffer is not refilled as a whole: each of its four 16 KB regions
//    (the X rows / W rows one phase reads) is re-requested two phases after its last read, for the stage after next —
//    five regions (80 KB) are in flight per CU, every region has >= 4 phases to land, and the only waits are counted
//    (s_waitcnt vmcnt(8)), never a drain.
//  * LDS image as in k_mmq_w16_p8: 128-byte rows (64 k), 16-byte chunks XOR-swizzled by (row >> 1) & 7 on the SOURCE
//    address (the DMA writes lane-linear), conflict-free ds_read_b128 fragments.
//  * Persistent: one workgroup per CU walks (tile x K-split) items; the DMA cursor runs on into the next item.
// Same f16 values, same k order and the same MFMA per 16-wide k step as k_mmq_w16_p8 / k_mmq_dma_p8: for equal K splits the
// results are bit-identical (tests/test_prompt_plan_gpu.py).
#pragma once
#include "mmq_w16.h"

#define T256_TM 256
#define T256_TN 256
#define T256_X 0
#define T256_W 32768
#define T256_SLOT 65536
#define T256_LDS (2 * T256_SLOT)

// The MFMA builtins are pure functions to the compiler: without an anchor it sinks / hoists them across s_barrier and out
// of the s_setprio window.  An empty asm that "rewrites" an accumulator tile pins every MFMA on it between the asm before
// and the asm after (volatile asm statements and barriers keep their order).
#define T256_PIN(t) asm volatile("" : "+v"(t))

// VAR (variants measured neutral in round 3 and no longer instantiated; 0 = production): bit 0 = the counted DMA wait sits behind the phase's MFMAs
// (issued, still executing) instead of in front of the barrier that precedes them; bit 1 = no s_setprio around the MFMAs.
template <int VAR>
__global__ void __launch_bounds__(512, 2) k_mmq_w16_256(const MmqArgs a, int n_items, int tiles_total, int splits) {
    constexpr bool WAIT_LATE = (VAR & 1) != 0, PRIO = (VAR & 2) == 0;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ww = wave & 3, wt = wave >> 2;  // wave tile: tokens wt*128 .. +128, weight rows ww*64 .. +64
    const bool grp_b = wave >= 4;             // the group that runs one barrier behind
    const int nstage_all = (int)(a.nb >> 1);
    const int per = (nstage_all + splits - 1) / splits;

    struct Item {
        int64_t m0, n0, M, ldd;
        const _Float16 *w16;
        float *dst;
        int s_begin, nstage;
    };
    auto load_item = [&](int w, Item &it) {
        const int y = w / tiles_total, b = w - y * tiles_total;
        const int t = xcd_tile_id(b, tiles_total);
        int tm = t / a.tiles_n;
        const int tn = t - tm * a.tiles_n;
        const _Float16 *w16 = (const _Float16 *)a.w.w16;
        float *dst = a.dst;
        int64_t ldd = a.ldd, M = a.w.M;
        if (a.nseg > 1) {
            if (a.nseg > 2 && tm >= a.tile_end[1]) {
                tm -= a.tile_end[1];
                w16 = (const _Float16 *)a.wc.w16;
                dst = a.dst_c;
                ldd = a.ldd_c;
                M = a.wc.M;
            } else if (tm >= a.tile_end[0]) {
                tm -= a.tile_end[0];
                w16 = (const _Float16 *)a.wb.w16;
                dst = a.dst_b;
                ldd = a.ldd_b;
                M = a.wb.M;
            }
        }
        it.m0 = (int64_t)tm * T256_TM;
        it.n0 = (int64_t)tn * T256_TN;
        it.M = M;
        it.ldd = ldd;
        it.w16 = w16;
        it.dst = dst + (int64_t)y * a.split_stride;
        it.s_begin = y * per;
        it.nstage = min(nstage_all, it.s_begin + per) - it.s_begin;
    };
    // DMA regions of a stage buffer (16 KB = 128 rows each, 2 instructions per wave): X0 / X1 = the token rows the phases with
    // token half 0 / 1 read (rows h*64 .. +64 of each 128-row wave slab), W0 / W1 = the weight rows of weight half 0 / 1
    // (rows h*32 .. +32 of each 64-row wave slab).  Instruction i of wave w covers region rows 16w + 8i .. +8.
    auto x_row = [&](int h, int i) { const int rr = 16 * wave + 8 * i; return (rr & 63) + (rr >> 6) * 128 + 64 * h; };
    auto w_row = [&](int h, int i) { const int rr = 16 * wave + 8 * i; return (rr & 31) + (rr >> 5) * 64 + 32 * h; };
    struct LaneAddr {
        const char *x[2][2], *w[2][2];  // [half][instruction]
    };
    const int64_t row_bytes = a.nb * 64;
    auto lane_addr = [&](const Item &it, LaneAddr &A) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int rx = x_row(h, i) + (lane >> 3), rw = w_row(h, i) + (lane >> 3);
                A.x[h][i] = (const char *)a.x + min(it.n0 + rx, a.N - 1) * row_bytes + (((lane & 7) ^ ((rx >> 1) & 7)) << 4);
                A.w[h][i] = (const char *)it.w16 + min(it.m0 + rw, it.M - 1) * row_bytes + (((lane & 7) ^ ((rw >> 1) & 7)) << 4);
            }
    };

    // ---- the DMA cursor: stage `is` of item `iw`, global stage number gi (buffer = gi & 1)
    int iw = (int)blockIdx.x, is = 0, gi = 0;
    Item Ti;
    LaneAddr Ai;
    load_item(iw, Ti);
    lane_addr(Ti, Ai);
    auto advance = [&]() {
        gi++;
        if (is + 1 < Ti.nstage) {
            is++;
        } else if (iw + (int)gridDim.x < n_items) {
            iw += (int)gridDim.x;
            is = 0;
            load_item(iw, Ti);
            lane_addr(Ti, Ai);
        }  // past the last stage: the last one is requested again into a buffer nobody reads any more (keeps the counts)
    };
    auto issue_x = [&](int h) {
        const int64_t koff = (int64_t)(Ti.s_begin + is) * 128;
        char *slot = lds + (gi & 1) * T256_SLOT + T256_X;
#pragma unroll
        for (int i = 0; i < 2; i++)
            __builtin_amdgcn_global_load_lds((gptr_t)(Ai.x[h][i] + koff), (lptr_t)(slot + x_row(h, i) * 128), 16, 0, 0);
    };
    auto issue_w = [&](int h) {
        const int64_t koff = (int64_t)(Ti.s_begin + is) * 128;
        char *slot = lds + (gi & 1) * T256_SLOT + T256_W;
#pragma unroll
        for (int i = 0; i < 2; i++)
            __builtin_amdgcn_global_load_lds((gptr_t)(Ai.w[h][i] + koff), (lptr_t)(slot + w_row(h, i) * 128), 16, 0, 0);
    };

    // prologue: stage 0 complete, X0 / W0 of stage 1 — in the order the loop goes on requesting
    issue_x(0);
    issue_w(0);
    issue_w(1);
    issue_x(1);
    advance();
    issue_x(0);
    issue_w(0);

    // fragment addresses: row = slab base + (lane & 31), chunk (ks * 2 + (lane >> 5)) ^ swizzle(row); the slab bases are
    // multiples of 32 rows, so the swizzle term depends on the lane only
    const int fr = lane & 31, fh = lane >> 5, sw = (fr >> 1) & 7;
    int off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) off[ks] = fr * 128 + (((ks * 2 + fh) ^ sw) << 4);
    const int xbase = T256_X + wt * 128 * 128, wbase = T256_W + ww * 64 * 128;

    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");  // X0 and W0 of stage 0 landed, every wave's part
    if (grp_b) __builtin_amdgcn_s_barrier();

    f16x8 fa[2][4], fb[2][4];
    int gc = 0;
    for (int cw = (int)blockIdx.x; cw < n_items; cw += (int)gridDim.x) {
        const int c_y = cw / tiles_total, c_begin = c_y * per;
        const int c_nstage = min(nstage_all, c_begin + per) - c_begin;
        f32x16 acc[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.0f;
        for (int s = 0; s < c_nstage; s++, gc++) {
            const char *S = lds + (gc & 1) * T256_SLOT;
            // ---- phase 0: tokens half 0 x weights half 0
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                fb[0][ks] = *(const f16x8 *)(S + wbase + off[ks]);
                fa[0][ks] = *(const f16x8 *)(S + xbase + off[ks]);
                fa[1][ks] = *(const f16x8 *)(S + xbase + 32 * 128 + off[ks]);
            }
            issue_w(1);
            if (!WAIT_LATE) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            T256_PIN(acc[0][0]);
            T256_PIN(acc[1][0]);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][ks], fb[0][ks], acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][ks], fb[0][ks], acc[1][0], 0, 0, 0);
            }
            T256_PIN(acc[0][0]);
            T256_PIN(acc[1][0]);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            if (WAIT_LATE) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase 1: tokens half 0 x weights half 1
#pragma unroll
            for (int ks = 0; ks < 4; ks++) fb[1][ks] = *(const f16x8 *)(S + wbase + 32 * 128 + off[ks]);
            issue_x(1);
            if (!WAIT_LATE) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            T256_PIN(acc[0][1]);
            T256_PIN(acc[1][1]);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][ks], fb[1][ks], acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][ks], fb[1][ks], acc[1][1], 0, 0, 0);
            }
            T256_PIN(acc[0][1]);
            T256_PIN(acc[1][1]);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            if (WAIT_LATE) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase 2: tokens half 1 x weights half 1
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                fa[0][ks] = *(const f16x8 *)(S + xbase + 64 * 128 + off[ks]);
                fa[1][ks] = *(const f16x8 *)(S + xbase + 96 * 128 + off[ks]);
            }
            advance();
            issue_x(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            T256_PIN(acc[2][1]);
            T256_PIN(acc[3][1]);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][ks], fb[1][ks], acc[2][1], 0, 0, 0);
                acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][ks], fb[1][ks], acc[3][1], 0, 0, 0);
            }
            T256_PIN(acc[2][1]);
            T256_PIN(acc[3][1]);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---- phase 3: tokens half 1 x weights half 0 (both fragment sets are in registers)
            issue_w(0);
            if (!WAIT_LATE) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // X0 and W0 of the next stage
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            T256_PIN(acc[2][0]);
            T256_PIN(acc[3][0]);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][ks], fb[0][ks], acc[2][0], 0, 0, 0);
                acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][ks], fb[0][ks], acc[3][0], 0, 0, 0);
            }
            T256_PIN(acc[2][0]);
            T256_PIN(acc[3][0]);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            if (WAIT_LATE) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the item's outputs: D rows (registers) = tokens, D columns (lanes & 31) = weight rows
        Item Tc;
        load_item(cw, Tc);
        float *const d0 = Tc.dst + (Tc.n0 + wt * 128 + 4 * (lane >> 5)) * Tc.ldd + Tc.m0 + ww * 64 + (lane & 31);
        const bool full = Tc.m0 + T256_TM <= Tc.M && Tc.n0 + T256_TN <= a.N;
        if (full && !(splits > 1 && a.split_stride == 0)) {  // the common case: no bounds, plain stores
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int mt = 0; mt < 4; mt++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        d0[(int64_t)(mt * 32 + (r & 3) + 8 * (r >> 2)) * Tc.ldd + nt * 32] = acc[mt][nt][r];
        } else {
            const bool atomic = splits > 1 && a.split_stride == 0;
            const int64_t m_lim = Tc.M - (Tc.m0 + ww * 64 + (lane & 31)), n_lim = a.N - (Tc.n0 + wt * 128 + 4 * (lane >> 5));
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int mt = 0; mt < 4; mt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int dn = mt * 32 + (r & 3) + 8 * (r >> 2);
                        if (nt * 32 < m_lim && dn < n_lim) {
                            float *q = d0 + (int64_t)dn * Tc.ldd + nt * 32;
                            if (atomic)
                                unsafeAtomicAdd(q, acc[mt][nt][r]);
                            else
                                *q = acc[mt][nt][r];
                        }
                    }
        }
    }
    if (!grp_b) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
