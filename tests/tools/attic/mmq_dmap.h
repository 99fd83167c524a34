// mmq_dmap.h — k_mmq_dma as a PERSISTENT kernel: one workgroup per CU walks a list of work items (tile x K split),
// and the LDS-DMA ring, the dequantization pipeline and the matrix pipe keep running across item boundaries.
//
// Why (rocprofv3 per-shape durations of k_mmq_dma at 512 tokens, profiles/r02_prefill_shapes.txt): E x E split in two
// (256 workgroups x 32 k-stages) 34.8 us, w1 (344 tiles, 2 rounds x 64 stages) 107.9, w1|w3 (688, 3 rounds) 167.5, lm_head
// (1000, 4 rounds) 216.9 — a straight line  duration = 3 + rounds x (15 + 0.6 x stages)  us.  A k-stage costs 0.6 us,
// but every tile pays ~15 us on top: the workgroup's dispatch (its 140 KB of LDS is only free once its predecessor on
// that CU has completely finished), the first DMA round trip with nothing to overlap it, and the drain + 64 KB of
// result stores with the matrix pipe idle.  At K = 4096 that is 28 % of a tile's time, for a split E x E tile 44 %.
// Here the fixed cost is paid once per launch: while the last stages of item i are multiplied, the DMA of item i+1's
// first stages is already landing (the issue cursor simply walks on into the next item), its first weight block is
// dequantized by the regular pipeline step, and the result stores of item i drain under item i+1's first stages.
//
// Same tile, ring, arithmetic and per-tile summation order as k_mmq_dma: results are bit-identical.  Work item w ->
// (split y = w / tiles, tile b = w % tiles), b -> (tm, tn) by the mapping k_mmq_dma applies to blockIdx.x; workgroup g
// takes items g, g + gridDim.x, ...
//
// Counted waits: G DMA instructions per stage and wave, three stages in flight; a wait for vmcnt <= G retires everything
// older than the newest group.  The 64 result stores of an item are older than the groups issued after them, so the
// first wait after an item boundary also waits for those stores to reach L2 (~1 us): the one bubble left per item.
#pragma once
#include "mmq_dma.h"

template <int QT>
__global__ void __launch_bounds__(256, 1) k_mmq_dma_p(const MmqArgs a, int n_items, int tiles_total, int splits) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    constexpr int G = dma_group<QT, false>();
    constexpr int SLOT = DMA_SLOT, WT = DMA_WT;
    const int nstage_all = (int)(a.nb >> 1);
    const int per = (nstage_all + splits - 1) / splits;

    // ---- work item -> everything that is uniform over the workgroup
    struct Item {
        int64_t m0, n0, M, ldd;
        const uint8_t *qs, *qs2;
        const uint32_t *qh;
        const __half *d, *m;
        float *dst;
        int s_begin, nstage;
    };
    auto load_item = [&](int w, Item &it) {
        const int y = w / tiles_total, b = w - y * tiles_total;
        int tm, tn;
        if (a.xcd_by_n) {
            const int tiles_m = tiles_total / a.tiles_n;
            tn = b / tiles_m;
            tm = b - tn * tiles_m;
        } else {
            const int t = xcd_tile_id(b, tiles_total);
            tm = t / a.tiles_n;
            tn = t - tm * a.tiles_n;
        }
        QWeight w_ = a.w;
        float *dst = a.dst;
        int64_t ldd = a.ldd;
        if (a.nseg > 1) {
            if (a.nseg > 2 && tm >= a.tile_end[1]) {
                tm -= a.tile_end[1];
                w_ = a.wc;
                dst = a.dst_c;
                ldd = a.ldd_c;
            } else if (tm >= a.tile_end[0]) {
                tm -= a.tile_end[0];
                w_ = a.wb;
                dst = a.dst_b;
                ldd = a.ldd_b;
            }
        }
        it.m0 = (int64_t)tm * MMQ_TM;
        it.n0 = (int64_t)tn * MMQ_TN;
        it.M = w_.M;
        it.ldd = ldd;
        it.qs = w_.qs; it.qs2 = w_.qs2; it.qh = w_.qh; it.d = w_.d; it.m = w_.m;
        it.dst = dst + (int64_t)y * a.split_stride;
        it.s_begin = y * per;
        it.nstage = min(nstage_all, it.s_begin + per) - it.s_begin;
    };
    // ---- per-lane DMA source addresses of an item (bytes), advanced by a fixed stride per stage
    struct LaneAddr {
        const char *xsrc[4];
        const char *wq, *wq2, *wh, *wd, *wm_;
    };
    auto lane_addr = [&](const Item &it, LaneAddr &A) {
        // X: instruction i (0..3) of wave w covers rows 32w + 8i .. +7; lane -> row +(lane>>3), physical chunk lane&7
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 32 * wave + 8 * i + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            A.xsrc[i] = (const char *)(a.x + min(it.n0 + r, a.N - 1) * (a.nb * 32)) + c * 16;
        }
        // W: lane -> row 32w + (lane>>1), block lane&1 of the stage; scales: lane & 31 -> row 32w + lane, both blocks in one dword
        const int64_t wrow = min(it.m0 + 32 * wave + (lane >> 1), it.M - 1);
        const int64_t wblk0 = wrow * a.nb + (lane & 1);
        const int64_t drow = min(it.m0 + 32 * wave + (lane & 31), it.M - 1) * a.nb;
        A.wq = (const char *)it.qs + wblk0 * 16;
        A.wq2 = (const char *)it.qs2 + wblk0 * 16;
        A.wh = (const char *)it.qh + wblk0 * 4;
        A.wd = (const char *)it.d + drow * 2;
        A.wm_ = (const char *)it.m + drow * 2;
    };

    // ---- issue cursor: the next (item, stage) whose operands are requested
    int iw = (int)blockIdx.x, is = 0, gi = 0;  // item, stage inside it, global stage number (ring slot = gi & 3)
    Item Ti;
    LaneAddr Ai;
    load_item(iw, Ti);
    lane_addr(Ti, Ai);
    auto issue = [&]() {
        const int64_t kb = (int64_t)(Ti.s_begin + is) * 2;  // first block of the stage
        char *slot = lds + (gi & (DMA_RING - 1)) * SLOT;
#pragma unroll
        for (int i = 0; i < 4; i++)
            __builtin_amdgcn_global_load_lds((gptr_t)(Ai.xsrc[i] + kb * 64), (lptr_t)(slot + DMA_XS + (32 * wave + 8 * i) * 128), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(Ai.wq + kb * 16), (lptr_t)(slot + DMA_WQ + wave * 1024), 16, 0, 0);
        if constexpr (QT == QT_Q8_0)
            __builtin_amdgcn_global_load_lds((gptr_t)(Ai.wq2 + kb * 16), (lptr_t)(slot + DMA_WQ2 + wave * 1024), 16, 0, 0);
        if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1)
            __builtin_amdgcn_global_load_lds((gptr_t)(Ai.wh + kb * 4), (lptr_t)(slot + DMA_WH + wave * 256), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(Ai.wd + kb * 2), (lptr_t)(slot + DMA_WD + wave * 256), 4, 0, 0);
        if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1)
            __builtin_amdgcn_global_load_lds((gptr_t)(Ai.wm_ + kb * 2), (lptr_t)(slot + DMA_WM + wave * 256), 4, 0, 0);
        gi++;
        if (is + 1 < Ti.nstage) {
            is++;
        } else if (iw + (int)gridDim.x < n_items) {  // walk on into the next item of this workgroup
            iw += (int)gridDim.x;
            is = 0;
            load_item(iw, Ti);
            lane_addr(Ti, Ai);
        }  // else: past the end — the last stage is requested again (lands in a slot nobody reads)
    };

    // dequant assignment: one block per thread (row wr, block wj of the stage)
    const int wr = tid >> 1, wj = tid & 1;
    const int woff = wr * MMQ_ROWB + wj * 64;
    auto raw_load = [&](int s, u32x4 &q, u32x4 &q2, uint32_t &qh, _Float16 &d, _Float16 &m) {
        const char *slot = lds + (s & (DMA_RING - 1)) * SLOT;
        q = *(const u32x4 *)(slot + DMA_WQ + tid * 16);
        q2 = q;
        qh = 0;
        m = (_Float16)0.0f;
        if constexpr (QT == QT_Q8_0) q2 = *(const u32x4 *)(slot + DMA_WQ2 + tid * 16);
        if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) qh = *(const uint32_t *)(slot + DMA_WH + tid * 4);
        d = *(const _Float16 *)(slot + DMA_WD + (wr >> 5) * 256 + (wr & 31) * 4 + wj * 2);
        if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1)
            m = *(const _Float16 *)(slot + DMA_WM + (wr >> 5) * 256 + (wr & 31) * 4 + wj * 2);
    };

    // ---- prologue: groups 0, 1, 2 in flight; global stage 0's weights dequantized into W tile 0
    issue();
    issue();
    issue();
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * G) : "memory");  // group 0 landed, every wave's part
    {
        u32x4 q, q2, o[4];
        uint32_t qh;
        _Float16 d, m;
        raw_load(0, q, q2, qh, d, m);
        mmq_dequant<QT>(q, q2, qh, d, m, o);
#pragma unroll
        for (int k = 0; k < 4; k++) *(u32x4 *)(lds + WT + woff + k * 16) = o[k];
    }

    // Two nested loops — items, then the item's stages — rather than one flat loop with the result stores under a
    // condition: with the accumulators live across such a branch hipcc copies all 64 of them between AGPRs and VGPRs in
    // EVERY stage (128 v_accvgpr moves per stage; measured 14 % slower than one workgroup per tile).  g = global stage
    // number: ring slots and the W-tile parity run on across items.
    const int frow_x = lane & 31, fh = lane >> 5;
    int g = 0;
    for (int cw = (int)blockIdx.x; cw < n_items; cw += (int)gridDim.x) {
      Item Tc;
      load_item(cw, Tc);
      f32x16 acc[2][2];
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
              for (int r = 0; r < 16; r++) acc[j][i][r] = 0.0f;
      for (int s = 0; s < Tc.nstage; s++, g++) {
        // group g+1 landed (group g+2 may still be in flight), this wave's W-tile writes of the previous stage done
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(G) : "memory");
        issue();  // global stage g+3 into the slot of stage g-1: every wave is past its MFMAs
        const char *X = lds + (g & (DMA_RING - 1)) * SLOT + DMA_XS;
        const char *W = lds + WT + (g & 1) * MMQ_TILEB;
        char *Wn = lds + WT + ((g + 1) & 1) * MMQ_TILEB;
        u32x4 q, q2;
        uint32_t qh;
        _Float16 d, m;
        raw_load(g + 1, q, q2, qh, d, m);  // stage g+1: the next item's first stage at an item boundary
        const f16x2 dd = {d, d}, mm = {m, m};
        f16x8 fa[2][2], fb[2][2];
        auto xfrag = [&](int j, int ks) {
            const int R = wn * 64 + j * 32 + frow_x;
            const int p = (ks * 2 + fh) ^ ((R >> 1) & 7);
            return *(const f16x8 *)(X + R * 128 + p * 16);
        };
        auto wfrag = [&](int i, int ks) {
            return *(const f16x8 *)(W + (wm * 64 + i * 32 + frow_x) * MMQ_ROWB + ks * 32 + fh * 16);
        };
#pragma unroll
        for (int j = 0; j < 2; j++) fa[0][j] = xfrag(j, 0);
#pragma unroll
        for (int i = 0; i < 2; i++) fb[0][i] = wfrag(i, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            u32x4 o;
#pragma unroll
            for (int t2 = 0; t2 < 4; t2++) {
                const int j = t2 >> 1, i = t2 & 1, cb = ks & 1, nb2 = cb ^ 1;
                acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cb][j], fb[cb][i], acc[j][i], 0, 0, 0);
                if (ks < 3) {
                    if (t2 < 2)
                        fa[nb2][t2] = xfrag(t2, ks + 1);
                    else
                        fb[nb2][t2 - 2] = wfrag(t2 - 2, ks + 1);
                }
                o[t2] = mmq_dequant_slice<QT>(q, q2, qh, ks, t2, dd, mm);
                if (t2 == 3) *(u32x4 *)(Wn + woff + ks * 16) = o;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
      }
      // the item's tile is complete
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
          for (int i = 0; i < 2; i++) {
              const int64_t mrow = Tc.m0 + wm * 64 + i * 32 + (lane & 31);
#pragma unroll
              for (int r = 0; r < 16; r++) {
                  const int64_t n = Tc.n0 + wn * 64 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                  if (mrow < Tc.M && n < a.N) {
                      if (splits > 1 && a.split_stride == 0)
                          unsafeAtomicAdd(Tc.dst + n * Tc.ldd + mrow, acc[j][i][r]);
                      else
                          Tc.dst[n * Tc.ldd + mrow] = acc[j][i][r];
                  }
              }
          }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail DMAs before the workgroup's LDS is released
}
