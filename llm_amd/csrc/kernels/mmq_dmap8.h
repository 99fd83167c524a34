// mmq_dmap8.h — k_mmq_dma_p8: the persistent prompt GEMM on the QUANTIZED blocks (no resident f16 copy: HBM full, option
// mmq_w16 = 0): 128 x 128 x 64 tile, eight waves per workgroup (two per SIMD), one workgroup per CU walking the
// (tile x K split) items.  Every global operand byte — the f16 activations, the raw quant nibbles, the block scales — goes
// global -> LDS by DMA (global_load_lds_dwordx4) into a ring of slots several k-stages ahead; the only waits are hand-placed
// `s_waitcnt vmcnt(G)` (G = DMA instructions per stage and wave) in front of a raw s_barrier; the weights of the next stage
// are dequantized slot -> VALU -> padded f16 tile (144-byte rows: conflict-free fragment reads) while the MFMAs of the
// current one run.  The DMA ring, the dequantization pipeline and the matrix pipe keep running across item boundaries: the
// fixed ~15 us a one-workgroup-per-tile launch paid per tile (dispatch, first DMA round trip, drain + result stores with the
// matrix pipe idle; profiles/r02_prefill_shapes.txt) is paid once per launch.
//
// History (tests/tools/attic/mmq_dma.h, mmq_dmap.h — retired in round 4): k_mmq_dma was this tile with one workgroup per
// tile and four waves, k_mmq_dma_p its persistent form.  With one wave per SIMD every stall of that wave — the LDS round trip
// in front of each group of MFMAs, the barrier skew, the scalar work of the DMA issue — idled the SIMD's matrix pipe (a
// k-stage took ~1440 clocks for 512 clocks of MFMA).  The LDS footprint is per workgroup, not per wave: the same ring serves
// 512 threads; each wave owns a 64 x 32 piece of the tile (8 MFMAs per stage and wave), dequantizes half a block per stage,
// requests a quarter of the DMA traffic — and the SIMD has a second wave to issue from while the first one waits.  Per output
// element the products are accumulated in the same k order as in every other f16 GEMM here: bit-identical for equal K splits.
//
// Ring slot: X 128 tokens x 128 B, chunk-XOR-swizzled (the DMA writes lane-linear, so the swizzle is applied to each lane's
// SOURCE address: physical chunk p of row r holds logical chunk p ^ ((r >> 1) & 7)); Wq [128 rows][2 blocks][16 B] (+ Wq2
// for Q8_0), Wh [row][blk] u32 (Q5), Wd / Wm [row][2] f16 (one aligned 4-byte DMA per row: needs K/32 even).
#pragma once
#include "mmq.h"

// Ring slot of k_mmq_dma_p8, packed per weight type: the X tile, then only the parts the type has.  The 4- and 5-bit types then fit FIVE slots next to the two dequantized W tiles in the CU's
// 160 KB — four stages in flight instead of three (1.5 us of latency cover at 0.5 us per stage).  Measured: no change
// (GEMMs 10.8 vs 10.7 ms per batch) — the waits the counters show (profiles/r02_prefill_mmq_p8_pmc.txt: waves spend 47 % of
// their cycles waiting, matrix pipes busy 26 %) are not DMA landing.  Q8_0 (two quant planes) stays at four.
template <int QT>
struct Dma8 {
    static constexpr int WQ = 16384;
    static constexpr int WQ2 = WQ + 4096;                                                    // Q8_0 only
    static constexpr int WH = WQ + 4096;                                                     // Q5 only
    static constexpr int WD = WQ + 4096 + (QT == QT_Q8_0 ? 4096 : 0) + ((QT == QT_Q5_0 || QT == QT_Q5_1) ? 1024 : 0);
    static constexpr int WM = WD + 1024;                                                     // Q4_1 / Q5_1 only
    static constexpr int SLOT = WD + 1024 + ((QT == QT_Q4_1 || QT == QT_Q5_1) ? 1024 : 0);
    static constexpr int RING = (5 * SLOT + 2 * MMQ_TILEB <= 160 * 1024) ? 5 : 4;
    static constexpr int WT = RING * SLOT;
    static constexpr int LDS = WT + 2 * MMQ_TILEB;
};

template <int QT>
__device__ __forceinline__ constexpr int dma8_group() {  // DMA instructions per stage and wave: 2 of X + 1 or 2 weight parts
    return QT == QT_Q4_0 ? 3 : 4;
}

template <int QT>
__global__ void __launch_bounds__(512, 1) k_mmq_dma_p8(const MmqArgs a, int n_items, int tiles_total, int splits) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;  // 2 x 4 waves: 64 weight rows x 32 tokens each
    const int wv = wave & 3;                  // DMA: which quarter of the weight rows this wave requests
    const bool lo = wave < 4;                 // waves 0..3 request the quants, 4..7 the scales (see issue)
    constexpr int G = dma8_group<QT>();
    typedef Dma8<QT> L;
    constexpr int SLOT = L::SLOT, WT = L::WT, RING = L::RING;
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
        const int t = xcd_tile_id(b, tiles_total);
        tm = t / a.tiles_n;
        tn = t - tm * a.tiles_n;
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
        const char *xsrc[2];
        const char *wq, *wq2, *wh, *wd, *wm_;
    };
    auto lane_addr = [&](const Item &it, LaneAddr &A) {
        // X: instruction i (0..1) of wave w covers rows 16w + 8i .. +7; lane -> row +(lane>>3), physical chunk lane&7
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r = 16 * wave + 8 * i + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            A.xsrc[i] = (const char *)(a.x + min(it.n0 + r, a.N - 1) * (a.nb * 32)) + c * 16;
        }
        // W: lane -> row 32w + (lane>>1), block lane&1 of the stage; scales: lane & 31 -> row 32w + lane, both blocks in one dword
        const int64_t wrow = min(it.m0 + 32 * wv + (lane >> 1), it.M - 1);
        const int64_t wblk0 = wrow * a.nb + (lane & 1);
        const int64_t drow = min(it.m0 + 32 * wv + (lane & 31), it.M - 1) * a.nb;
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
        char *slot = lds + (gi % RING) * SLOT;
#pragma unroll
        for (int i = 0; i < 2; i++)
            __builtin_amdgcn_global_load_lds((gptr_t)(Ai.xsrc[i] + kb * 64), (lptr_t)(slot + DMA_XS + (16 * wave + 8 * i) * 128), 16, 0, 0);
        // the weight parts: 4 instructions each (32 rows per instruction); waves 0..3 take the quant planes, waves 4..7 the
        // scales.  Every wave issues the same NUMBER of DMAs per stage (the counted waits rely on it): where a type has an
        // odd number of parts the scale waves request the scales twice.
        if (lo) {
            __builtin_amdgcn_global_load_lds((gptr_t)(Ai.wq + kb * 16), (lptr_t)(slot + L::WQ + wv * 1024), 16, 0, 0);
            if constexpr (QT == QT_Q8_0)
                __builtin_amdgcn_global_load_lds((gptr_t)(Ai.wq2 + kb * 16), (lptr_t)(slot + L::WQ2 + wv * 1024), 16, 0, 0);
            if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1)
                __builtin_amdgcn_global_load_lds((gptr_t)(Ai.wh + kb * 4), (lptr_t)(slot + L::WH + wv * 256), 4, 0, 0);
            if constexpr (QT == QT_Q4_1)
                __builtin_amdgcn_global_load_lds((gptr_t)(Ai.wm_ + kb * 2), (lptr_t)(slot + L::WM + wv * 256), 4, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((gptr_t)(Ai.wd + kb * 2), (lptr_t)(slot + L::WD + wv * 256), 4, 0, 0);
            if constexpr (QT == QT_Q5_1)
                __builtin_amdgcn_global_load_lds((gptr_t)(Ai.wm_ + kb * 2), (lptr_t)(slot + L::WM + wv * 256), 4, 0, 0);
            if constexpr (QT == QT_Q8_0 || QT == QT_Q5_0 || QT == QT_Q4_1)
                __builtin_amdgcn_global_load_lds((gptr_t)(Ai.wd + kb * 2), (lptr_t)(slot + L::WD + wv * 256), 4, 0, 0);
        }
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

    // dequant assignment: half a block per thread — block tid & 255 (row wr, block wj of the stage), words 2h and 2h + 1 of
    // its four 16-byte output words, h = 0 for waves 0..3, 1 for waves 4..7 (uniform)
    const int bt = tid & 255, h = wave >> 2;
    const int wr = bt >> 1, wj = bt & 1;
    const int woff = wr * MMQ_ROWB + wj * 64 + h * 32;
    auto raw_load = [&](int s, u32x4 &q, u32x4 &q2, uint32_t &qh, _Float16 &d, _Float16 &m) {
        const char *slot = lds + (s % RING) * SLOT;
        q = *(const u32x4 *)(slot + L::WQ + bt * 16);
        q2 = q;
        qh = 0;
        m = (_Float16)0.0f;
        if constexpr (QT == QT_Q8_0) q2 = *(const u32x4 *)(slot + L::WQ2 + bt * 16);
        if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) qh = *(const uint32_t *)(slot + L::WH + bt * 4);
        d = *(const _Float16 *)(slot + L::WD + (wr >> 5) * 256 + (wr & 31) * 4 + wj * 2);
        if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1)
            m = *(const _Float16 *)(slot + L::WM + (wr >> 5) * 256 + (wr & 31) * 4 + wj * 2);
    };

    // the thread's two words moved to positions 0, 1 (and the fifth bits of those words to the low bits of each half of
    // qh), so that mmq_dequant_slice can be called with compile-time word indices 0 and 1
    auto take_half = [&](u32x4 &q, u32x4 &q2, uint32_t &qh) {
        if (h) {
            q[0] = q[2]; q[1] = q[3];
            q2[0] = q2[2]; q2[1] = q2[3];
            qh >>= 8;
        }
    };
    // ---- prologue: groups 0 .. RING-2 in flight; global stage 0's weights dequantized into W tile 0
#pragma unroll
    for (int i = 0; i < RING - 1; i++) issue();
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((RING - 2) * G) : "memory");  // group 0 landed, every wave's part
    {
        u32x4 q, q2, o[4];
        uint32_t qh;
        _Float16 d, m;
        raw_load(0, q, q2, qh, d, m);
        take_half(q, q2, qh);
        const f16x2 dd0 = {d, d}, mm0 = {m, m};
#pragma unroll
        for (int k = 0; k < 2; k++) {
#pragma unroll
            for (int t = 0; t < 4; t++) o[k][t] = mmq_dequant_slice<QT>(q, q2, qh, k, t, dd0, mm0);
            *(u32x4 *)(lds + WT + woff + k * 16) = o[k];
        }
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
      f32x16 acc[2];
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
          for (int r = 0; r < 16; r++) acc[i][r] = 0.0f;
      for (int s = 0; s < Tc.nstage; s++, g++) {
        // group g+1 landed (groups g+2 .. g+RING-2 may still be in flight), this wave's W-tile writes of the previous stage done
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((RING - 3) * G) : "memory");
        issue();  // global stage g+RING-1 into the slot of stage g-1: every wave is past its MFMAs
        const char *X = lds + (g % RING) * SLOT + DMA_XS;
        const char *W = lds + WT + (g & 1) * MMQ_TILEB;
        char *Wn = lds + WT + ((g + 1) & 1) * MMQ_TILEB;
        u32x4 q, q2;
        uint32_t qh;
        _Float16 d, m;
        raw_load(g + 1, q, q2, qh, d, m);  // stage g+1: the next item's first stage at an item boundary
        take_half(q, q2, qh);
        const f16x2 dd = {d, d}, mm = {m, m};
        // fragments are requested TWO k-steps ahead of the MFMAs that use them (all four k-steps have their own registers).
        // Measured against one step of lookahead: no difference (11.1 vs 10.8 ms per batch on different boxes, the 4-wave
        // kernel moved by the same 2 %) — the LDS round trip in front of a k-step is covered by the SIMD's other wave
        f16x8 fa[4], fb[4][2];
        auto xfrag = [&](int ks) {
            const int R = wn * 32 + frow_x;
            const int p = (ks * 2 + fh) ^ ((R >> 1) & 7);
            return *(const f16x8 *)(X + R * 128 + p * 16);
        };
        auto wfrag = [&](int i, int ks) {
            return *(const f16x8 *)(W + (wm * 64 + i * 32 + frow_x) * MMQ_ROWB + ks * 32 + fh * 16);
        };
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            fa[ks] = xfrag(ks);
#pragma unroll
            for (int i = 0; i < 2; i++) fb[ks][i] = wfrag(i, ks);
        }
        __builtin_amdgcn_sched_barrier(0);
        u32x4 o;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
#pragma unroll
            for (int t2 = 0; t2 < 2; t2++) {
                const int idx = ks * 2 + t2;  // idx 0..7: slice idx & 3 of the thread's word idx >> 2
                acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks], fb[ks][t2], acc[t2], 0, 0, 0);
                if (ks < 2) {
                    if (t2 == 0) {
                        fa[ks + 2] = xfrag(ks + 2);
                        fb[ks + 2][0] = wfrag(0, ks + 2);
                    } else {
                        fb[ks + 2][1] = wfrag(1, ks + 2);
                    }
                }
                o[idx & 3] = mmq_dequant_slice<QT>(q, q2, qh, idx >> 2, idx & 3, dd, mm);
                if ((idx & 3) == 3) *(u32x4 *)(Wn + woff + (idx >> 2) * 16) = o;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
      }
      // the item's tile is complete
#pragma unroll
      for (int i = 0; i < 2; i++) {
          const int64_t mrow = Tc.m0 + wm * 64 + i * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; r++) {
              const int64_t n = Tc.n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
              if (mrow < Tc.M && n < a.N) {
                  if (splits > 1 && a.split_stride == 0)
                      unsafeAtomicAdd(Tc.dst + n * Tc.ldd + mrow, acc[i][r]);
                  else
                      Tc.dst[n * Tc.ldd + mrow] = acc[i][r];
              }
          }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the tail DMAs before the workgroup's LDS is released
}
