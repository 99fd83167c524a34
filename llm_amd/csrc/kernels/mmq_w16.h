// mmq_w16.h — the prompt GEMM on RESIDENT f16 copies of the quantized weights.
//
// 288 GB of HBM hold a second copy of a model's 2-D weights: the f16 values w16 = f16(d * (q - zero) [+ m]) that
// k_mmq_dma_p8 dequantizes into LDS in every k-stage of every tile of every batch, computed ONCE (k_dequant_w16, at the
// first prompt batch of a model: 13.2 GB for LLaMA-7B, 130 GB for 65B next to its 69 GB of Q8_0).  The decode mat-vecs
// keep streaming the quantized blocks (they are HBM-bound: 4.5 bits per weight is what makes them fast); prompt batches
// are compute-bound, and what bounded k_mmq_dma_p8 was not the matrix pipe but the dependency chain around the
// dequantized W tile: its next stage is complete only at the barrier, so every stage began with
// barrier -> W-fragment round trip -> first MFMA, on top of ~100 VALU instructions of dequantization per wave and stage.
//
// k_mmq_w16_p8: the persistent eight-wave kernel of mmq_dmap8.h with BOTH operands arriving by LDS-DMA — the W tile is
// just a second X tile in the ring slot (same 128-byte rows, same XOR swizzle on the source address): no dequantization,
// no W double buffer, no raw-slot reads, and nothing a stage needs is produced by the stage before it.  Same tile, same
// k order, same MFMA sequence per output element as k_mmq_dma_p8: bit-identical results.
#pragma once
#include "mmq_dmap8.h"

// one thread per block: 32 weights -> 32 f16 in the GEMM's k order (mmq_kperm), 64 contiguous bytes
template <int QT>
__global__ void __launch_bounds__(256) k_dequant_w16(const QWeight w, _Float16 *__restrict__ out) {
    const int64_t blk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (blk >= w.M * w.nb) return;
    const u32x4 q = ((const u32x4 *)w.qs)[blk];
    u32x4 q2 = q;
    uint32_t qh = 0;
    _Float16 m = (_Float16)0.0f;
    if constexpr (QT == QT_Q8_0) q2 = ((const u32x4 *)w.qs2)[blk];
    if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) qh = w.qh[blk];
    const _Float16 d = ((const _Float16 *)w.d)[blk];
    if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1) m = ((const _Float16 *)w.m)[blk];
    u32x4 o[4];
    mmq_dequant<QT>(q, q2, qh, d, m, o);
    u32x4 *dst = (u32x4 *)(out + blk * 32);
#pragma unroll
    for (int k = 0; k < 4; k++) dst[k] = o[k];
}

#define W16_MIN_TOKENS 64  /* generic mul_mat: token count from which a resident weight gets its f16 copy */
#define W16_X 0
#define W16_W 16384
#define W16_SLOT 32768
#define W16_RING 5  /* 5 x 32 KB = the CU's whole 160 KB */
#define W16_LDS (W16_RING * W16_SLOT)

// MmqArgs: w.w16 / wb.w16 / wc.w16 hold the f16 copies ([M][nb * 32] f16, k permuted inside each block like the activations)
__global__ void __launch_bounds__(512, 1) k_mmq_w16_p8(const MmqArgs a, int n_items, int tiles_total, int splits) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;  // 2 x 4 waves: 64 weight rows x 32 tokens each
    constexpr int G = 4, RING = W16_RING, SLOT = W16_SLOT;
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
        int tm, tn;
        const int t = xcd_tile_id(b, tiles_total);
        tm = t / a.tiles_n;
        tn = t - tm * a.tiles_n;
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
        it.m0 = (int64_t)tm * MMQ_TM;
        it.n0 = (int64_t)tn * MMQ_TN;
        it.M = M;
        it.ldd = ldd;
        it.w16 = w16;
        it.dst = dst + (int64_t)y * a.split_stride;
        it.s_begin = y * per;
        it.nstage = min(nstage_all, it.s_begin + per) - it.s_begin;
    };
    // per-lane DMA sources: instruction i (0..1) of wave w covers rows 16w + 8i .. +7 of the X tile and of the W tile;
    // lane -> row +(lane>>3), physical 16-byte chunk lane&7 holds logical chunk (lane&7) ^ ((row >> 1) & 7)
    struct LaneAddr {
        const char *xsrc[2], *wsrc[2];
    };
    auto lane_addr = [&](const Item &it, LaneAddr &A) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r = 16 * wave + 8 * i + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            A.xsrc[i] = (const char *)(a.x + min(it.n0 + r, a.N - 1) * (a.nb * 32)) + c * 16;
            A.wsrc[i] = (const char *)(it.w16 + min(it.m0 + r, it.M - 1) * (a.nb * 32)) + c * 16;
        }
    };

    int iw = (int)blockIdx.x, is = 0, gi = 0;
    Item Ti;
    LaneAddr Ai;
    load_item(iw, Ti);
    lane_addr(Ti, Ai);
    auto issue = [&]() {
        const int64_t kb = (int64_t)(Ti.s_begin + is) * 2;  // first block of the stage: 64 bytes per block and row
        char *slot = lds + (gi % RING) * SLOT;
#pragma unroll
        for (int i = 0; i < 2; i++)
            __builtin_amdgcn_global_load_lds((gptr_t)(Ai.xsrc[i] + kb * 64), (lptr_t)(slot + W16_X + (16 * wave + 8 * i) * 128), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; i++)
            __builtin_amdgcn_global_load_lds((gptr_t)(Ai.wsrc[i] + kb * 64), (lptr_t)(slot + W16_W + (16 * wave + 8 * i) * 128), 16, 0, 0);
        gi++;
        if (is + 1 < Ti.nstage) {
            is++;
        } else if (iw + (int)gridDim.x < n_items) {
            iw += (int)gridDim.x;
            is = 0;
            load_item(iw, Ti);
            lane_addr(Ti, Ai);
        }
    };

#pragma unroll
    for (int i = 0; i < RING - 1; i++) issue();

    const int frow_x = lane & 31, fh = lane >> 5;
    auto frag = [&](const char *T, int R, int ks) {
        const int p = (ks * 2 + fh) ^ ((R >> 1) & 7);
        return *(const f16x8 *)(T + R * 128 + p * 16);
    };
    const int Rx = wn * 32 + frow_x, Rw0 = wm * 64 + frow_x, Rw1 = wm * 64 + 32 + frow_x;
    // Fragments of the first two k-steps of a stage are requested at the END of the stage before it (the wait at the
    // top of a stage retires the DMA group of the NEXT stage too), so the first MFMA after a barrier never waits for LDS.
    f16x8 fa[4], fb[4][2];
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((RING - 2) * G) : "memory");  // group 0 landed, every wave's part
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        fa[ks] = frag(lds + W16_X, Rx, ks);
        fb[ks][0] = frag(lds + W16_W, Rw0, ks);
        fb[ks][1] = frag(lds + W16_W, Rw1, ks);
    }
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
        // group g+1 landed (groups g+2 .. g+RING-2 may still be in flight); every wave is past its LDS reads of stage g-1
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((RING - 3) * G) : "memory");
        issue();  // global stage g+RING-1 into the slot of stage g-1
        const char *X = lds + (g % RING) * SLOT + W16_X, *W = lds + (g % RING) * SLOT + W16_W;
        const char *Xn = lds + ((g + 1) % RING) * SLOT + W16_X, *Wn = lds + ((g + 1) % RING) * SLOT + W16_W;
        f16x8 na[2], nb_[2][2];  // the next stage's first two k-steps
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
#pragma unroll
            for (int t2 = 0; t2 < 2; t2++) {
                acc[t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks], fb[ks][t2], acc[t2], 0, 0, 0);
                if (ks < 2) {
                    if (t2 == 0) {
                        fa[ks + 2] = frag(X, Rx, ks + 2);
                        fb[ks + 2][0] = frag(W, Rw0, ks + 2);
                    } else {
                        fb[ks + 2][1] = frag(W, Rw1, ks + 2);
                    }
                } else {
                    if (t2 == 0) {
                        na[ks - 2] = frag(Xn, Rx, ks - 2);
                        nb_[ks - 2][0] = frag(Wn, Rw0, ks - 2);
                    } else {
                        nb_[ks - 2][1] = frag(Wn, Rw1, ks - 2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            fa[ks] = na[ks];
            fb[ks][0] = nb_[ks][0];
            fb[ks][1] = nb_[ks][1];
        }
      }
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
