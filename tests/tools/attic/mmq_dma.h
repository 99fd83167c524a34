// mmq_dma.h — the prompt GEMM with LDS-DMA staging (global_load_lds_dwordx4), same arithmetic and tile as k_mmq.
//
// Why a second kernel: k_mmq stages both operands through registers, and hipcc's s_waitcnt insertion drains the
// whole global-load queue at every k-stage whatever the depth of the register ring (it waits for vmcnt <= 5 right
// after issuing the stage's 6 loads), so the effective prefetch distance is one stage and the kernel is
// latency-bound: 43 % of its wave cycles parked in s_waitcnt/barriers, 18 % of the f16 MFMA peak
// (profiles/r01_run23_prefill_mmq_pmc.txt).  LDS-DMA writes have no register result the compiler could attach a
// wait to, so here EVERY global operand byte — the f16 activations, the raw quant nibbles, the block scales —
// goes global -> LDS by DMA into a 4-slot ring, three k-stages ahead, and the only waits are the hand-placed
// `s_waitcnt vmcnt(G)` (G = DMA instructions per stage and wave, the same for every wave) in front of a raw
// s_barrier.  One workgroup per CU (140 KB of LDS), MFMAs interleaved slice by slice with the dequantization of
// the next stage's weights (raw slot -> VALU -> padded f16 tile) exactly as in k_mmq.
//
//   ring slot (27,648 B): X 128 tokens x 128 B, chunk-XOR-swizzled (the DMA writes lane-linear, so the swizzle is
//                         applied to each lane's SOURCE address: physical chunk p of row r holds logical chunk
//                         p ^ ((r >> 1) & 7); fragment reads of 32 consecutive rows are then conflict-free)
//                         Wq [128 rows][2 blocks][16 B] (+ Wq2 for Q8_0), Wh [row][blk] u32 (Q5),
//                         Wd / Wm [row][2] f16 (one aligned 4-byte DMA per row: needs K/32 even)
//   W tile x 2 (18,432 B each): dequantized f16, 144-byte rows as in k_mmq
#pragma once
#include "mmq.h"

// X8 variant (template parameter): the activations stay int8 + f16 block scale in global memory (1.06 B per
// element instead of 2: a stage ingests 13.3 KB instead of 20.5 — the kernel is bound by what a CU can ingest) and are
// dequantized by the same threads and the same code path as Q8_0 weights into a second padded f16 tile.
#define DMA_XS 0
#define DMA_WQ 16384
#define DMA_WQ2 20480
#define DMA_WH 24576
#define DMA_WD 25600   /* 4 waves x 256 B: lanes 0..31 = the wave's rows, lanes 32..63 repeat them (lane-linear DMA) */
#define DMA_WM 26624
#define DMA_SLOT 27648
#define DMA_RING 4
#define DMA_WT (DMA_RING * DMA_SLOT)
#define DMA_LDS (DMA_WT + 2 * MMQ_TILEB)
// X8 ring slot: X8 [128 rows][64 B] at 0, DX (4 waves x 256 B) at 8192, then the weight parts at the same relative
// offsets as above minus 7168
#define D8_X8 0
#define D8_DX 8192
#define D8_SHIFT 7168 /* DMA_WQ - 9216: weight parts start at 9216 */
#define D8_SLOT (DMA_SLOT - D8_SHIFT)
#define D8_WT (DMA_RING * D8_SLOT)
#define D8_XT (D8_WT + 2 * MMQ_TILEB)
#define D8_LDS (D8_XT + 2 * MMQ_TILEB)

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int QT, bool X8 = false>
__device__ __forceinline__ constexpr int dma_group() {  // DMA instructions per stage and wave
    return (X8 ? 3 : 4) + 1 + (QT == QT_Q8_0 ? 1 : 0) + ((QT == QT_Q5_0 || QT == QT_Q5_1) ? 1 : 0) + 1 +
           ((QT == QT_Q4_1 || QT == QT_Q5_1) ? 1 : 0);
}

template <int QT, bool X8 = false>
__global__ void __launch_bounds__(256, 1) k_mmq_dma(const MmqArgs a_in) {
    MmqArgs a = a_in;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    constexpr int G = dma_group<QT, X8>();
    constexpr int WSH = X8 ? D8_SHIFT : 0, SLOT = X8 ? D8_SLOT : DMA_SLOT, WT = X8 ? D8_WT : DMA_WT;

    int tm, tn;
    if (a.xcd_by_n) {  // token-tile-major walk: the whole chip works on ONE 128-token slice of X at a time
        const int tiles_m = (int)gridDim.x / a.tiles_n;
        tn = (int)blockIdx.x / tiles_m;
        tm = (int)blockIdx.x % tiles_m;
    } else {
        const int t = xcd_tile_id(blockIdx.x, gridDim.x);
        tm = t / a.tiles_n;
        tn = t % a.tiles_n;
    }
    mmq_select_seg(a, tm);
    const int64_t m0 = (int64_t)tm * MMQ_TM, n0 = (int64_t)tn * MMQ_TN;

    // this workgroup's stages [s_begin, s_end) of the K loop (K/32 is even: checked by the launcher)
    const int nstage_all = (int)(a.nb >> 1);
    const int per = (nstage_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int s_begin = (int)blockIdx.y * per, s_end = min(nstage_all, s_begin + per);
    const int nstage = s_end - s_begin;
    if (nstage <= 0) return;  // uniform

    // ---- per-lane DMA source addresses (bytes), advanced by a fixed stride per stage
    // X: instruction i (0..3) of wave w covers rows 32w + 8i .. +7; lane -> row +(lane>>3), physical chunk lane&7
    const char *xsrc[4];
    const char *dxsrc = nullptr;
    if constexpr (X8) {
        // int8 rows: instruction i (0..1) of wave w covers rows 32w + 16i .. +15; lane -> row +(lane>>2), 16-B chunk lane&3
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r = 32 * wave + 16 * i + (lane >> 2);
            xsrc[i] = (const char *)a.x8 + min(n0 + r, a.N - 1) * (a.nb * 32) + (lane & 3) * 16;
        }
        xsrc[2] = xsrc[3] = xsrc[0];
        dxsrc = (const char *)a.dx + min(n0 + 32 * wave + (lane & 31), a.N - 1) * (a.nb * 2);
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 32 * wave + 8 * i + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            xsrc[i] = (const char *)(a.x + min(n0 + r, a.N - 1) * (a.nb * 32)) + c * 16;
        }
    }
    // W: lane -> row 32w + (lane>>1), block lane&1 of the stage
    const int64_t wrow = min(m0 + 32 * wave + (lane >> 1), a.M - 1);
    const int64_t wblk0 = wrow * a.nb + (lane & 1);
    // scales: lane < 32 -> row 32w + lane, both blocks of the stage in one dword
    const int64_t drow = min(m0 + 32 * wave + (lane & 31), a.M - 1) * a.nb;

    auto issue = [&](int s /* local stage, clamped */) {
        const int sc = min(s, nstage - 1);
        const int64_t kb = (int64_t)(s_begin + sc) * 2;  // first block of the stage
        char *slot = lds + (s & (DMA_RING - 1)) * SLOT;
        if constexpr (X8) {
#pragma unroll
            for (int i = 0; i < 2; i++)
                __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[i] + kb * 32), (lptr_t)(slot + D8_X8 + (32 * wave + 16 * i) * 64),
                                                 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(dxsrc + kb * 2), (lptr_t)(slot + D8_DX + wave * 256), 4, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
                __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[i] + kb * 64),
                                                 (lptr_t)(slot + DMA_XS + (32 * wave + 8 * i) * 128), 16, 0, 0);
        }
        slot -= WSH;  // the weight parts of an X8 slot sit WSH bytes lower than in the f16 layout
        __builtin_amdgcn_global_load_lds((gptr_t)(a.w.qs + (wblk0 + kb) * 16), (lptr_t)(slot + DMA_WQ + wave * 1024), 16, 0, 0);
        if constexpr (QT == QT_Q8_0)
            __builtin_amdgcn_global_load_lds((gptr_t)(a.w.qs2 + (wblk0 + kb) * 16), (lptr_t)(slot + DMA_WQ2 + wave * 1024), 16,
                                             0, 0);
        if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1)
            __builtin_amdgcn_global_load_lds((gptr_t)(a.w.qh + wblk0 + kb), (lptr_t)(slot + DMA_WH + wave * 256), 4, 0, 0);
        // one dword = the two scales of a row's stage; lanes 32..63 repeat rows 0..31 into the upper half of the
        // wave's own 256-byte strip (the destination is lane-linear, all 64 lanes write)
        __builtin_amdgcn_global_load_lds((gptr_t)((const char *)a.w.d + (drow + kb) * 2), (lptr_t)(slot + DMA_WD + wave * 256), 4,
                                         0, 0);
        if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1)
            __builtin_amdgcn_global_load_lds((gptr_t)((const char *)a.w.m + (drow + kb) * 2),
                                             (lptr_t)(slot + DMA_WM + wave * 256), 4, 0, 0);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[j][i][r] = 0.0f;

    // dequant assignment: one block per thread (row wr, block wj of the stage)
    const int wr = tid >> 1, wj = tid & 1;
    const int woff = wr * MMQ_ROWB + wj * 64;
    auto raw_load = [&](int s, u32x4 &q, u32x4 &q2, uint32_t &qh, _Float16 &d, _Float16 &m) {
        const char *slot = lds + (s & (DMA_RING - 1)) * SLOT - WSH;
        q = *(const u32x4 *)(slot + DMA_WQ + tid * 16);
        q2 = q;
        qh = 0;
        m = (_Float16)0.0f;
        if constexpr (QT == QT_Q8_0) q2 = *(const u32x4 *)(slot + DMA_WQ2 + tid * 16);
        if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) qh = *(const uint32_t *)(slot + DMA_WH + tid * 4);
        // scales of row wr live in wave (wr>>5)'s 256-byte strip: [row & 31][2] f16
        d = *(const _Float16 *)(slot + DMA_WD + (wr >> 5) * 256 + (wr & 31) * 4 + wj * 2);
        if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1)
            m = *(const _Float16 *)(slot + DMA_WM + (wr >> 5) * 256 + (wr & 31) * 4 + wj * 2);
    };

    // X8: the thread's activation block (token row wr, block wj of the stage): 32 int8 + its f16 scale
    auto xraw_load = [&](int s, u32x4 &xq, u32x4 &xq2, _Float16 &xd) {
        const char *slot = lds + (s & (DMA_RING - 1)) * SLOT;
        xq = *(const u32x4 *)(slot + D8_X8 + wr * 64 + wj * 32);
        xq2 = *(const u32x4 *)(slot + D8_X8 + wr * 64 + wj * 32 + 16);
        xd = *(const _Float16 *)(slot + D8_DX + (wr >> 5) * 256 + (wr & 31) * 4 + wj * 2);
    };

    // ---- prologue: groups 0, 1, 2 in flight; stage 0's weights dequantized into W tile 0
    issue(0);
    issue(1);
    issue(2);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * G) : "memory");  // group 0 landed, every wave's part
    {
        u32x4 q, q2, o[4];
        uint32_t qh;
        _Float16 d, m;
        raw_load(0, q, q2, qh, d, m);
        mmq_dequant<QT>(q, q2, qh, d, m, o);
#pragma unroll
        for (int k = 0; k < 4; k++) *(u32x4 *)(lds + WT + woff + k * 16) = o[k];
        if constexpr (X8) {
            u32x4 xq, xq2;
            _Float16 xd;
            xraw_load(0, xq, xq2, xd);
            mmq_dequant<QT_Q8_0>(xq, xq2, 0u, xd, (_Float16)0.0f, o);
#pragma unroll
            for (int k = 0; k < 4; k++) *(u32x4 *)(lds + D8_XT + woff + k * 16) = o[k];
        }
    }

    const int frow_x = lane & 31, fh = lane >> 5;
    for (int s = 0; s < nstage; s++) {
        // group s+1 landed (groups s+2 may still be in flight), this wave's W-tile writes of the previous stage done
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(G) : "memory");
        issue(s + 3);  // slot of stage s-1: every wave is past its MFMAs
        const char *X = X8 ? lds + D8_XT + (s & 1) * MMQ_TILEB : lds + (s & (DMA_RING - 1)) * SLOT + DMA_XS;
        const char *W = lds + WT + (s & 1) * MMQ_TILEB;
        char *Wn = lds + WT + ((s + 1) & 1) * MMQ_TILEB;
        char *Xn = lds + D8_XT + ((s + 1) & 1) * MMQ_TILEB;
        u32x4 q, q2, xq = {0, 0, 0, 0}, xq2 = {0, 0, 0, 0};
        uint32_t qh;
        _Float16 d, m, xd = (_Float16)0.0f;
        raw_load(s + 1, q, q2, qh, d, m);  // stage s+1 (clamped duplicates at the end are dequantized and never read)
        if constexpr (X8) xraw_load(s + 1, xq, xq2, xd);
        const f16x2 dd = {d, d}, mm = {m, m}, xdd = {xd, xd}, zz = {(_Float16)0.0f, (_Float16)0.0f};
        f16x8 fa[2][2], fb[2][2];
        auto xfrag = [&](int j, int ks) {
            const int R = wn * 64 + j * 32 + frow_x;
            if constexpr (X8) {
                return *(const f16x8 *)(X + R * MMQ_ROWB + ks * 32 + fh * 16);
            } else {
                const int p = (ks * 2 + fh) ^ ((R >> 1) & 7);
                return *(const f16x8 *)(X + R * 128 + p * 16);
            }
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
            u32x4 o, ox;
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
                if constexpr (X8) ox[t2] = mmq_dequant_slice<QT_Q8_0>(xq, xq2, 0u, ks, t2, xdd, zz);
                if (t2 == 3) {
                    *(u32x4 *)(Wn + woff + ks * 16) = o;
                    if constexpr (X8) *(u32x4 *)(Xn + woff + ks * 16) = ox;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the clamped tail DMAs before the workgroup's LDS is released

    // K split in two: either f32 atomic adds into a zeroed dst (two addends commute), or — split_stride != 0 — each
    // half stores its partial tile to its own buffer (dst + blockIdx.y * split_stride) and the consumer adds them
    const bool split = gridDim.y > 1 && a.split_stride == 0;
    float *const dstp = a.dst + (int64_t)blockIdx.y * a.split_stride;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int64_t mrow = m0 + wm * 64 + i * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t n = n0 + wn * 64 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (mrow < a.M && n < a.N) {
                    if (split)
                        unsafeAtomicAdd(a.dst + n * a.ldd + mrow, acc[j][i][r]);
                    else
                        dstp[n * a.ldd + mrow] = acc[j][i][r];
                }
            }
        }
}
