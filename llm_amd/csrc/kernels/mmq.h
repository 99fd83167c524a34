// mmq.h — the prefill path: quantized GEMM  dst[n][m] = sum_k W[m][k] * x[n][k]  for N >= 32 tokens on the
// f16 matrix cores (v_mfma_f32_32x32x16_f16), W in Q4_0 / Q4_1 / Q5_0 / Q5_1 / Q8_0.
//
// Replaces ggml_compute_forward_mul_mat for quantized src0 when src1 has many rows (the prompt batch:
// crates/llm-base/src/inference_session.rs:315-316 feeds `n_batch` tokens per evaluate; same call sites as
// mmvq.h).  Arithmetic contract: as in the reference CPU path the activation row is first re-quantized to the
// weight type's vec_dot_type (Q8_0 / Q8_1: d = amax/127, q = round(x/d)); the GEMM then runs on
//     w16 = f16(d_w * (q_w - zero) [+ m_w])         x16 = f16(d_x * q_x)
// with f32 accumulation in the MFMA.  Against the reference's exact integer block dots this adds two f16
// roundings (2^-12 relative, unbiased) per product — two orders below the Q8 activation quantization noise the
// reference itself carries, which is reproduced exactly.  The parity tests state the resulting tolerance.
//
// MI355X shape (compute-bound: 2*M*N*K flop over M*K*0.56 B of weights, AI ~ 1800 flop/B at N=512):
//  * workgroup = 256 threads = 4 waves (2 x 2), tile 128 weight rows x 128 tokens, K advanced 64 at a time;
//    each wave owns 64 x 64 outputs = 2 x 2 MFMA tiles of 32x32 (64 accumulator VGPRs);
//  * weights come from the SoA planes of common.h: one thread dequantizes one 32-wide block per stage
//    (16 B of nibbles -> 32 f16 with the 0x6400 exponent trick + v_pk_*_f16) into LDS, so the dequant is done
//    once per workgroup and amortised over the 128 token columns;
//  * the k index inside a block is PERMUTED (mmq_kperm) to the order the nibble trick produces; the
//    activation pre-pass writes x16 in the same order, so no cross-lane shuffles or byte permutes are needed
//    (a dot product does not care about the order of k as long as both operands agree);
//  * LDS rows are 64 f16 + 16 B pad = 144 B: the 16-byte fragment reads of 32 consecutive rows then fall in
//    distinct banks; two stages are double-buffered (73.7 KB per workgroup -> 2 workgroups per CU);
//  * blockIdx -> tile mapping is XCD-aware: the token tiles that share one 128-row weight slab run on the
//    same XCD back to back, so the slab is fetched from HBM once and re-read from that XCD's L2.
#pragma once
#include "common.h"

#define MMQ_TM 128        // weight rows per workgroup
#define MMQ_TN 128        // tokens per workgroup
#define MMQ_BK 64         // k per stage (2 blocks)
#define MMQ_ROWB 144      // LDS bytes per tile row
#define MMQ_TILEB (128 * MMQ_ROWB)
#define MMQ_LDS (4 * MMQ_TILEB)  // {W, X} x 2 stages

// position p (0..31) inside a block of the permuted k order -> element index of the GGML block
__host__ __device__ __forceinline__ int mmq_kperm(int p) {
    const int k = p >> 3, j = p & 7, jj = j & 3;
    return 4 * k + (((jj & 1) << 1) | (jj >> 1)) + ((j & 4) ? 16 : 0);
}
// element index e (0..31) -> position in the permuted order
__host__ __device__ __forceinline__ int mmq_kperm_inv(int e) {
    const int h = e >> 4, k = (e & 15) >> 2, i = e & 3;
    return 8 * k + 4 * h + (((i & 1) << 1) | (i >> 1));
}

// ---------------------------------------------------------------------------------------------
// activation pre-pass: f32 row -> Q8_0/Q8_1 quantization (as k_quantize_act) -> f16(d*q), permuted.
// 32 lanes per block.  Output x16: [nrows][nblk*32] f16.
// ---------------------------------------------------------------------------------------------
template <bool F16_D>
__global__ void __launch_bounds__(256) k_quant_act_f16(const char *__restrict__ x, int64_t nb_row /*bytes*/,
                                                       int64_t nblk, int64_t nrows, _Float16 *__restrict__ out) {
    const int64_t gblock = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int l = threadIdx.x & 31;
    if (gblock >= nblk * nrows) return;
    const int64_t row = gblock / nblk, b = gblock % nblk;
    const float v = ((const float *)(x + row * nb_row))[b * 32 + l];
    float amax = fabsf(v);
    amax = g32_max_f32(amax);
    float d = amax / 127.0f;
    const float id = act_id(amax, d, aq_scalar());
    const int q = act_q(v * id, aq_scalar());
    if (F16_D) d = round_f16(d);
    float r = d * (float)q;
    r = fminf(fmaxf(r, -65504.0f), 65504.0f);
    out[gblock * 32 + mmq_kperm_inv(l)] = (_Float16)r;
}

// X8 pre-pass: the same quantization, kept as int8 (block order as in a Q8_0 block) + one f16 scale per block.
// The GEMM then forms f16(f16(d) * q): identical to the f16 pre-pass for the Q8_0 kind (d is an f16 value there),
// one extra f16 rounding of d (2^-12) for the Q8_1 kind.
template <bool F16_D>
__global__ void __launch_bounds__(256) k_quant_act_q8p(const char *__restrict__ x, int64_t nb_row /*bytes*/, int64_t nblk,
                                                       int64_t nrows, int8_t *__restrict__ q8, _Float16 *__restrict__ dx) {
    const int64_t gblock = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int l = threadIdx.x & 31;
    if (gblock >= nblk * nrows) return;
    const int64_t row = gblock / nblk, b = gblock % nblk;
    const float v = ((const float *)(x + row * nb_row))[b * 32 + l];
    float amax = fabsf(v);
    amax = g32_max_f32(amax);
    float d = amax / 127.0f;
    const float id = act_id(amax, d, aq_scalar());
    const int q = act_q(v * id, aq_scalar());
    q8[gblock * 32 + l] = (int8_t)q;
    if (l == 0) dx[gblock] = (_Float16)fminf(F16_D ? round_f16(d) : d, 65504.0f / 127.0f);
}

// ---------------------------------------------------------------------------------------------
// block dequantization: 32 weights -> 4 x (8 f16) in the permuted order
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f16x2 as_h2(uint32_t u) { return __builtin_bit_cast(f16x2, u); }
__device__ __forceinline__ uint32_t as_u32(f16x2 h) { return __builtin_bit_cast(uint32_t, h); }

template <int QT>
__device__ __forceinline__ uint32_t mmq_cvt(uint32_t magic, f16x2 dd, f16x2 mm) {
    // `magic` holds two f16 of value 1024 + u (u = unsigned code in the low mantissa bits)
    f16x2 t = as_h2(magic);
    if constexpr (QT == QT_Q4_0) {
        t = (t - (_Float16)1032.0f) * dd;
    } else if constexpr (QT == QT_Q5_0) {
        t = (t - (_Float16)1040.0f) * dd;
    } else if constexpr (QT == QT_Q8_0) {
        t = (t - (_Float16)1152.0f) * dd;
    } else {  // Q4_1 / Q5_1: d*q + m with a single rounding
        t = t - (_Float16)1024.0f;
        f16x2 r;
        r.x = __builtin_fmaf16(t.x, dd.x, mm.x);
        r.y = __builtin_fmaf16(t.y, dd.y, mm.y);
        t = r;
    }
    return as_u32(t);
}

template <int QT>
__device__ __forceinline__ void mmq_dequant(const u32x4 q, const u32x4 q2, const uint32_t qh, const _Float16 d,
                                            const _Float16 m, u32x4 out[4]) {
    const f16x2 dd = {d, d}, mm = {m, m};
    const uint32_t MAGIC = 0x64006400u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t lo, hi;  // bytes = codes of elements 4k..4k+3 and 16+4k..16+4k+3
        if constexpr (QT == QT_Q4_0 || QT == QT_Q4_1) {
            lo = q[k] & 0x0F0F0F0Fu;
            hi = (q[k] >> 4) & 0x0F0F0F0Fu;
        } else if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) {
            lo = (q[k] & 0x0F0F0F0Fu) | ((((qh >> (4 * k)) & 0xFu) * 0x00204081u & 0x01010101u) << 4);
            hi = ((q[k] >> 4) & 0x0F0F0F0Fu) | ((((qh >> (16 + 4 * k)) & 0xFu) * 0x00204081u & 0x01010101u) << 4);
        } else {
            lo = q[k] ^ 0x80808080u;
            hi = q2[k] ^ 0x80808080u;
        }
        u32x4 o;
        o[0] = mmq_cvt<QT>((lo & 0x00FF00FFu) | MAGIC, dd, mm);         // (e4k,   e4k+2)
        o[1] = mmq_cvt<QT>(((lo >> 8) & 0x00FF00FFu) | MAGIC, dd, mm);  // (e4k+1, e4k+3)
        o[2] = mmq_cvt<QT>((hi & 0x00FF00FFu) | MAGIC, dd, mm);         // (e16+4k,   e16+4k+2)
        o[3] = mmq_cvt<QT>(((hi >> 8) & 0x00FF00FFu) | MAGIC, dd, mm);  // (e16+4k+1, e16+4k+3)
        out[k] = o;
    }
}

// one 32-bit output (two f16) of the block dequantization: word k (0..3), slot t (0..3) — see mmq_dequant
template <int QT>
__device__ __forceinline__ uint32_t mmq_dequant_slice(const u32x4 q, const u32x4 q2, const uint32_t qh, int k, int t,
                                                      f16x2 dd, f16x2 mm) {
    uint32_t src;  // bytes = codes of elements 4k..4k+3 (t < 2) or 16+4k..16+4k+3 (t >= 2)
    if constexpr (QT == QT_Q4_0 || QT == QT_Q4_1) {
        src = (t < 2 ? q[k] : (q[k] >> 4)) & 0x0F0F0F0Fu;
    } else if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) {
        const uint32_t h4 = (qh >> ((t < 2 ? 0 : 16) + 4 * k)) & 0xFu;
        src = ((t < 2 ? q[k] : (q[k] >> 4)) & 0x0F0F0F0Fu) | (((h4 * 0x00204081u) & 0x01010101u) << 4);
    } else {
        src = (t < 2 ? q[k] : q2[k]) ^ 0x80808080u;
    }
    const uint32_t magic = (((t & 1) ? (src >> 8) : src) & 0x00FF00FFu) | 0x64006400u;
    return mmq_cvt<QT>(magic, dd, mm);
}

// One k-stage (64 k) of the 128 x 128 workgroup tile on the matrix cores, shared by the quantized GEMM and the
// f16 batched GEMM (gemm_f16.h).  Wt / Xt: LDS tiles [128 rows][MMQ_ROWB bytes] of f16, k contiguous.
// acc[j][i]: rows = Xt rows wn*64 + j*32.. (MFMA A operand), columns = Wt rows wm*64 + i*32.. (B operand).
__device__ __forceinline__ void mma_stage_128x128(const char *Wt, const char *Xt, int lane, int wm, int wn,
                                                  f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        f16x8 fa[2], fb[2];
        const int koff = ks * 32 + (lane >> 5) * 16;
#pragma unroll
        for (int j = 0; j < 2; j++) fa[j] = *(const f16x8 *)(Xt + (wn * 64 + j * 32 + (lane & 31)) * MMQ_ROWB + koff);
#pragma unroll
        for (int i = 0; i < 2; i++) fb[i] = *(const f16x8 *)(Wt + (wm * 64 + i * 32 + (lane & 31)) * MMQ_ROWB + koff);
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[j], fb[i], acc[j][i], 0, 0, 0);
    }
}

// XCD-aware tile id: workgroups b, b+8, b+16, ... (one XCD) walk consecutive tiles
__device__ __forceinline__ int xcd_tile_id(int b, int nwg) {
    const int xq = nwg >> 3, xr = nwg & 7, xcd = b & 7;
    return xcd * xq + (xcd < xr ? xcd : xr) + (b >> 3);
}

struct MmqArgs {
    QWeight w;
    const _Float16 *x;  // [N][nb*32] permuted f16 activations
    const int8_t *x8;   // X8 variant of k_mmq_dma: [N][nb*32] int8 quants (ggml order inside a block)
    const _Float16 *dx; //                           [N][nb] f16 block scales
    float *dst;         // dst[n*ldd + m]
    int64_t ldd;
    int64_t M, N, nb;
    int tiles_n;
    int xcd_by_n;  // 1: XCD x works on token tile x % tiles_n only (its slice of the activations stays in that L2)
    // Several weight matrices that share the activations in ONE launch (the prompt plan's wq|wk|wv and w1|w3: the tiles
    // of all matrices fill the chip together instead of each launch ending on a partial round of workgroups — 344
    // tiles of a 7B w1 on 256 CUs are two rounds, 688 tiles of w1|w3 are three, not four).  Tile rows [0, tile_end[0])
    // belong to w / dst, [tile_end[0], tile_end[1]) to wb / dst_b, the rest to wc / dst_c.  nseg <= 1: single matrix.
    int64_t split_stride;  // != 0 with a K split: partial tiles are stored at dst + split * split_stride instead of added atomically
    int nseg;
    int tile_end[2];
    QWeight wb, wc;
    float *dst_b, *dst_c;
    int64_t ldd_b, ldd_c;
};

// multi-matrix launch: tile row -> (matrix, tile row inside it); everything here is workgroup-uniform (scalar registers)
__device__ __forceinline__ void mmq_select_seg(MmqArgs &a, int &tm) {
    if (a.nseg > 1) {
        if (a.nseg > 2 && tm >= a.tile_end[1]) {
            tm -= a.tile_end[1];
            a.w = a.wc;
            a.dst = a.dst_c;
            a.ldd = a.ldd_c;
            a.M = a.wc.M;
        } else if (tm >= a.tile_end[0]) {
            tm -= a.tile_end[0];
            a.w = a.wb;
            a.dst = a.dst_b;
            a.ldd = a.ldd_b;
            a.M = a.wb.M;
        }
    }
}

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
template <int QT>
__device__ __forceinline__ void mmq_load_w(MmqW<QT> &s, const MmqArgs &a, int64_t wrow, int64_t kb /*first block*/, int wj) {
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
__device__ __forceinline__ void mmq_load_x(MmqX &s, const MmqArgs &a, int64_t kb, const _Float16 *xrow[4], int xc) {
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
__global__ void __launch_bounds__(256, 2) k_mmq(const MmqArgs a_in) {
    MmqArgs a = a_in;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;

    // workgroup -> tile.  The activation operand is the bigger stream (a workgroup reads 128 tokens x K f16 = 1 MB
    // at K = 4096 against 0.3 MB of weights), so each XCD is pinned to ONE token tile: its 1 MB slice stays in that
    // XCD's 4 MB L2 and is re-read from there by every weight slab; the weights stream through once per XCD.
    int tm, tn;
    if (a.xcd_by_n) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per = 8 / a.tiles_n;
        tn = xcd % a.tiles_n;
        tm = idx * per + xcd / a.tiles_n;
    } else {
        const int t = xcd_tile_id(blockIdx.x, gridDim.x);
        tm = t / a.tiles_n;
        tn = t % a.tiles_n;
    }
    mmq_select_seg(a, tm);
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
