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
    float *dst;         // dst[n*ldd + m]
    int64_t ldd;
    int64_t M, N, nb;
    int tiles_n;
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

// LDS-DMA staging (global_load_lds_dwordx4) address-space casts, shared by the persistent GEMMs and mmq_cols.h
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
#define DMA_XS 0  // the activation tile sits at the start of a ring slot (mmq_dmap8.h)
