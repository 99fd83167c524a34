// mmvq.h — the decode hot path: quantized mat-vec  dst[n][m] = sum_k W[m][k] * x[n][k]
// for W in Q4_0 / Q4_1 / Q5_0 / Q5_1 / Q8_0 and 1..8 activation columns.
//
// Replaces ggml_compute_forward_mul_mat for quantized src0 (op GGML_OP_MUL_MAT = 21,
// crates/ggml/sys/src/lib.rs:110; builder crates/ggml/src/context.rs:314-324; call sites
// crates/models/llama/src/lib.rs:194,208,223,310,323,325,332,352) with the SAME arithmetic
// contract as the reference CPU path: the activation row is re-quantized to the weight type's
// vec_dot_type (Q8_0 or Q8_1), each 32-wide block contributes an exact integer dot product, and
// blocks are combined in f32 with the per-type formula of ggml's vec_dot (see block_dot below).
//
// MI355X shape of the kernel (HBM-bound: 18..34 B per 32 weights, read exactly once):
//  * weights live in the SoA layout of common.h → every lane issues one 16-byte non-temporal
//    global_load_dwordx4 per block, a wave covers 1 KiB of contiguous HBM per instruction;
//  * the re-quantized activation (K bytes + 8 B/block) is staged once per workgroup in LDS in a
//    planar layout, so lane l's two ds_read_b128 are at l*16 — conflict-free;
//  * integer dots run on v_dot4_i32_i8 (4 MACs/lane/instr): ≈0.7 VALU ops per weight, far below
//    the ≈5 ops/weight the VALU could spend at 8 TB/s, so the kernel stays memory-bound;
//  * one wave owns R rows × NCOLS columns; partial sums are reduced across the 64 lanes with
//    shuffles (no LDS, no atomics); grid = M/(4·R) workgroups of 256 threads ≫ 256 CUs.
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------------------------
// upload-time re-layout: raw GGML blocks (row-major [M][nb] of 18/20/22/24/34-byte structs, 2-byte
// aligned) → SoA.  One thread per block; runs once per tensor inside ggml_hip_transform_tensor.
// ---------------------------------------------------------------------------------------------
__global__ void k_relayout_q(const uint8_t *__restrict__ raw, int qt, int64_t nblocks, uint8_t *qs, uint8_t *qs2,
                             uint32_t *qh, __half *d, __half *m) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks) return;
    const int bs = qt == QT_Q4_0 ? 18 : qt == QT_Q4_1 ? 20 : qt == QT_Q5_0 ? 22 : qt == QT_Q5_1 ? 24 : 34;
    const uint16_t *src = (const uint16_t *)(raw + i * bs);  // blocks are 2-byte aligned
    int o = 0;
    ((uint16_t *)d)[i] = src[o++];
    if (qt == QT_Q4_1 || qt == QT_Q5_1) ((uint16_t *)m)[i] = src[o++];
    if (qt == QT_Q5_0 || qt == QT_Q5_1) {
        const uint32_t lo = src[o], hi = src[o + 1];
        qh[i] = lo | (hi << 16);
        o += 2;
    }
    uint16_t *q = (uint16_t *)(qs + i * 16);
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = src[o + j];
    if (qt == QT_Q8_0) {
        uint16_t *q2 = (uint16_t *)(qs2 + i * 16);
#pragma unroll
        for (int j = 0; j < 8; j++) q2[j] = src[o + 8 + j];
    }
}

// ---------------------------------------------------------------------------------------------
// activation re-quantization (ggml: quantize_row_q8_0 / quantize_row_q8_1 in the INIT phase of
// mul_mat, into cplan.work_data).  32 lanes per block, 2 blocks per wave.
//   d = amax/127 ; id = 127/amax, q = rne(x*id) (ggml's AVX2 branch, the default) or id = 1/d, q = roundf(x*id) (its scalar
//   branch): common.h act_quant ; sum = Σq
// Q8_0 kind stores d after an f16 round trip (block_q8_0.d is fp16); Q8_1 kind keeps d in f32.
// ---------------------------------------------------------------------------------------------
template <bool F16_D>
__global__ void __launch_bounds__(256) k_quantize_act(const char *__restrict__ x, int64_t nb_row /*bytes*/,
                                                      int64_t nblk /*blocks per row*/, int64_t nrows, int8_t *lo,
                                                      int8_t *hi, float *dq, int *sumq) {
    const int64_t gblock = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int l = threadIdx.x & 31;
    if (gblock >= nblk * nrows) return;  // whole 32-lane group exits together
    const int64_t row = gblock / nblk, b = gblock % nblk;
    const float v = ((const float *)(x + row * nb_row))[b * 32 + l];
    float amax = fabsf(v);
    amax = g32_max_f32(amax);
    const float d = amax / 127.0f;
    const float id = act_id(amax, d, aq_scalar());
    const int q = act_q(v * id, aq_scalar());
    int s = q;
    s = g32_sum_i32(s);
    int8_t *dst = (l < 16 ? lo : hi) + gblock * 16 + (l & 15);
    *dst = (int8_t)q;
    if (l == 0) {
        dq[gblock] = F16_D ? round_f16(d) : d;
        sumq[gblock] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// per-block arithmetic, one function per weight type.  `lo`,`hi`: the activation block's 32 int8
// (planar halves); xd: activation scale; xs: integer sum of the activation quants.
// Each returns the f32 term ggml's scalar vec_dot adds to sumf for this block:
//   q4_0·q8_0: sumi*d_w*d_x                     q5_0·q8_0: (d_w*d_x)*sumi
//   q4_1·q8_1: (d_w*d_x)*sumi + m_w*(sum*d_x)   q5_1·q8_1: same       q8_0·q8_0: sumi*(d_w*d_x)
// with sumi = Σ (w_q - zero)·x_q computed exactly in int32 on v_dot4_i32_i8.
// ---------------------------------------------------------------------------------------------
#define SDOT4(a, b, c) __builtin_amdgcn_sdot4((int)(a), (int)(b), (c), false)

// bits i..i+3 of `h` → byte i bit 4 (the position of the fifth quant bit above a nibble)
// (the compiler folds the << 4 into the constant and multiplies with v_mul_lo_u32; forcing v_mul_u32_u24 + a separate shift was
// measured: one instruction more per dword and the 13B Q5_1 w1|w3 launch 4 % slower — the 32-bit multiply is not the slow one here)
__device__ __forceinline__ uint32_t spread_hi4(uint32_t h4) { return ((h4 * 0x00204081u) & 0x01010101u) << 4; }

template <int QT>
__device__ __forceinline__ float block_dot(const u32x4 q, const u32x4 q2, const uint32_t qh, const float dw,
                                           const float mw, const i32x4 lo, const i32x4 hi, const float xd,
                                           const int xs) {
    int s = 0;
    if constexpr (QT == QT_Q4_0 || QT == QT_Q4_1) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            s = SDOT4(q[k] & 0x0F0F0F0Fu, lo[k], s);
            s = SDOT4((q[k] >> 4) & 0x0F0F0F0Fu, hi[k], s);
        }
        if constexpr (QT == QT_Q4_0) {
            s -= 8 * xs;
            return ((float)s * dw) * xd;
        } else {
            return (dw * xd) * (float)s + mw * ((float)xs * xd);
        }
    } else if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t wl = (q[k] & 0x0F0F0F0Fu) | spread_hi4((qh >> (4 * k)) & 0xFu);
            const uint32_t wh = ((q[k] >> 4) & 0x0F0F0F0Fu) | spread_hi4((qh >> (16 + 4 * k)) & 0xFu);
            s = SDOT4(wl, lo[k], s);
            s = SDOT4(wh, hi[k], s);
        }
        if constexpr (QT == QT_Q5_0) {
            s -= 16 * xs;
            return (dw * xd) * (float)s;
        } else {
            return (dw * xd) * (float)s + mw * ((float)xs * xd);
        }
    } else {  // Q8_0: signed bytes straight into the dot
#pragma unroll
        for (int k = 0; k < 4; k++) {
            s = SDOT4(q[k], lo[k], s);
            s = SDOT4(q2[k], hi[k], s);
        }
        return (float)s * (dw * xd);
    }
}

// The same arithmetic split in two for kernels that dot one weight block with several activation blocks
// (decode_big8.h): unpack the 32 codes of the block once into 8 dwords of int8 (zero point NOT applied), then
// block_dot_codes per activation column = 8 v_dot4 + the type's scale formula.
template <int QT>
__device__ __forceinline__ void block_unpack(const u32x4 q, const u32x4 q2, const uint32_t qh, uint32_t (&wl)[4],
                                             uint32_t (&wh)[4]) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if constexpr (QT == QT_Q4_0 || QT == QT_Q4_1) {
            wl[k] = q[k] & 0x0F0F0F0Fu;
            wh[k] = (q[k] >> 4) & 0x0F0F0F0Fu;
        } else if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) {
            wl[k] = (q[k] & 0x0F0F0F0Fu) | spread_hi4((qh >> (4 * k)) & 0xFu);
            wh[k] = ((q[k] >> 4) & 0x0F0F0F0Fu) | spread_hi4((qh >> (16 + 4 * k)) & 0xFu);
        } else {
            wl[k] = q[k];
            wh[k] = q2[k];
        }
    }
}
template <int QT>
__device__ __forceinline__ float block_dot_codes(const uint32_t (&wl)[4], const uint32_t (&wh)[4], const float dw,
                                                 const float mw, const i32x4 lo, const i32x4 hi, const float xd,
                                                 const int xs) {
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        s = SDOT4(wl[k], lo[k], s);
        s = SDOT4(wh[k], hi[k], s);
    }
    if constexpr (QT == QT_Q4_0) {
        s -= 8 * xs;
        return ((float)s * dw) * xd;
    } else if constexpr (QT == QT_Q4_1) {
        return (dw * xd) * (float)s + mw * ((float)xs * xd);
    } else if constexpr (QT == QT_Q5_0) {
        s -= 16 * xs;
        return (dw * xd) * (float)s;
    } else if constexpr (QT == QT_Q5_1) {
        return (dw * xd) * (float)s + mw * ((float)xs * xd);
    } else {
        return (float)s * (dw * xd);
    }
}

// ---------------------------------------------------------------------------------------------
// the mat-vec kernel.  grid.x = ceil(M / (4*R)); block = 256 threads = 4 waves; wave w owns rows
// m0..m0+R-1.  dst column stride in floats = ldd.  Dynamic LDS = NCOLS*nb*40 bytes.
// Up to three weight matrices that share the same activation (wq|wk|wv, w1|w3) can be served by one
// launch: `seg` entries partition blockIdx.x (see MmvqArgs).
// ---------------------------------------------------------------------------------------------
struct MmvqSeg {
    QWeight w;
    float *dst;
    int64_t ldd;
    int wg_begin;  // first workgroup index of this segment
};
struct MmvqArgs {
    MmvqSeg seg[3];
    int nseg;
    QAct x;
    int64_t nb;  // blocks per row (shared K)
};

template <int QT, int NCOLS, int R>
__global__ void __launch_bounds__(256) k_mmvq(const MmvqArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t nb = a.nb;
    i32x4 *s_lo = (i32x4 *)smem;
    i32x4 *s_hi = s_lo + NCOLS * nb;
    float *s_d = (float *)(s_hi + NCOLS * nb);
    int *s_sum = (int *)(s_d + NCOLS * nb);
    const int tid = threadIdx.x;

    // segment lookup (wave-uniform)
    int sg = 0;
    if (a.nseg > 1 && (int)blockIdx.x >= a.seg[1].wg_begin) sg = 1;
    if (a.nseg > 2 && (int)blockIdx.x >= a.seg[2].wg_begin) sg = 2;
    const QWeight w = a.seg[sg].w;
    const int wg = blockIdx.x - a.seg[sg].wg_begin;
    const int wave = tid >> 6, lane = tid & 63;
    const int64_t M = w.M;
    const int64_t m0 = ((int64_t)wg * 4 + wave) * R;

    // stage the quantized activation (planar) into LDS
    for (int64_t i = tid; i < NCOLS * nb; i += 256) {
        s_lo[i] = a.x.lo[i];
        s_hi[i] = a.x.hi[i];
        s_d[i] = a.x.d[i];
        s_sum[i] = a.x.sum[i];
    }
    __syncthreads();
    if (m0 >= M) return;

    float acc[R][NCOLS];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int c = 0; c < NCOLS; c++) acc[r][c] = 0.0f;

    int64_t rowoff[R];
#pragma unroll
    for (int r = 0; r < R; r++) rowoff[r] = (m0 + r < M ? m0 + r : M - 1) * nb;

    for (int64_t b = lane; b < nb; b += 64) {
        u32x4 q[R], q2[R];
        uint32_t qh[R];
        float dw[R], mw[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int64_t i = rowoff[r] + b;
            q[r] = __builtin_nontemporal_load((const u32x4 *)(w.qs + i * 16));
            if constexpr (QT == QT_Q8_0)
                q2[r] = __builtin_nontemporal_load((const u32x4 *)(w.qs2 + i * 16));
            else
                q2[r] = q[r];
            if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1)
                qh[r] = __builtin_nontemporal_load(w.qh + i);
            else
                qh[r] = 0;
            dw[r] = __half2float(w.d[i]);
            if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1)
                mw[r] = __half2float(w.m[i]);
            else
                mw[r] = 0.0f;
        }
#pragma unroll
        for (int c = 0; c < NCOLS; c++) {
            const i32x4 lo = s_lo[c * nb + b];
            const i32x4 hi = s_hi[c * nb + b];
            const float xd = s_d[c * nb + b];
            const int xs = s_sum[c * nb + b];
#pragma unroll
            for (int r = 0; r < R; r++) acc[r][c] += block_dot<QT>(q[r], q2[r], qh[r], dw[r], mw[r], lo, hi, xd, xs);
        }
    }
    float *dst = a.seg[sg].dst;
    const int64_t ldd = a.seg[sg].ldd;
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int c = 0; c < NCOLS; c++) {
            const float v = wave_sum_f32(acc[r][c]);
            if (lane == 0 && m0 + r < M) dst[c * ldd + m0 + r] = v;
        }
}

// dequantize rows of a SoA weight to f32: get_rows (crates/models/llama/src/lib.rs:170; builder
// crates/ggml/src/context.rs:283-287).  grid = (ceil(nb/256), N); one thread per block.
__global__ void k_get_rows_q(const QWeight w, const int *__restrict__ ids, float *dst, int64_t ldd) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= w.nb) return;
    const int64_t row = ids[blockIdx.y];
    const int64_t i = row * w.nb + b;
    float *y = dst + (int64_t)blockIdx.y * ldd + b * 32;
    const float d = __half2float(w.d[i]);
    const uint8_t *qs = w.qs + i * 16;
    if (w.qt == QT_Q8_0) {
        const int8_t *q1 = (const int8_t *)qs, *q2 = (const int8_t *)(w.qs2 + i * 16);
        for (int j = 0; j < 16; j++) {
            y[j] = q1[j] * d;
            y[j + 16] = q2[j] * d;
        }
        return;
    }
    const bool has_m = w.qt == QT_Q4_1 || w.qt == QT_Q5_1;
    const bool has_h = w.qt == QT_Q5_0 || w.qt == QT_Q5_1;
    const float m = has_m ? __half2float(w.m[i]) : 0.0f;
    const uint32_t qh = has_h ? w.qh[i] : 0u;
    const int zero = w.qt == QT_Q4_0 ? 8 : w.qt == QT_Q5_0 ? 16 : 0;
    for (int j = 0; j < 16; j++) {
        const int xh0 = has_h ? (int)(((qh >> j) << 4) & 0x10u) : 0;
        const int xh1 = has_h ? (int)((qh >> (j + 12)) & 0x10u) : 0;
        const int x0 = ((qs[j] & 0x0F) | xh0) - zero;
        const int x1 = ((qs[j] >> 4) | xh1) - zero;
        // ggml: y = x*d (+ m) — separate multiply and add (no fma: built with -ffp-contract=off)
        y[j] = has_m ? x0 * d + m : x0 * d;
        y[j + 16] = has_m ? x1 * d + m : x1 * d;
    }
}
