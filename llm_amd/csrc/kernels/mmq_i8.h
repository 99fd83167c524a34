// mmq_i8.h — the prompt GEMM on the INTEGER matrix cores: ggml's block dots, bit for bit, at matrix-core rate.
//
// ggml_compute_forward_mul_mat for quantized src0 (crates/models/llama/src/lib.rs:194-352; op builder
// crates/ggml/src/context.rs:314-324) computes, per 32-wide block, an exact integer dot of the weight codes with the
// int8-requantized activations and scales it in f32.  v_mfma_i32_32x32x32_i8 has K = 32: ONE instruction = the 32 x 32
// integer block dots of 32 weight rows x 32 tokens for one block column, exact in i32.  So, unlike the f16 GEMMs
// (which round d*q of both operands to f16 and accumulate in the f16 pipe: 1.1e-3*scale), this kernel's only
// difference from ggml's CPU result is the f32 association of the per-block terms — the mat-vec kernels' bound.
//
//   tile     128 tokens x 128 weight rows per 256-thread workgroup, 4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles
//   k stage  2 blocks (64 weights); every operand byte goes HBM -> LDS by DMA into a ring of I8_RING slots, RING - 1
//            stages ahead (hand-placed counted waits, one raw s_barrier per stage, as in mmq_dmap8.h)
//   A (X)    int8 activations [token][32 B per block], 16-byte chunks XOR-swizzled on the DMA source address so that the
//            fragment reads (32 token rows, 64 B apart) are conflict-free; lanes 0-31 take elements 0-15 of the block,
//            lanes 32-63 elements 16-31 (any k order works as long as both operands use the same one)
//   B (W)    the raw 16 nibble bytes of a block: lanes 0-31 use the low nibbles (elements 0-15), lanes 32-63 the high
//            nibbles (16-31): one shift + mask per dword; Q5: the fifth bits from qh; Q8_0: the two 16-byte planes
//   scaling  C tile element (token n, row m) of block b:  acc += (sumi - z * sum_n,b) * d_x[n,b] * d_w[m,b]
//            (+ m_w[m,b] * s_x[n,b] for Q4_1 / Q5_1).  The zero point rides in the MFMA's C operand (-z*sum of the lane's
//            16 tokens, straight from LDS); the rest is f32 on the VALU: a lane owns ONE weight row (its d_w, m_w) and
//            16 tokens whose d_x come from LDS as four float4.  cvt + mul + fma per output and block: the kernel is
//            bound by this scaling stream, not by the matrix pipe.
#pragma once
#include "mmq.h"

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void *gptr8_t;
typedef __attribute__((address_space(3))) void *lptr8_t;

#define I8_X 0        /* 128 tokens x 64 B */
#define I8_DX 8192    /* 4 strips (32 tokens each) x [2 blocks][32 tokens] f32 */
#define I8_XS 9216    /* same shape: -z * sum of the token's quants as i32 (Q4_1 / Q5_1: s = d * sum as f32) */
#define I8_WQ 10240   /* 128 rows x 2 blocks x 16 B */
#define I8_WQ2 14336  /* Q8_0: elements 16..31 */
#define I8_WH 18432   /* Q5: 128 rows x 2 u32 */
#define I8_WD 19456   /* 4 strips x 256 B: [row & 31][2] f16, duplicated in the upper half (lane-linear DMA) */
#define I8_WM 20480
#define I8_SLOT 21504
#ifndef I8_RING
#define I8_RING 6  /* stages in the ring: RING - 1 in flight ahead of the one being consumed */
#endif
#define I8_LDS (I8_RING * I8_SLOT)

struct MmqI8Args {
    QWeight w;
    const int8_t *x8;   // [N][nb][32] int8 quants of the activations (ggml element order)
    const float *dx;    // [N][nb] block scales as ggml stores them (Q8_0 kind: after the f16 round trip)
    const float *xs;    // [N][nb] zero-point term: -8*sum (Q4_0), -16*sum (Q5_0), 0 (Q8_0) as INT32 bits — it is the MFMA's
                        // C operand, so the zero point costs no instruction; d*sum as f32 for Q4_1 / Q5_1 (ggml's q8_1.s)
    float *dst;         // dst[n*ldd + m]
    int64_t ldd;
    int64_t M, N, nb;
    int tiles_n;
};

// activations -> int8 + f32 scale + zero-point term; 32 lanes per block (the arithmetic of k_quantize_act)
template <bool F16_D>
__global__ void __launch_bounds__(256) k_quant_act_i8(const char *__restrict__ x, int64_t nb_row, int64_t nblk, int64_t nrows,
                                                      float zp, int8_t *q8, float *dq, float *xs) {
    const int64_t gblock = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int l = threadIdx.x & 31;
    if (gblock >= nblk * nrows) return;
    const int64_t row = gblock / nblk, b = gblock % nblk;
    const float v = ((const float *)(x + row * nb_row))[b * 32 + l];
    float amax = fabsf(v);
    amax = g32_max_f32(amax);
    const float d = amax / 127.0f;
    const float id = act_id(amax, d, aq_scalar());
    const int q = act_q(v * id, aq_scalar());
    int s = q;
    s = g32_sum_i32(s);
    q8[gblock * 32 + l] = (int8_t)q;
    if (l == 0) {
        const float dd = F16_D ? round_f16(d) : d;
        dq[gblock] = dd;
        xs[gblock] = F16_D ? __builtin_bit_cast(float, -(int)zp * s) : (float)s * dd;  // Q8_1: s = sum * d (block_q8_1.s)
    }
}

template <int QT>
__device__ __forceinline__ constexpr int i8_group() {  // DMA instructions per stage and wave
    return 2 + 1 + 1 + 1 + (QT == QT_Q8_0 ? 1 : 0) + ((QT == QT_Q5_0 || QT == QT_Q5_1) ? 1 : 0) + 1 +
           ((QT == QT_Q4_1 || QT == QT_Q5_1) ? 1 : 0);
}

template <int QT>
__global__ void __launch_bounds__(256, I8_RING <= 3 ? 2 : 1) k_mmq_i8(const MmqI8Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    constexpr int G = i8_group<QT>();
    constexpr bool HAS_M = QT == QT_Q4_1 || QT == QT_Q5_1;

    const int t = xcd_tile_id(blockIdx.x, gridDim.x);
    const int tm = t / a.tiles_n, tn = t % a.tiles_n;
    const int64_t m0 = (int64_t)tm * MMQ_TM, n0 = (int64_t)tn * MMQ_TN;
    const int nstage_all = (int)(a.nb >> 1);
    const int per = (nstage_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int s_begin = (int)blockIdx.y * per, s_end = min(nstage_all, s_begin + per);
    const int nstage = s_end - s_begin;
    if (nstage <= 0) return;  // uniform

    // ---- per-lane DMA sources
    // X: instruction i (0..1) of wave w covers token rows 32w + 16i .. +15; lane -> row +(lane>>2); physical chunk
    //    lane&3 of the 64-byte row holds logical chunk (lane&3) ^ ((row>>2)&3)
    const char *xsrc[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int r = 32 * wave + 16 * i + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        xsrc[i] = (const char *)a.x8 + min(n0 + r, a.N - 1) * (a.nb * 32) + c * 16;
    }
    // per-token scalars: lane -> token 32w + (lane&31), block lane>>5 of the stage
    const int64_t tok = min(n0 + 32 * wave + (lane & 31), a.N - 1) * a.nb + (lane >> 5);
    // W: lane -> row 32w + (lane>>1), block lane&1 of the stage
    const int64_t wrow = min(m0 + 32 * wave + (lane >> 1), a.M - 1);
    const int64_t wblk0 = wrow * a.nb + (lane & 1);
    const int64_t drow = min(m0 + 32 * wave + (lane & 31), a.M - 1) * a.nb;

    auto issue = [&](int s, int ring_slot) {
        const int sc = min(s, nstage - 1);
        const int64_t kb = (int64_t)(s_begin + sc) * 2;
        char *slot = lds + ring_slot * I8_SLOT;
#pragma unroll
        for (int i = 0; i < 2; i++)
            __builtin_amdgcn_global_load_lds((gptr8_t)(xsrc[i] + kb * 32), (lptr8_t)(slot + I8_X + (32 * wave + 16 * i) * 64), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr8_t)(a.dx + tok + kb), (lptr8_t)(slot + I8_DX + wave * 256), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr8_t)(a.xs + tok + kb), (lptr8_t)(slot + I8_XS + wave * 256), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr8_t)(a.w.qs + (wblk0 + kb) * 16), (lptr8_t)(slot + I8_WQ + wave * 1024), 16, 0, 0);
        if constexpr (QT == QT_Q8_0)
            __builtin_amdgcn_global_load_lds((gptr8_t)(a.w.qs2 + (wblk0 + kb) * 16), (lptr8_t)(slot + I8_WQ2 + wave * 1024), 16, 0, 0);
        if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1)
            __builtin_amdgcn_global_load_lds((gptr8_t)(a.w.qh + wblk0 + kb), (lptr8_t)(slot + I8_WH + wave * 256), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr8_t)((const char *)a.w.d + (drow + kb) * 2), (lptr8_t)(slot + I8_WD + wave * 256), 4, 0, 0);
        if constexpr (HAS_M)
            __builtin_amdgcn_global_load_lds((gptr8_t)((const char *)a.w.m + (drow + kb) * 2), (lptr8_t)(slot + I8_WM + wave * 256), 4, 0, 0);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[j][i][r] = 0.0f;

    const int fl = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int p = 0; p < I8_RING - 1; p++) issue(p, p);
    int cur = 0;  // s mod RING
    for (int s = 0; s < nstage; s++) {
        // group s landed (s+1 .. s+RING-2 may be in flight); every wave is past its reads of the slot of stage s-1
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((I8_RING - 2) * G) : "memory");
        issue(s + I8_RING - 1, cur == 0 ? I8_RING - 1 : cur - 1);  // into the slot of stage s-1
        const char *slot = lds + cur * I8_SLOT;
        cur = cur == I8_RING - 1 ? 0 : cur + 1;
#pragma unroll 1
        for (int kb = 0; kb < 2; kb++) {  // not unrolled: both blocks' fragments live at once cost ~80 more registers
            // A fragments: token row R, 16 bytes = elements 16*fh.. of block kb: logical chunk kb*2 + fh
            i32x4 fa[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int R = wn * 64 + j * 32 + fl;
                const int p = (kb * 2 + fh) ^ ((R >> 2) & 3);
                fa[j] = *(const i32x4 *)(slot + I8_X + R * 64 + p * 16);
            }
            // B fragments + the row's scales
            i32x4 fb[2];
            float dwf[2], mwf[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int Rw = wm * 64 + i * 32 + fl;
                u32x4 q = *(const u32x4 *)(slot + I8_WQ + (Rw * 2 + kb) * 16);
                if constexpr (QT == QT_Q8_0) {
                    if (fh) q = *(const u32x4 *)(slot + I8_WQ2 + (Rw * 2 + kb) * 16);
#pragma unroll
                    for (int k = 0; k < 4; k++) fb[i][k] = (int)q[k];
                } else {
                    uint32_t h16 = 0;
                    if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) h16 = (*(const uint32_t *)(slot + I8_WH + (Rw * 2 + kb) * 4)) >> (16 * fh);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        uint32_t v = (q[k] >> (4 * fh)) & 0x0F0F0F0Fu;
                        if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) v |= spread_hi4((h16 >> (4 * k)) & 0xFu);
                        fb[i][k] = (int)v;
                    }
                }
                dwf[i] = (float)*(const _Float16 *)(slot + I8_WD + (Rw >> 5) * 256 + (Rw & 31) * 4 + kb * 2);
                mwf[i] = 0.0f;
                if constexpr (HAS_M) mwf[i] = (float)*(const _Float16 *)(slot + I8_WM + (Rw >> 5) * 256 + (Rw & 31) * 4 + kb * 2);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                // this lane's 16 tokens of tile j: local index (r&3) + 8*(r>>2) + 4*fh in strip wn*2 + j
                const char *sx = slot + I8_DX + (wn * 2 + j) * 256 + kb * 128 + fh * 16;
                const char *ss = slot + I8_XS + (wn * 2 + j) * 256 + kb * 128 + fh * 16;
                f32x4 dxq[4], xsf[4];
                i32x4 xsq[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    dxq[g4] = *(const f32x4 *)(sx + g4 * 32);
                    if constexpr (HAS_M) xsf[g4] = *(const f32x4 *)(ss + g4 * 32);            // s_x = d * sum (f32)
                    else if constexpr (QT != QT_Q8_0) xsq[g4] = *(const i32x4 *)(ss + g4 * 32);  // -z * sum (i32)
                }
                // the MFMA's C operand: the tokens' negated zero-point sums (codes are 0..15 / 0..31 here), 0 otherwise
                i32x16 cin;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    if constexpr (HAS_M || QT == QT_Q8_0) cin[r] = 0;
                    else cin[r] = xsq[r >> 2][r & 3];
                }
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const i32x16 c = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[j], fb[i], cin, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const float t = (float)c[r] * dxq[r >> 2][r & 3];
                        acc[j][i][r] = __builtin_fmaf(t, dwf[i], acc[j][i][r]);
                        if constexpr (HAS_M)  // + m_w * s_x
                            acc[j][i][r] = __builtin_fmaf(mwf[i], xsf[r >> 2][r & 3], acc[j][i][r]);
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the clamped tail DMAs before the workgroup's LDS is released

    const bool split = gridDim.y > 1;
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
                        a.dst[n * a.ldd + mrow] = acc[j][i][r];
                }
            }
        }
}
