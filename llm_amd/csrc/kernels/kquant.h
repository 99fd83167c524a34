// K-quant weights (Q4_K, Q6_K) on gfx950: upload re-layout, Q8_K activation quantizer, mat-vec, get_rows.
//
// Formats: the super-block structs the reference's generated bindings carry — block_q4_K
// (crates/ggml/sys/src/lib.rs:3103-3108: d, dmin f16, scales[12] six-bit packed, qs[128]), block_q6_K (:3240-3245: ql[128],
// qh[64], scales[16] i8, d f16), block_q8_K (:3303-3307: d f32, qs[256] i8, bsums[16] i16); QK_K = 256.  Arithmetic:
// ggml's k_quants.c (absent from the reference tree), restated in oracle/ggml_oracle.c
// (vec_dot_q4_K_q8_K / vec_dot_q6_K_q8_K / quantize_row_q8_K) which is what tests/test_kquant_gpu.py checks against:
//   Q4_K · Q8_K : sum_sb  d8·d·( sum_j sc_j·<q4_j, q8_j> ) − d8·dmin·( sum_j m_j·bsum32_j )      j = 8 sub-blocks of 32
//   Q6_K · Q8_K : sum_sb  d8·d·( sum_j sc_j·<q6_j − 32, q8_j> )                                  j = 16 sub-blocks of 16
// The integer parts are exact (v_dot4_i32_i8); the f32 scaling happens once per lane and 16-byte chunk and the lanes
// of a wave are summed by DPP — a different f32 summation order than ggml's 8 running lanes, same bound as the other
// mat-vecs (2e-5 · sum|w||x|).
//
// Device layout (planar, private to this backend; one "chunk" = 16 bytes of nibbles = what one lane dots per step):
//   Q4_K  qs [M][nsb][128]                      as in the block: chunk c = bytes 16c..: low nibbles = weights
//                                               64(c/2) + 16(c%2) + i, high nibbles = the same + 32
//         sc [M][nsb][16]  u8                   sc[0..7], m[0..7]: the 6-bit pairs unpacked once at upload
//         d  [M][nsb][2]   f16                  d, dmin
//   Q6_K  qs [M][nsb][128] (= ql)               chunk c: low nibbles = weights 128(c/4) + 16(c%4) + i, high = the same + 64
//         aux[M][nsb][8][2] u32                 per chunk hA, hB: the two high bits of the 16 low-nibble / high-nibble
//                                               weights, position p at bits 8(p%4) + 2(p/4): ((h >> 2k) & 0x03030303) << 4
//                                               drops them onto the four nibbles of dword k
//         sc [M][nsb][16]  i8                   sub-block scales
//         d  [M][nsb]      f16
// HBM bytes per super-block: Q4_K 148 (4 more than the file format: the unpacked scales), Q6_K 210.
#pragma once
#include "common.h"

enum { KT_Q4_K = 0, KT_Q6_K = 1 };

struct KWeight {
    const uint8_t *qs;
    const uint32_t *aux;
    const uint8_t *sc;
    const __half *d;
    int64_t M;    // rows
    int64_t nsb;  // super-blocks per row (ne0 / 256)
    int kt;
};

// Activations as Q8_K (quantize_row_q8_K_reference): q8 [N][K] i8, d8 [N][nsb] f32, bs [N][nsb][16] i16
struct KAct {
    const int8_t *q8;
    const float *d8;
    const int16_t *bs;
};

// ---- upload re-layout: one thread per chunk (8 per super-block) -------------------------------------------------
__global__ void k_relayout_k(const uint8_t *__restrict__ raw, int kt, int64_t nsbt /* super-blocks in total */, uint8_t *qs,
                             uint32_t *aux, uint8_t *sc, __half *d) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nsbt * 8) return;
    const int64_t sb = t >> 3;
    const int c = (int)(t & 7);
    if (kt == KT_Q4_K) {
        const uint8_t *b = raw + sb * 144;  // d(2) dmin(2) scales(12) qs(128)
        for (int i = 0; i < 16; i++) qs[sb * 128 + c * 16 + i] = b[16 + c * 16 + i];
        // get_scale_min_k4: j < 4: sc = q[j] & 63, m = q[j+4] & 63; else sc = (q[j+4] & 15) | (q[j-4] >> 6) << 4,
        //                                                         m  = (q[j+4] >> 4) | (q[j] >> 6) << 4
        const uint8_t *q = b + 4;
        uint8_t s, m;
        if (c < 4) {
            s = q[c] & 63;
            m = q[c + 4] & 63;
        } else {
            s = (uint8_t)((q[c + 4] & 0xF) | ((q[c - 4] >> 6) << 4));
            m = (uint8_t)((q[c + 4] >> 4) | ((q[c] >> 6) << 4));
        }
        sc[sb * 16 + c] = s;
        sc[sb * 16 + 8 + c] = m;
        if (c == 0) {
            d[sb * 2] = *(const __half *)b;
            d[sb * 2 + 1] = *(const __half *)(b + 2);
        }
    } else {
        const uint8_t *b = raw + sb * 210;  // ql(128) qh(64) scales(16) d(2)
        const int n2 = c >> 2, o = 16 * (c & 3);
        uint32_t hA = 0, hB = 0;
        for (int i = 0; i < 16; i++) {
            qs[sb * 128 + c * 16 + i] = b[64 * n2 + o + i];
            const uint8_t h = b[128 + 32 * n2 + (o & 31) + i];
            const uint32_t a2 = o < 32 ? (h & 3u) : ((h >> 2) & 3u), b2 = o < 32 ? ((h >> 4) & 3u) : ((h >> 6) & 3u);
            const int sh = 8 * (i & 3) + 2 * (i >> 2);
            hA |= a2 << sh;
            hB |= b2 << sh;
        }
        aux[(sb * 8 + c) * 2] = hA;
        aux[(sb * 8 + c) * 2 + 1] = hB;
        sc[sb * 16 + c] = b[192 + c];
        sc[sb * 16 + 8 + c] = b[192 + 8 + c];
        if (c == 0) d[sb] = *(const __half *)(b + 208);
    }
}

// ---- quantize_row_q8_K: one 256-thread workgroup per (super-block, row) ------------------------------------------
// max = the value of largest magnitude (first one in index order on ties), iscale = -128 / max,
// q = min(127, nearest_int(iscale * x)), bsums over 16, d = 1 / iscale; an all-zero block stores d = 0.
__global__ void __launch_bounds__(256) k_quant_q8k(const char *__restrict__ x, int64_t row_stride_bytes, int64_t nsb, int8_t *q8,
                                                   float *d8, int16_t *bs) {
    __shared__ float s_a[4];
    __shared__ int s_i[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t sb = blockIdx.x, n = blockIdx.y;
    const float v = ((const float *)(x + n * row_stride_bytes))[sb * 256 + tid];
    const float av = fabsf(v);
    float am = wave_max_f32(av);
    if (lane == 0) s_a[wave] = am;
    __syncthreads();
    am = fmaxf(fmaxf(s_a[0], s_a[1]), fmaxf(s_a[2], s_a[3]));
    int idx = av == am ? tid : 256;  // first index holding the extreme magnitude
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) idx = min(idx, __shfl_xor(idx, o, 64));
    if (lane == 0) s_i[wave] = idx;
    __syncthreads();
    idx = min(min(s_i[0], s_i[1]), min(s_i[2], s_i[3]));
    const float mx = ((const float *)(x + n * row_stride_bytes))[sb * 256 + (idx & 255)];
    int q = 0;
    float dd = 0.0f;
    if (am != 0.0f) {
        const float iscale = -128.0f / mx;
        q = min(127, __float2int_rn(iscale * v));
        dd = 1.0f / iscale;
    }
    q8[(n * nsb + sb) * 256 + tid] = (int8_t)q;
    const int s16 = g16_sum_i32(q);
    if ((tid & 15) == 0) bs[(n * nsb + sb) * 16 + (tid >> 4)] = (int16_t)s16;
    if (tid == 0) d8[n * nsb + sb] = dd;
}

// ---- mat-vec ---------------------------------------------------------------------------------------------------
struct MmvqKArgs {
    KWeight w;
    KAct x;
    float *dst;
    int64_t ldd;  // dst column stride in floats
    const float *res;  // nullable: dst = row sum + res (same layout as dst: the residual add that follows wo / w2; single matrix)
    // up to three matrices of ONE type sharing the activations in one launch (the K plan's wq|wk|wv and w1|w3): rows
    // [0, w.M) belong to w / dst, [w.M, w.M + wb.M) to wb / dst_b, the rest to wc / dst_c.  nseg <= 1: single matrix.
    int nseg;
    KWeight wb, wc;
    float *dst_b, *dst_c;
    int64_t ldd_b, ldd_c;
};
// global row of a multi-matrix launch -> (matrix, row inside it, its dst and column stride); wave-uniform
__device__ __forceinline__ void mmvq_k_select(const MmvqKArgs &a, int64_t row, KWeight &w, int64_t &lrow, float *&dst, int64_t &ldd) {
    w = a.w;
    lrow = row;
    dst = a.dst;
    ldd = a.ldd;
    if (a.nseg > 1 && row >= a.w.M) {
        if (a.nseg > 2 && row >= a.w.M + a.wb.M) {
            w = a.wc;
            lrow = row - a.w.M - a.wb.M;
            dst = a.dst_c;
            ldd = a.ldd_c;
        } else {
            w = a.wb;
            lrow = row - a.w.M;
            dst = a.dst_b;
            ldd = a.ldd_b;
        }
    }
}
__device__ __forceinline__ int64_t mmvq_k_rows(const MmvqKArgs &a) {
    return a.w.M + (a.nseg > 1 ? a.wb.M : 0) + (a.nseg > 2 ? a.wc.M : 0);
}

template <int KT>
struct KStep {  // what one lane holds of one step: its chunk + the super-block's scales
    u32x4 q;
    u32x4 sc;
    uint32_t hA, hB;  // Q6_K
    uint32_t dm;      // Q4_K: d | dmin << 16;  Q6_K: d
};

__device__ __forceinline__ int dot16(const u32x4 a, const i32x4 b, int acc) {
    acc = __builtin_amdgcn_sdot4((int)a[0], b[0], acc, false);
    acc = __builtin_amdgcn_sdot4((int)a[1], b[1], acc, false);
    acc = __builtin_amdgcn_sdot4((int)a[2], b[2], acc, false);
    return __builtin_amdgcn_sdot4((int)a[3], b[3], acc, false);
}

// 256-thread workgroups, one row per wave at a time (rows dealt round-robin over all waves of the grid), a step =
// 8 super-blocks (64 lanes x 16 bytes); the loads of step s+1 are requested before step s is dotted.  Activations:
// NCOLS Q8_K columns in LDS — q8 [NCOLS][K], d8 [NCOLS][nsb], bsums as i32 [NCOLS][nsb*16].
template <int KT, int NCOLS>
__global__ void __launch_bounds__(256) k_mmvq_k(const MmvqKArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nsb = (int)a.w.nsb, K = nsb * 256;
    int8_t *s_q = (int8_t *)smem;                        // [NCOLS][K]
    float *s_d = (float *)(smem + (size_t)NCOLS * K);    // [NCOLS][nsb]
    int *s_b = (int *)(s_d + NCOLS * nsb);               // [NCOLS][nsb*16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NCOLS * K / 16; i += 256) ((i32x4 *)s_q)[i] = ((const i32x4 *)a.x.q8)[i];
    for (int i = tid; i < NCOLS * nsb; i += 256) s_d[i] = a.x.d8[i];
    for (int i = tid; i < NCOLS * nsb * 16; i += 256) s_b[i] = (int)a.x.bs[i];
    __syncthreads();

    const int c = lane & 7, sbl = lane >> 3;
    const int nsteps = (nsb + 7) >> 3;
    const int64_t row0 = (int64_t)blockIdx.x * 4 + wave, rstride = (int64_t)gridDim.x * 4;
    const int64_t Mt = mmvq_k_rows(a);
    auto load = [&](KStep<KT> &st, int64_t grow, int s) {
        KWeight w;
        int64_t row, ldd_;
        float *dst_;
        mmvq_k_select(a, grow, w, row, dst_, ldd_);
        int sb = s * 8 + sbl;
        sb = sb < nsb ? sb : nsb - 1;  // lanes past the row end re-read the last super-block and are masked below
        const int64_t g = row * nsb + sb;
        st.q = __builtin_nontemporal_load((const u32x4 *)(w.qs + g * 128 + c * 16));
        st.sc = *(const u32x4 *)(w.sc + g * 16);
        if constexpr (KT == KT_Q4_K) {
            st.dm = *(const uint32_t *)(w.d + g * 2);
        } else {
            const u32x2 h = __builtin_nontemporal_load((const u32x2 *)(w.aux + (g * 8 + c) * 2));
            st.hA = h[0];
            st.hB = h[1];
            st.dm = (uint32_t) * (const uint16_t *)(w.d + g);
        }
    };
    KStep<KT> cur, nxt;
    if (row0 < Mt) load(cur, row0, 0);
    for (int64_t row = row0; row < Mt; row += rstride) {
        float acc[NCOLS];
#pragma unroll
        for (int n = 0; n < NCOLS; n++) acc[n] = 0.0f;
        for (int s = 0; s < nsteps; s++) {
            // request the next step (of this row, or the first of the wave's next row)
            const bool last = s + 1 == nsteps;
            const int64_t nrow = last ? row + rstride : row;
            if (nrow < Mt) load(nxt, nrow, last ? 0 : s + 1);
            const int sb = s * 8 + sbl;
            if (sb < nsb) {
                if constexpr (KT == KT_Q4_K) {
                    const int j = c >> 1, half = c & 1;
                    const uint32_t scw = j < 2 ? cur.sc[0] : cur.sc[1];
                    const int sc_lo = (int)((scw >> ((j & 1) * 16)) & 0xFF), sc_hi = (int)((scw >> ((j & 1) * 16 + 8)) & 0xFF);
                    const uint32_t mw = c < 4 ? cur.sc[2] : cur.sc[3];
                    const int mc = (int)((mw >> ((c & 3) * 8)) & 0xFF);
                    const float d = __half2float(__ushort_as_half((unsigned short)(cur.dm & 0xFFFF)));
                    const float dmin = __half2float(__ushort_as_half((unsigned short)(cur.dm >> 16)));
                    const u32x4 lo = cur.q & 0x0F0F0F0Fu, hi = (cur.q >> 4) & 0x0F0F0F0Fu;
#pragma unroll
                    for (int n = 0; n < NCOLS; n++) {
                        const int8_t *xq = s_q + (size_t)n * K + sb * 256 + 64 * j + 16 * half;
                        const i32x4 xl = *(const i32x4 *)xq, xh = *(const i32x4 *)(xq + 32);
                        const int isum = sc_lo * dot16(lo, xl, 0) + sc_hi * dot16(hi, xh, 0);
                        const int *bp = s_b + (n * nsb + sb) * 16 + 2 * c;  // sub-block c = 16-sums 2c, 2c+1
                        const int msum = mc * (bp[0] + bp[1]);
                        const float d8 = s_d[n * nsb + sb];
                        acc[n] += (d * d8) * (float)isum - (dmin * d8) * (float)msum;
                    }
                } else {
                    const int n2 = c >> 2, o = 16 * (c & 3);
                    const uint32_t wa = n2 ? cur.sc[2] : cur.sc[0], wb = n2 ? cur.sc[3] : cur.sc[1];
                    const int sc_a = (int)(int8_t)((wa >> (8 * (c & 3))) & 0xFF), sc_b = (int)(int8_t)((wb >> (8 * (c & 3))) & 0xFF);
                    const float d = __half2float(__ushort_as_half((unsigned short)cur.dm));
                    u32x4 lo, hi;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        lo[k] = (cur.q[k] & 0x0F0F0F0Fu) | (((cur.hA >> (2 * k)) & 0x03030303u) << 4);
                        hi[k] = ((cur.q[k] >> 4) & 0x0F0F0F0Fu) | (((cur.hB >> (2 * k)) & 0x03030303u) << 4);
                    }
#pragma unroll
                    for (int n = 0; n < NCOLS; n++) {
                        const int8_t *xq = s_q + (size_t)n * K + sb * 256 + 128 * n2 + o;
                        const i32x4 xl = *(const i32x4 *)xq, xh = *(const i32x4 *)(xq + 64);
                        const int *bp = s_b + (n * nsb + sb) * 16 + 8 * n2 + (c & 3);
                        const int isum = sc_a * (dot16(lo, xl, 0) - 32 * bp[0]) + sc_b * (dot16(hi, xh, 0) - 32 * bp[4]);
                        acc[n] += (d * s_d[n * nsb + sb]) * (float)isum;
                    }
                }
            }
            cur = nxt;
        }
#pragma unroll
        for (int n = 0; n < NCOLS; n++) {
            const float v = wave_sum_f32(acc[n]);
            if (lane == 0) {
                KWeight w_;
                int64_t lrow, ldd_;
                float *dst_;
                mmvq_k_select(a, row, w_, lrow, dst_, ldd_);
                dst_[(int64_t)n * ldd_ + lrow] = a.res ? v + a.res[(int64_t)n * ldd_ + lrow] : v;
            }
        }
    }
}

// ---- get_rows: one thread per chunk (32 outputs) -------------------------------------------------------------
// dequantize_row_q4_K: y = (d * sc) * q - (dmin * m);  dequantize_row_q6_K: y = d * sc * q   (left to right, no fma)
__global__ void k_get_rows_k(const KWeight w, const int *__restrict__ ids, float *dst, int64_t ldd) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= w.nsb * 8) return;
    const int64_t sb = t >> 3, row = ids ? ids[blockIdx.y] : (int64_t)blockIdx.y, g = row * w.nsb + sb;
    const int c = (int)(t & 7);
    float *y = dst + (int64_t)blockIdx.y * ldd + sb * 256;
    const uint8_t *q = w.qs + g * 128 + c * 16;
    if (w.kt == KT_Q4_K) {
        const int j = c >> 1, half = c & 1;
        const float d = __half2float(w.d[g * 2]), dmin = __half2float(w.d[g * 2 + 1]);
        const float d1 = d * (float)w.sc[g * 16 + 2 * j], m1 = dmin * (float)w.sc[g * 16 + 8 + 2 * j];
        const float d2 = d * (float)w.sc[g * 16 + 2 * j + 1], m2 = dmin * (float)w.sc[g * 16 + 8 + 2 * j + 1];
        for (int i = 0; i < 16; i++) {
            y[64 * j + 16 * half + i] = d1 * (float)(q[i] & 0xF) - m1;
            y[64 * j + 32 + 16 * half + i] = d2 * (float)(q[i] >> 4) - m2;
        }
    } else {
        const int n2 = c >> 2, o = 16 * (c & 3);
        const float d = __half2float(w.d[g]);
        const uint32_t hA = w.aux[(g * 8 + c) * 2], hB = w.aux[(g * 8 + c) * 2 + 1];
        const int sa = (int)(int8_t)w.sc[g * 16 + 8 * n2 + (c & 3)], sb2 = (int)(int8_t)w.sc[g * 16 + 8 * n2 + 4 + (c & 3)];
        for (int i = 0; i < 16; i++) {
            const int sh = 8 * (i & 3) + 2 * (i >> 2);
            const int qa = (int)((q[i] & 0xF) | (((hA >> sh) & 3u) << 4)) - 32;
            const int qb = (int)((q[i] >> 4) | (((hB >> sh) & 3u) << 4)) - 32;
            y[128 * n2 + o + i] = d * (float)sa * (float)qa;
            y[128 * n2 + 64 + o + i] = d * (float)sb2 * (float)qb;
        }
    }
}
