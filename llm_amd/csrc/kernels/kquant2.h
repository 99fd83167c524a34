// kquant2.h — the remaining K-quant weight types of the ABI (block_q2_K, block_q3_K, block_q5_K: crates/ggml/sys/src/lib.rs:2977,
// 3040, 3166; file types crates/llm-base/src/loader.rs:80-93) on gfx950: upload re-layout, mat-vec against Q8_K
// activations, get_rows; and for ALL five K types the resident f16 copy that puts their prompt batches on the MFMA GEMM.
//
// Arithmetic: ggml's k_quants.c (absent from the reference tree), restated in oracle/ggml_oracle.c
// (dequantize_row_q{2,3,5}_K, vec_dot_q{2,3,5}_K_q8_K), which is what tests/test_kquant_gpu.py checks against:
//   Q2_K . Q8_K : sum_sb  d8*d   *( sum_g (sc_g & 15) * <q2_g, q8_g> ) - d8*dmin*( sum_g (sc_g >> 4) * bsum_g )     g = 16 groups of 16
//   Q3_K . Q8_K : sum_sb  d8*d   *( sum_g (sc_g - 32) * <q3_g - 4, q8_g> )                                            q3 = 2 low bits | hmask bit << 2
//   Q5_K . Q8_K : sum_sb  d8*d   *( sum_j sc_j * <q5_j, q8_j> )          - d8*dmin*( sum_j m_j * bsum32_j )           j = 8 sub-blocks of 32
// Integer parts exact (v_dot4_i32_i8), f32 scaling per lane and 16-element group, lanes summed by DPP: the mat-vec bound
// of the other kernels (2e-5 * sum|w||x|).  These kernels are correct and coalesced, not tuned: 16 lanes share a
// super-block and several of them fetch the same 16-byte chunk (the 2-bit types pack four groups into one).
//
// Device layout (planar; KWeight of kquant.h): qs [M][nsb][64 | 128] the block's quant bytes unchanged; aux [M][nsb][8] u32 =
// hmask (Q3_K) / qh (Q5_K); sc [M][nsb][16] u8: Q2_K the 16 scale bytes, Q3_K the 16 six-bit scales unpacked (0..63),
// Q5_K sc[0..7], m[0..7] unpacked (as Q4_K); d [M][nsb][2] f16: d, dmin (Q3_K: d, 0).
#pragma once
#include "kquant.h"

enum { KT_Q2_K = 2, KT_Q3_K = 3, KT_Q5_K = 4 };

__host__ __device__ __forceinline__ int kt_qs_bytes(int kt) { return (kt == KT_Q2_K || kt == KT_Q3_K) ? 64 : 128; }

// one thread per super-block (upload time only)
__global__ void k_relayout_k2(const uint8_t *__restrict__ raw, int kt, int64_t nsbt, uint8_t *qs, uint32_t *aux, uint8_t *sc, __half *d) {
    const int64_t sb = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sb >= nsbt) return;
    uint8_t *aux8 = (uint8_t *)aux;
    if (kt == KT_Q2_K) {
        const uint8_t *b = raw + sb * 84;  // scales(16) qs(64) d(2) dmin(2)
        for (int i = 0; i < 16; i++) sc[sb * 16 + i] = b[i];
        for (int i = 0; i < 64; i++) qs[sb * 64 + i] = b[16 + i];
        d[sb * 2] = *(const __half *)(b + 80);
        d[sb * 2 + 1] = *(const __half *)(b + 82);
    } else if (kt == KT_Q3_K) {
        const uint8_t *b = raw + sb * 110;  // hmask(32) qs(64) scales(12) d(2)
        for (int i = 0; i < 32; i++) aux8[sb * 32 + i] = b[i];
        for (int i = 0; i < 64; i++) qs[sb * 64 + i] = b[32 + i];
        const uint8_t *ps = b + 96;
        for (int j = 0; j < 16; j++) {
            const int lo = j < 8 ? (ps[j] & 0xF) : (ps[j - 8] >> 4), hi = (ps[8 + (j & 3)] >> (2 * (j >> 2))) & 3;
            sc[sb * 16 + j] = (uint8_t)(lo | (hi << 4));
        }
        d[sb * 2] = *(const __half *)(b + 108);
        d[sb * 2 + 1] = __ushort_as_half((unsigned short)0);
    } else {
        const uint8_t *b = raw + sb * 176;  // d(2) dmin(2) scales(12) qh(32) qs(128)
        d[sb * 2] = *(const __half *)b;
        d[sb * 2 + 1] = *(const __half *)(b + 2);
        const uint8_t *q = b + 4;
        for (int c = 0; c < 8; c++) {
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
        }
        for (int i = 0; i < 32; i++) aux8[sb * 32 + i] = b[16 + i];
        for (int i = 0; i < 128; i++) qs[sb * 128 + i] = b[48 + i];
    }
}

// what a lane needs of one 16-element group g (0..15) of a super-block: its codes as 16 unsigned bytes + the group's terms
template <int KT>
struct KGroup {
    u32x4 w;        // codes: Q2_K 0..3, Q3_K 0..7 (value = code - 4), Q5_K 0..31
    int sc, mn;     // integer scale / min multiplier of the group
    float d, dmin;
};
template <int KT>
__device__ __forceinline__ KGroup<KT> k2_load(const KWeight &w, int64_t gsb /* row * nsb + sb */, int g) {
    KGroup<KT> r;
    if constexpr (KT == KT_Q2_K) {
        const u32x4 q = *(const u32x4 *)(w.qs + gsb * 64 + (2 * (g >> 3) + (g & 1)) * 16);
        const int shift = 2 * ((g >> 1) & 3);
        r.w = (q >> shift) & 0x03030303u;
        const int s = w.sc[gsb * 16 + g];
        r.sc = s & 15;
        r.mn = s >> 4;
    } else if constexpr (KT == KT_Q3_K) {
        const u32x4 q = *(const u32x4 *)(w.qs + gsb * 64 + (2 * (g >> 3) + (g & 1)) * 16);
        const u32x4 hm = *(const u32x4 *)((const uint8_t *)w.aux + gsb * 32 + (g & 1) * 16);
        const int shift = 2 * ((g >> 1) & 3), bit = g >> 1;
        r.w = ((q >> shift) & 0x03030303u) | (((hm >> bit) & 0x01010101u) << 2);
        r.sc = (int)w.sc[gsb * 16 + g] - 32;
        r.mn = 0;
    } else {
        const int j64 = g >> 2, hi = (g >> 1) & 1, half = g & 1;
        const u32x4 q = *(const u32x4 *)(w.qs + gsb * 128 + j64 * 32 + half * 16);
        const u32x4 qh = *(const u32x4 *)((const uint8_t *)w.aux + gsb * 32 + half * 16);
        const u32x4 nib = hi ? ((q >> 4) & 0x0F0F0F0Fu) : (q & 0x0F0F0F0Fu);
        r.w = nib | (((qh >> (2 * j64 + hi)) & 0x01010101u) << 4);
        r.sc = w.sc[gsb * 16 + (g >> 1)];
        r.mn = w.sc[gsb * 16 + 8 + (g >> 1)];
    }
    r.d = __half2float(w.d[gsb * 2]);
    r.dmin = __half2float(w.d[gsb * 2 + 1]);
    return r;
}

// 256-thread workgroups, one row per wave at a time, 16 lanes per super-block (4 super-blocks per step).  Activations as
// in k_mmvq_k: NCOLS Q8_K columns in LDS — q8 [NCOLS][K], d8 [NCOLS][nsb], bsums as i32 [NCOLS][nsb*16].
template <int KT, int NCOLS>
__global__ void __launch_bounds__(256) k_mmvq_k2(const MmvqKArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nsb = (int)a.w.nsb, K = nsb * 256;
    int8_t *s_q = (int8_t *)smem;
    float *s_d = (float *)(smem + (size_t)NCOLS * K);
    int *s_b = (int *)(s_d + NCOLS * nsb);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NCOLS * K / 16; i += 256) ((i32x4 *)s_q)[i] = ((const i32x4 *)a.x.q8)[i];
    for (int i = tid; i < NCOLS * nsb; i += 256) s_d[i] = a.x.d8[i];
    for (int i = tid; i < NCOLS * nsb * 16; i += 256) s_b[i] = (int)a.x.bs[i];
    __syncthreads();
    const int g = lane & 15, sbl = lane >> 4;
    const int nsteps = (nsb + 3) >> 2;
    const int64_t Mt = mmvq_k_rows(a);
    for (int64_t grow = (int64_t)blockIdx.x * 4 + wave; grow < Mt; grow += (int64_t)gridDim.x * 4) {
        KWeight w;
        int64_t row, ldd;
        float *dst;
        mmvq_k_select(a, grow, w, row, dst, ldd);
        float acc[NCOLS];
#pragma unroll
        for (int n = 0; n < NCOLS; n++) acc[n] = 0.0f;
        for (int s = 0; s < nsteps; s++) {
            const int sb = s * 4 + sbl;
            if (sb < nsb) {
                const KGroup<KT> r = k2_load<KT>(w, row * nsb + sb, g);
#pragma unroll
                for (int n = 0; n < NCOLS; n++) {
                    const i32x4 x = *(const i32x4 *)(s_q + (size_t)n * K + sb * 256 + 16 * g);
                    const int isum = dot16(r.w, x, 0), bsum = s_b[(n * nsb + sb) * 16 + g];
                    const float d8 = s_d[n * nsb + sb];
                    if constexpr (KT == KT_Q3_K)
                        acc[n] += (r.d * d8) * (float)(r.sc * (isum - 4 * bsum));
                    else
                        acc[n] += (r.d * d8) * (float)(r.sc * isum) - (r.dmin * d8) * (float)(r.mn * bsum);
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NCOLS; n++) {
            const float v = wave_sum_f32(acc[n]);
            if (lane == 0) dst[(int64_t)n * ldd + row] = a.res ? v + a.res[(int64_t)n * ldd + row] : v;
        }
    }
}

// get_rows / dequantization to f32: one thread per 16-element group, the operation order of dequantize_row_q{2,3,5}_K
// (dl = d * sc first, then dl * q - ml).  rows: ids[blockIdx.y], or blockIdx.y itself when ids == nullptr.
template <int KT>
__global__ void k_get_rows_k2(const KWeight w, const int *__restrict__ ids, float *dst, int64_t ldd) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= w.nsb * 16) return;
    const int64_t sb = t >> 4, row = ids ? ids[blockIdx.y] : (int64_t)blockIdx.y;
    const int g = (int)(t & 15);
    const KGroup<KT> r = k2_load<KT>(w, row * w.nsb + sb, g);
    float *y = dst + (int64_t)blockIdx.y * ldd + sb * 256 + 16 * g;
    const float dl = r.d * (float)r.sc, ml = r.dmin * (float)r.mn;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int code = (int)((r.w[i >> 2] >> (8 * (i & 3))) & 0xFF);
        if constexpr (KT == KT_Q3_K)
            y[i] = dl * (float)(code - 4);
        else
            y[i] = dl * (float)code - ml;
    }
}

// ---- K-quant prompt batches on the f16 GEMM (kernels/mmq_w16.h, mmq_w16_256.h) ----------------------------------------
// The resident f16 copy of a K-quant weight: its dequantized values (f32, the decoders above / k_get_rows_k) rounded to
// f16 and stored in the GEMM's k order (mmq_kperm inside every 32 elements).  src: [rows][K] f32, out: [rows][K] f16.
__global__ void __launch_bounds__(256) k_f32_to_w16(const float *__restrict__ src, int64_t n32 /* 32-element blocks */, _Float16 *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t blk = t >> 5;
    if (blk >= n32) return;
    const int e = (int)(t & 31);
    float v = src[blk * 32 + e];
    v = fminf(fmaxf(v, -65504.0f), 65504.0f);
    out[blk * 32 + mmq_kperm_inv(e)] = (_Float16)v;
}
// The activation operand of that GEMM: the row quantized as Q8_K (quantize_row_q8_K: what ggml's K-quant dots consume),
// then f16(d8 * q) in the same k order.  One 256-thread workgroup per (super-block, row), as k_quant_q8k.
__global__ void __launch_bounds__(256) k_quant_act_f16_k(const char *__restrict__ x, int64_t row_stride_bytes, int64_t nsb, _Float16 *__restrict__ out) {
    __shared__ float s_a[4];
    __shared__ int s_i[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t sb = blockIdx.x, n = blockIdx.y;
    const float *xr = (const float *)(x + n * row_stride_bytes) + sb * 256;
    const float v = xr[tid];
    const float av = fabsf(v);
    float am = wave_max_f32(av);
    if (lane == 0) s_a[wave] = am;
    __syncthreads();
    am = fmaxf(fmaxf(s_a[0], s_a[1]), fmaxf(s_a[2], s_a[3]));
    int idx = av == am ? tid : 256;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) idx = min(idx, __shfl_xor(idx, o, 64));
    if (lane == 0) s_i[wave] = idx;
    __syncthreads();
    idx = min(min(s_i[0], s_i[1]), min(s_i[2], s_i[3]));
    const float mx = xr[idx & 255];
    float r = 0.0f;
    if (am != 0.0f) {
        const float iscale = -128.0f / mx;
        const int q = min(127, __float2int_rn(iscale * v));
        r = (1.0f / iscale) * (float)q;
    }
    r = fminf(fmaxf(r, -65504.0f), 65504.0f);
    out[(n * nsb + sb) * 256 + (tid & ~31) + mmq_kperm_inv(tid & 31)] = (_Float16)r;
}
