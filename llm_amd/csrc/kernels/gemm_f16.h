// gemm_f16.h — batched GEMM with f16 src0 and f32 src1 on the matrix cores: the two attention products of a
// prompt batch,  KQ = mul_mat(K, Q)  and  KQV = mul_mat(V^T, softmax(KQ))
// (crates/models/llama/src/lib.rs:264-296 build them as GGML_OP_MUL_MAT over views of memory_k / memory_v).
//
// ggml's CPU path converts each src1 row to f16 (vec_dot_type of F16 is F16) and accumulates f16 x f16 products
// in f32; this kernel does the same with v_mfma_f32_32x32x16_f16 — only the order of the f32 sum differs.
//
//   dst[b][n][m] = sum_k A[b / r][m][k] * f16(B[b][n][k])        A: f16, k contiguous, rows / batches strided
//                                                                B: f32, k contiguous, rows / batches strided
// Same 128 x 128 x 64 workgroup tile, LDS layout and MFMA schedule as kernels/mmq.h (mma_stage_128x128); rows
// beyond M / N and the k tail are zero-filled in LDS (the V cache beyond n_past may hold anything, so tail
// elements are SELECTED to zero, not multiplied).  grid = (tiles_m * tiles_n, batches).
#pragma once
#include "mmq.h"

typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

struct GemmF16Args {
    const char *a;
    int64_t a_nb1, a_nb2, a_nb3;
    const char *b;
    int64_t b_nb1, b_nb2, b_nb3;
    char *d;
    int64_t d_nb1, d_nb2, d_nb3;
    int64_t M, N, K;
    int64_t ne12, r2, r3;  // batch dims of src1; broadcast ratios src1/src0 (GQA)
    int tiles_n;
    // causal structure of a prompt batch (the prompt plan; 0 = plain GEMM).  Row n of src1 is query n at position
    // causal_past + n and may look at keys 0 .. causal_past + n.
    //   1 (K.Q: M runs over keys): tiles whose keys all lie beyond the last query of the tile are skipped — the softmax
    //     never reads those scores.   2 (V.P: K runs over keys): the k loop stops after the last key any query of the
    //     tile can see — the probabilities beyond are exact zeros.
    int causal;
    int causal_past;
};

__device__ __forceinline__ u32x4 gf16_mask_tail(u32x4 v, int valid /*elements 0..8*/) {
    // keep the first `valid` f16 of the 8
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const int e = valid - 2 * w;  // elements of this dword that are valid: <=0, 1, >=2
        v[w] = e >= 2 ? v[w] : e == 1 ? (v[w] & 0xFFFFu) : 0u;
    }
    return v;
}

template <bool B16>
__device__ __forceinline__ void gf16_load(u32x4 (&ra)[4], u32x4 (&rb)[4], const GemmF16Args &g, const char *ab,
                                          const char *bb, int64_t m0, int64_t n0, int64_t k0, int tid) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = (tid >> 3) + 32 * i, kc = tid & 7;
        const int64_t k = k0 + kc * 8;
        const int valid = (int)min((int64_t)8, g.K - k);
        u32x4 va = {0, 0, 0, 0}, vb = {0, 0, 0, 0};
        if (valid > 0) {
            if (m0 + row < g.M) {
                va = *(const u32x4 *)(ab + (m0 + row) * g.a_nb1 + k * 2);  // host checked 16-B alignment
                if (valid < 8) va = gf16_mask_tail(va, valid);
            }
            if (B16) {  // src1 already converted (rows 16-byte aligned: host checked)
                if (n0 + row < g.N) {
                    vb = *(const u32x4 *)(bb + (n0 + row) * g.b_nb1 + k * 2);
                    if (valid < 8) vb = gf16_mask_tail(vb, valid);
                }
            } else if (n0 + row < g.N) {
                const float *bp = (const float *)(bb + (n0 + row) * g.b_nb1) + k;
                float f[8];
                if (valid == 8) {
                    const f32x4_u x0 = *(const f32x4_u *)bp, x1 = *(const f32x4_u *)(bp + 4);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        f[e] = x0[e];
                        f[4 + e] = x1[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++) f[e] = e < valid ? bp[e] : 0.0f;
                }
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const f16x2 h = {(_Float16)f[2 * w], (_Float16)f[2 * w + 1]};
                    vb[w] = __builtin_bit_cast(uint32_t, h);
                }
            }
        }
        ra[i] = va;
        rb[i] = vb;
    }
}

template <bool B16>
__device__ __forceinline__ void gemm_f16_body(const GemmF16Args &g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int t = xcd_tile_id(blockIdx.x, gridDim.x);
    const int tm = t / g.tiles_n, tn = t % g.tiles_n;
    const int64_t m0 = (int64_t)tm * 128, n0 = (int64_t)tn * 128;
    const int64_t i12 = blockIdx.y % g.ne12, i13 = blockIdx.y / g.ne12;
    const char *ab = g.a + (i12 / g.r2) * g.a_nb2 + (i13 / g.r3) * g.a_nb3;
    const char *bb = g.b + i12 * g.b_nb2 + i13 * g.b_nb3;
    char *db = g.d + i12 * g.d_nb2 + i13 * g.d_nb3;

    const int64_t last_key = (int64_t)g.causal_past + n0 + 127;  // the last key a query of this tile can see
    if (g.causal == 1 && m0 > last_key) return;                 // uniform
    f32x16 acc[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[j][i][r] = 0.0f;

    const int64_t K_eff = g.causal == 2 ? min(g.K, last_key + 1) : g.K;
    const int nstage = (int)((K_eff + 63) >> 6);
    u32x4 ra[4], rb[4];
    gf16_load<B16>(ra, rb, g, ab, bb, m0, n0, 0, tid);
    for (int s = 0; s < nstage; s++) {
        char *W = lds + (s & 1) * 2 * MMQ_TILEB, *X = W + MMQ_TILEB;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int off = ((tid >> 3) + 32 * i) * MMQ_ROWB + (tid & 7) * 16;
            *(u32x4 *)(W + off) = ra[i];
            *(u32x4 *)(X + off) = rb[i];
        }
        __syncthreads();
        if (s + 1 < nstage) gf16_load<B16>(ra, rb, g, ab, bb, m0, n0, (int64_t)(s + 1) * 64, tid);
        mma_stage_128x128(W, X, lane, wm, wn, acc);
    }
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int64_t m = m0 + wm * 64 + i * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t n = n0 + wn * 64 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < g.M && n < g.N) *(float *)(db + n * g.d_nb1 + m * 4) = acc[j][i][r];
            }
        }
}

__global__ void __launch_bounds__(256, 2) k_gemm_f16(const GemmF16Args g) { gemm_f16_body<false>(g); }
// src1 given as f16 (the prompt plan's probabilities, written as f16 by k_p_soft_max)
__global__ void __launch_bounds__(256, 2) k_gemm_f16_b16(const GemmF16Args g) { gemm_f16_body<true>(g); }
