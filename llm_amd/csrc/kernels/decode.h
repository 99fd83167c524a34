// decode.h — fused kernels of the single-token (N=1) decode plan: what the LLaMA graph of
// crates/models/llama/src/lib.rs:174-352 collapses to on MI355X once the ~37 ggml nodes per layer are
// regrouped around the HBM-bound mat-vecs (launch-bound decode):
//
//   k_rmsnorm_quant      rms_norm → ·weight → re-quantize to Q8_0/Q8_1 (the INIT phase of the next mul_mat)
//   k_mmvq_dec<EPI_QKV>  wq|wk|wv mat-vec → RoPE(Q,K) → Q to f32, K/V to the f16 KV cache (V transposed)
//   k_attn_decode        K·Q → scale → mask → softmax → V·P → merge heads → re-quantize (input of wo)
//   k_mmvq_dec<EPI_ADD>  wo / w2 mat-vec + residual add
//   k_mmvq_dec<EPI_GATE> w1|w3 mat-vec → silu(w1x)·(w3x)
//
// Arithmetic is the same as the generic kernels (ops.h / mmvq.h), i.e. ggml's CPU semantics; only the
// grouping differs.  Every kernel reads the position from device memory (DecParams) so that a captured
// hipGraph can be replayed for the next token without patching kernel arguments.
#pragma once
#include "common.h"
#include "mmvq.h"
#include "ops.h"

struct DecParams {
    int n_past;   // tokens already in the KV cache = position of the (first) token being evaluated
    int token;    // its id (row of tok_embeddings)
    int store_at; // k_mmvq_big / k_qkv_attn: cache position the token's K / V rows are WRITTEN to; 0 = n_past (always, outside the
                  // roofline leg: a replay of the mat-vec class alone runs on stale activations and parks its rows in the last slot)
    int pad1;
    int tokens[32];  // multi-token plan (2..31 tokens of a prompt chunk): ids of all tokens, tokens[0] == token
};

// Greedy sampling on the device: token = first index of the maximum logit — what a host loop
// `if (l[i] > l[best]) best = i` returns — written straight into the decode parameters of the NEXT replay of the plan
// (token id, position + 1) and into out[0], so that a chain of greedy tokens needs neither the 128 KB logits
// read-back nor a host round trip per token.  One 1024-thread workgroup.
__global__ void __launch_bounds__(1024) k_argmax_next(const float *__restrict__ logits, int V, DecParams *prm,
                                                      int *__restrict__ out) {
    __shared__ float s_v[16];
    __shared__ int s_i[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    // eight independent loads in flight per thread, then the scan in index order (one load per dependent compare made this
    // kernel a chain of 32 L2 round trips: ~25 us of the device-sampled chain and of the speculative next token, per token)
    for (int i0 = tid; i0 < V; i0 += 8 * 1024) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i0 + u * 1024;
            v8[u] = i < V ? logits[i] : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {  // increasing i: strict > keeps the first maximum of this thread's subsequence
            const int i = i0 + u * 1024;
            if (i < V && (v8[u] > bv || bi == 0x7fffffff)) {
                bv = v8[u];
                bi = i;
            }
        }
    }
    auto better = [](float v, int i, float bv_, int bi_) { return v > bv_ || (v == bv_ && i < bi_); };
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        if (better(ov, oi, bv, bi)) {
            bv = ov;
            bi = oi;
        }
    }
    if (lane == 0) {
        s_v[wave] = bv;
        s_i[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; w++)
            if (better(s_v[w], s_i[w], bv, bi)) {
                bv = s_v[w];
                bi = s_i[w];
            }
        prm->token = bi;
        prm->tokens[0] = bi;
        prm->n_past = prm->n_past + 1;
        out[0] = bi;
    }
}

// (cos, sin) of the RoPE angle of pair kk of a head at this token's position, as ggml's rope computes them:
// theta_kk = freq_scale*p * theta_scale^kk by repeated f32 multiplication (reference ggml.c rope, f32 branch).  One
// launch per token; the wq|wk|wv mat-vec of every layer reads the table instead of re-deriving it (64 dependent
// multiplies + sincos on two waves held up the whole workgroup's first barrier by ~2.5 us per layer).
// Block 0 also opens the token's epoch (`epoch`, nullable): the tag of the granules that k_qkv_attn's mat-vec workgroups hand
// to its attention workgroups (kernels/decode_fused.h) — bumped on the device, once per token, so that a replayed hipGraph
// never meets its own previous granules as current; 0 is skipped (a zeroed granule is never valid).
__global__ void __launch_bounds__(128) k_rope_table(const DecParams *__restrict__ prm, float theta_scale, float freq_scale,
                                                    int half_d, float *__restrict__ out, unsigned *epoch = nullptr) {
    const int kk = threadIdx.x, c = blockIdx.x;  // block c: token c of a prompt chunk (position n_past + c), 128 floats each
    if (epoch && kk == 0 && c == 0) {
        const unsigned e = *epoch + 1u;
        *epoch = e ? e : 1u;
    }
    if (kk >= half_d) return;
    float theta = freq_scale * (float)(prm->n_past + c);
    for (int t = 0; t < kk; t++) theta *= theta_scale;
    out[c * 128 + 2 * kk] = cosf(theta);
    out[c * 128 + 2 * kk + 1] = sinf(theta);
}

// ---------------------------------------------------------------------------------------------------
// rms_norm (f64 Σx², eps) → multiply by weight → optional f32 copy → Q8 re-quantization, one 1024-thread
// workgroup for the single activation row.
// ---------------------------------------------------------------------------------------------------
template <bool F16_D>
__device__ __forceinline__ void rmsnorm_quant_body(const float *__restrict__ x, const float *__restrict__ w, float eps, int E,
                                                   float *y_f32 /*nullable*/, int8_t *lo, int8_t *hi, float *dq, int *sumq,
                                                   float *dT /* [E/32][8] copy of the scales, row r in column r */,
                                                   int *sT /* ... of the sums (k_mmq_cols stages them by DMA) */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_y = (float *)smem;                 // E floats
    __shared__ double s_part[16];
    const int tid = threadIdx.x;
    {  // one workgroup per activation row (blockIdx.x; a single row for decode, 2..8 for a prompt chunk)
        const int64_t r = blockIdx.x;
        x += r * E;
        if (y_f32) y_f32 += r * E;
        lo += r * (E / 2);
        hi += r * (E / 2);
        dq += r * (E / 32);
        sumq += r * (E / 32);
    }
    // the row and the norm weight are fetched once, all loads in flight together (E <= 8192: 8 elements per thread; the
    // sums run in the order of the plain loops they replace)
    constexpr int PT = 8;
    const bool in_regs = E <= PT * 1024;
    float xv[PT], wv[PT];
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < PT; k++) {
            const int i = tid + k * 1024;
            xv[k] = i < E ? x[i] : 0.0f;
            wv[k] = i < E ? w[i] : 0.0f;
        }
    }
    double s = 0.0;
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < PT; k++)
            if (tid + k * 1024 < E) s += (double)(xv[k] * xv[k]);
    } else {
        for (int i = tid; i < E; i += 1024) {
            const float v = x[i];
            s += (double)(v * v);
        }
    }
    s = wave_sum_f64(s);
    if ((tid & 63) == 0) s_part[tid >> 6] = s;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) tot += s_part[i];
    const float mean = (float)(tot / (double)E);
    const float scale = 1.0f / sqrtf(mean + eps);
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < PT; k++) {
            const int i = tid + k * 1024;
            if (i < E) {
                const float v = (xv[k] * scale) * wv[k];
                s_y[i] = v;
                if (y_f32) y_f32[i] = v;
            }
        }
    } else {
        for (int i = tid; i < E; i += 1024) {
            const float v = (x[i] * scale) * w[i];
            s_y[i] = v;
            if (y_f32) y_f32[i] = v;
        }
    }
    __syncthreads();
    const int nblk = E / 32, l = tid & 31;
    for (int b = tid >> 5; b < nblk; b += 32) {
        const float v = s_y[b * 32 + l];
        float amax = fabsf(v);
        amax = g32_max_f32(amax);
        const float d = amax / 127.0f;
        const float id = act_id(amax, d, aq_scalar());
        const int q = act_q(v * id, aq_scalar());
        int sq = q;
        sq = g32_sum_i32(sq);
        (l < 16 ? lo : hi)[b * 16 + (l & 15)] = (int8_t)q;
        if (l == 0) {
            dq[b] = F16_D ? round_f16(d) : d;
            sumq[b] = sq;
            if (dT) {  // rows 8p .. 8p+7 form table p
                const int64_t o = (int64_t)(blockIdx.x >> 3) * nblk * 8 + b * 8 + (blockIdx.x & 7);
                dT[o] = F16_D ? round_f16(d) : d;
                sT[o] = sq;
            }
        }
    }
}
template <bool F16_D>
__global__ void __launch_bounds__(1024) k_rmsnorm_quant(const float *__restrict__ x, const float *__restrict__ w, float eps, int E,
                                                        float *y_f32, int8_t *lo, int8_t *hi, float *dq, int *sumq,
                                                        float *dT = nullptr, int *sT = nullptr) {
    rmsnorm_quant_body<F16_D>(x, w, eps, E, y_f32, lo, hi, dq, sumq, dT, sT);
}

// The norm of a prompt chunk keeps N (<= 8) of the chip's CUs busy for ~6 us.  The other workgroups of the launch — one per remaining
// CU — spend it pulling the leading 16-row groups of every workgroup of the NEXT launch (k_mmq_cols: wq|wk|wv or w1|w3, dealt in
// contiguous runs of groups: workgroup c owns gq (+1 if c < gr) groups from c * gq + min(c, gr)) into the L2 of the XCD that
// workgroup will run on: workgroup b of a one-workgroup-per-CU launch sits on XCD b mod 8 (llama_plan.inc xcd_labels_probe), and an
// XCD's L2 keeps its lines across the kernel boundary.  One 4-byte LDS-DMA load per 128-byte line (nothing to wait for but the
// wave's end), as warm_next does for decode (kernels/decode_fused.h).
constexpr int CW_MAX = 12;  // arrays: qs, d (+ qs2 | qh | m) of up to three matrices
struct ColsWarm {
    const uint8_t *base[CW_MAX];
    uint32_t row_bytes[CW_MAX];
    int row0[CW_MAX], rows[CW_MAX];  // the array's matrix in the next launch's row space: first row, row count (w1 and w3 both start at 0)
    int n;                           // arrays in use; 0 = off
    int G, gq, gr;                   // the next launch's grid and dealing
    int wg;                          // leading groups of every workgroup to warm
};
__device__ __forceinline__ void cols_warm(const ColsWarm &cw, const int first /* first warming workgroup of the launch */, unsigned *junk) {
    typedef const __attribute__((address_space(1))) void *wg_ptr;
    typedef __attribute__((address_space(3))) void *wl_ptr;
    const int b = (int)blockIdx.x, x = b & 7, tid = (int)threadIdx.x;
    const int b0 = first + ((x - first) & 7);                                // the first warming workgroup with this label
    const int j = (b - b0) >> 3, nj = ((int)gridDim.x - b0 + 7) >> 3;        // this one among them
    for (int c = x + 8 * j; c < cw.G; c += 8 * nj) {                         // the next launch's workgroups on this XCD, dealt round robin
        const int g0 = c * cw.gq + (c < cw.gr ? c : cw.gr), nlg = cw.gq + (c < cw.gr ? 1 : 0);
        const int R0 = 16 * g0, R1 = 16 * (g0 + (nlg < cw.wg ? nlg : cw.wg));
        for (int a = 0; a < cw.n; a++) {
            const int lo_r = (R0 > cw.row0[a] ? R0 : cw.row0[a]) - cw.row0[a];
            const int hi_r = (R1 < cw.row0[a] + cw.rows[a] ? R1 : cw.row0[a] + cw.rows[a]) - cw.row0[a];
            if (lo_r >= hi_r) continue;
            const uint8_t *p = cw.base[a] + (size_t)lo_r * cw.row_bytes[a];
            const uint32_t bytes = (uint32_t)(hi_r - lo_r) * cw.row_bytes[a];
            for (uint32_t off = (uint32_t)tid << 7; off < bytes; off += 1024u << 7)
                __builtin_amdgcn_global_load_lds((wg_ptr)(p + off), (wl_ptr)junk, 4, 0, 0);
        }
    }
}
template <bool F16_D>
__global__ void __launch_bounds__(1024) k_rmsnorm_quant_warm(const float *__restrict__ x, const float *__restrict__ w, float eps, int E,
                                                             float *y_f32, int8_t *lo, int8_t *hi, float *dq, int *sumq, float *dT,
                                                             int *sT, const int nrows, const ColsWarm cw) {
    if ((int)blockIdx.x >= nrows) {
        __shared__ unsigned s_junk[64];
        cols_warm(cw, nrows, s_junk);
        return;
    }
    rmsnorm_quant_body<F16_D>(x, w, eps, E, y_f32, lo, hi, dq, sumq, dT, sT);
}

// re-quantize a plain f32 row (the FFN gate before w2): 32 lanes per block
template <bool F16_D>
__global__ void __launch_bounds__(256) k_quant_row(const float *__restrict__ x, int nblk, int8_t *lo, int8_t *hi,
                                                   float *dq, int *sumq, float *dT = nullptr, int *sT = nullptr) {
    const int b = (blockIdx.x * 256 + threadIdx.x) >> 5, l = threadIdx.x & 31;
    if (b >= nblk) return;
    {  // blockIdx.y = activation row
        const int64_t r = blockIdx.y;
        x += r * nblk * 32;
        lo += r * nblk * 16;
        hi += r * nblk * 16;
        dq += r * nblk;
        sumq += r * nblk;
    }
    const float v = x[b * 32 + l];
    float amax = fabsf(v);
    amax = g32_max_f32(amax);
    const float d = amax / 127.0f;
    const float id = act_id(amax, d, aq_scalar());
    const int q = act_q(v * id, aq_scalar());
    int sq = q;
    sq = g32_sum_i32(sq);
    (l < 16 ? lo : hi)[b * 16 + (l & 15)] = (int8_t)q;
    if (l == 0) {
        dq[b] = F16_D ? round_f16(d) : d;
        sumq[b] = sq;
        if (dT) {
            const int64_t o = (int64_t)(blockIdx.y >> 3) * nblk * 8 + b * 8 + (blockIdx.y & 7);
            dT[o] = F16_D ? round_f16(d) : d;
            sT[o] = sq;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// decode mat-vec with fused epilogues.  One wave = one PAIR of consecutive rows (m0, m0+1) of each
// weight matrix it serves; 4 waves per workgroup; x staged in LDS exactly like k_mmvq.
// ---------------------------------------------------------------------------------------------------
enum { EPI_STORE = 0, EPI_ADD = 1, EPI_GATE = 2, EPI_QKV = 3 };

struct DecMmvqArgs {
    QWeight w[3];        // STORE/ADD: w[0]; GATE: w[0]=w1, w[1]=w3; QKV: wq, wk, wv
    int wg_begin[3];     // QKV: first workgroup of each segment
    QAct x;              // XSRC_Q8: activation already re-quantized (attention output, final norm)
    const float *xf;     // XSRC_NORM / XSRC_F32: f32 activation row, re-quantized by every workgroup while staging
    const float *xw;     // XSRC_NORM: the norm weight;  eps below
    float eps;
    int64_t nb;          // K/32
    float *dst;          // STORE/ADD/GATE output (f32); QKV: Q output [E] f32
    const float *res;    // ADD: residual
    // QKV epilogue
    const DecParams *prm;
    __half *mem_k;       // + layer offset: K element (pos p, chan c) at p*Egqa + c
    __half *mem_v;       // + layer offset: V element (chan c, pos p) at c*C + p
    int64_t Egqa, C;
    int D;               // head size (rope row length)
    float theta_scale, freq_scale;
};

// Where the activation of a decode mat-vec comes from.
//   XSRC_Q8   : pre-quantized planar Q8 blocks in global memory (copied into LDS)
//   XSRC_NORM : f32 residual row → rms_norm (f64 Σx², eps) → ·weight → Q8, computed by EVERY workgroup while it
//               stages x into LDS.  Redundant (each WG re-reads 16 KB from L2 and spends ~100 VALU ops/thread)
//               but it removes the separate norm+quantize launch (≈6.5 µs + a kernel boundary per use, twice
//               per layer) from the serial decode chain.
//   XSRC_F32  : plain f32 row → Q8 (the FFN gate feeding w2), same idea.
enum { XSRC_Q8 = 0, XSRC_NORM = 1, XSRC_F32 = 2 };

// thread t of the 256 owns elements 4i..4i+3 with i = it*256 + t; a Q8 block = 8 consecutive threads
template <bool F16_D, bool SC>
__device__ __forceinline__ void quant4_to_lds_t(const f32x4 v, int i4, int nb, int tid, i32x4 *s_lo, i32x4 *s_hi,
                                                float *s_d, int *s_sum) {
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = g8_max_f32(amax);
    const float d = amax / 127.0f;
    const float id = act_id<SC>(amax, d);
    const int q0 = act_q<SC>(v[0] * id), q1 = act_q<SC>(v[1] * id), q2 = act_q<SC>(v[2] * id), q3 = act_q<SC>(v[3] * id);
    int sq = (q0 + q1) + (q2 + q3);
    sq = g8_sum_i32(sq);
    const int b = i4 >> 3;  // block index (32-bit: 64-bit index math costs the staging several VALU ops per element group)
    if (b >= nb) return;
    const int j = tid & 7;
    const int packed = (q0 & 0xFF) | ((q1 & 0xFF) << 8) | ((q2 & 0xFF) << 16) | ((int)((unsigned)q3 << 24));
    ((int *)(j < 4 ? s_lo : s_hi))[b * 4 + (j & 3)] = packed;
    if (j == 0) {
        s_d[b] = F16_D ? round_f16(d) : d;
        s_sum[b] = sq;
    }
}
// the quantizer branch (common.h: act_quant) is wave-uniform: one scalar branch around two straight-line bodies
template <bool F16_D>
__device__ __forceinline__ void quant4_to_lds(const f32x4 v, int i4, int nb, int tid, i32x4 *s_lo, i32x4 *s_hi,
                                              float *s_d, int *s_sum) {
    if (aq_scalar())
        quant4_to_lds_t<F16_D, true>(v, i4, nb, tid, s_lo, s_hi, s_d, s_sum);
    else
        quant4_to_lds_t<F16_D, false>(v, i4, nb, tid, s_lo, s_hi, s_d, s_sum);
}

template <bool F16_D, int XSRC>
__device__ __forceinline__ void stage_x(const QAct &xq, const float *xf, const float *xw, float eps, int64_t nb,
                                        int tid, i32x4 *s_lo, i32x4 *s_hi, float *s_d, int *s_sum) {
    if constexpr (XSRC == XSRC_Q8) {
        for (int64_t i = tid; i < nb; i += 256) {
            s_lo[i] = xq.lo[i];
            s_hi[i] = xq.hi[i];
            s_d[i] = xq.d[i];
            s_sum[i] = xq.sum[i];
        }
    } else if constexpr (XSRC == XSRC_F32) {
        const int64_t n4 = nb * 8;  // float4 count
        for (int64_t i4 = tid; i4 < ((n4 + 255) & ~(int64_t)255); i4 += 256) {
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (i4 < n4) v = ((const f32x4 *)xf)[i4];
            quant4_to_lds<F16_D>(v, (int)i4, (int)nb, tid, s_lo, s_hi, s_d, s_sum);
        }
    } else {
        __shared__ double s_part[4];
        constexpr int MAXIT = 8;  // rows up to 8192 wide stay in registers between the two passes
        const int64_t n4 = nb * 8;
        f32x4 v[MAXIT];
        double ss = 0.0;
#pragma unroll
        for (int it = 0; it < MAXIT; it++) {
            const int64_t i4 = (int64_t)it * 256 + tid;
            v[it] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (i4 < n4) {
                v[it] = ((const f32x4 *)xf)[i4];
                ss += (double)(v[it][0] * v[it][0]);
                ss += (double)(v[it][1] * v[it][1]);
                ss += (double)(v[it][2] * v[it][2]);
                ss += (double)(v[it][3] * v[it][3]);
            }
        }
        ss = wave_sum_f64(ss);
        if ((tid & 63) == 0) s_part[tid >> 6] = ss;
        __syncthreads();
        const double tot = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        const float mean = (float)(tot / (double)(nb * 32));
        const float scale = 1.0f / sqrtf(mean + eps);
#pragma unroll
        for (int it = 0; it < MAXIT; it++) {
            const int64_t i4 = (int64_t)it * 256 + tid;
            if (it * 256 >= n4) break;  // uniform
            f32x4 y = {0.0f, 0.0f, 0.0f, 0.0f};
            if (i4 < n4) {
                const f32x4 w4 = ((const f32x4 *)xw)[i4];
                y[0] = (v[it][0] * scale) * w4[0];
                y[1] = (v[it][1] * scale) * w4[1];
                y[2] = (v[it][2] * scale) * w4[2];
                y[3] = (v[it][3] * scale) * w4[3];
            }
            quant4_to_lds<F16_D>(y, (int)i4, (int)nb, tid, s_lo, s_hi, s_d, s_sum);
        }
    }
}

// One K-step (block column b) of rows (m0, m0+1) of NW weight matrices: the raw loads, kept in registers.
template <int QT, int NW>
struct DecRegs {
    u32x4 q[NW][2], p[NW][2];
    uint32_t h[NW][2];
    float dw[NW][2], mw[NW][2];
    __device__ __forceinline__ void load(const QWeight *w, const int64_t (&r)[NW][2], int64_t b) {
#pragma unroll
        for (int i = 0; i < NW; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int64_t o = r[i][j] + b;
                q[i][j] = __builtin_nontemporal_load((const u32x4 *)(w[i].qs + o * 16));
                if constexpr (QT == QT_Q8_0)
                    p[i][j] = __builtin_nontemporal_load((const u32x4 *)(w[i].qs2 + o * 16));
                else
                    p[i][j] = q[i][j];
                if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1)
                    h[i][j] = __builtin_nontemporal_load(w[i].qh + o);
                else
                    h[i][j] = 0;
                dw[i][j] = __half2float(w[i].d[o]);
                if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1)
                    mw[i][j] = __half2float(w[i].m[o]);
                else
                    mw[i][j] = 0.0f;
            }
    }
};

// rows (m0, m0+1) of NW weight matrices (same K, same activation) in ONE pass over the blocks, so that all
// 2*NW row streams have their loads in flight together.  `cur` holds the first K-step, loaded by the caller
// BEFORE the activation was staged (the weight stream does not depend on x); inside the loop the loads of
// step b+64 are issued before step b is consumed (register double buffering).
template <int QT, int NW>
__device__ __forceinline__ void dec_rows2(const QWeight *w, const int64_t (&r)[NW][2], DecRegs<QT, NW> cur, int64_t nb,
                                          int lane, const i32x4 *s_lo, const i32x4 *s_hi, const float *s_d,
                                          const int *s_sum, float (&acc)[NW][2]) {
#pragma unroll
    for (int i = 0; i < NW; i++) acc[i][0] = acc[i][1] = 0.0f;
    for (int64_t b = lane; b < nb; b += 64) {
        DecRegs<QT, NW> nxt = cur;
        if (b + 64 < nb) nxt.load(w, r, b + 64);
        const i32x4 lo = s_lo[b], hi = s_hi[b];
        const float xd = s_d[b];
        const int xs = s_sum[b];
#pragma unroll
        for (int i = 0; i < NW; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
                acc[i][j] += block_dot<QT>(cur.q[i][j], cur.p[i][j], cur.h[i][j], cur.dw[i][j], cur.mw[i][j], lo, hi, xd, xs);
        cur = nxt;
    }
#pragma unroll
    for (int i = 0; i < NW; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = wave_sum_f32(acc[i][j]);
}

template <int QT, int EPI, int XSRC>
__global__ void __launch_bounds__(256) k_mmvq_dec(const DecMmvqArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t nb = a.nb;
    i32x4 *s_lo = (i32x4 *)smem;
    i32x4 *s_hi = s_lo + nb;
    float *s_d = (float *)(s_hi + nb);
    int *s_sum = (int *)(s_d + nb);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr bool F16_D = QT == QT_Q4_0 || QT == QT_Q5_0 || QT == QT_Q8_0;  // vec_dot_type Q8_0 vs Q8_1
    constexpr int NW = EPI == EPI_GATE ? 2 : 1;
    int sg = 0;
    if constexpr (EPI == EPI_QKV) {
        if ((int)blockIdx.x >= a.wg_begin[1]) sg = 1;
        if ((int)blockIdx.x >= a.wg_begin[2]) sg = 2;
    }
    const int wg = blockIdx.x - (EPI == EPI_QKV ? a.wg_begin[sg] : 0);
    const int64_t m0 = ((int64_t)wg * 4 + wave) * 2;
    const QWeight *w = &a.w[sg];
    const bool valid = m0 < w->M;
    // first K-step of the weight stream: issued before x is staged, so HBM latency overlaps the staging
    int64_t r[NW][2];
    DecRegs<QT, NW> first;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        r[i][0] = (valid ? m0 : 0) * nb;
        r[i][1] = (m0 + 1 < w[i].M ? m0 + 1 : w[i].M - 1) * nb;
    }
    if (valid && lane < nb)
        first.load(w, r, lane);
    else
        first = DecRegs<QT, NW>{};

    stage_x<F16_D, XSRC>(a.x, a.xf, a.xw, a.eps, nb, tid, s_lo, s_hi, s_d, s_sum);
    __syncthreads();
    if (!valid) return;
    const bool has1 = m0 + 1 < w->M;
    float acc[NW][2];
    dec_rows2<QT, NW>(w, r, first, nb, lane, s_lo, s_hi, s_d, s_sum, acc);
    const float v0 = acc[0][0], v1 = acc[0][1];
    float u0 = 0.0f, u1 = 0.0f;
    if constexpr (EPI == EPI_GATE) {
        u0 = acc[1][0];
        u1 = acc[1][1];
    }
    (void)u0;
    (void)u1;

    if constexpr (EPI == EPI_STORE) {
        if (lane == 0) {
            a.dst[m0] = v0;
            if (has1) a.dst[m0 + 1] = v1;
        }
    } else if constexpr (EPI == EPI_ADD) {
        if (lane == 0) {
            a.dst[m0] = v0 + a.res[m0];
            if (has1) a.dst[m0 + 1] = v1 + a.res[m0 + 1];
        }
    } else if constexpr (EPI == EPI_GATE) {
        if (lane == 0) {
            a.dst[m0] = silu_table(v0) * u0;
            if (has1) a.dst[m0 + 1] = silu_table(v1) * u1;
        }
    } else {  // EPI_QKV
        if (lane != 0) return;
        const int p = a.prm->n_past;
        if (sg == 2) {  // V: f16 into the transposed cache (llama lib.rs:234-244)
            a.mem_v[m0 * a.C + p] = __float2half_rn(v0);
            if (has1) a.mem_v[(m0 + 1) * a.C + p] = __float2half_rn(v1);
            return;
        }
        // RoPE mode 0 on the adjacent pair (m0, m0+1): theta = freq_scale*p * theta_scale^k, iterated product
        const int k = (int)(m0 % a.D) >> 1;
        float theta = a.freq_scale * (float)p;
        for (int j = 0; j < k; j++) theta *= a.theta_scale;
        const float c = cosf(theta), s = sinf(theta);
        const float r0 = v0 * c - v1 * s, r1 = v0 * s + v1 * c;
        if (sg == 0) {
            a.dst[m0] = r0;
            a.dst[m0 + 1] = r1;
        } else {  // K: f16, contiguous run at position p (llama lib.rs:228-243)
            a.mem_k[(int64_t)p * a.Egqa + m0] = __float2half_rn(r0);
            a.mem_k[(int64_t)p * a.Egqa + m0 + 1] = __float2half_rn(r1);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// decode attention for one query token: one 1024-thread workgroup (16 waves) per head.
//   s_t = Σ_d K[t][d]·f16(q[d])  (f32 accumulate)  for t = 0..P   (P = n_past, the new token included)
//   p   = softmax(s·scale) with ggml's f16-rounded exp, then rounded to f16 (src1 of the V matmul)
//   o_d = Σ_t V[d][t]·p_t ; the head's D outputs are re-quantized to Q8 blocks for the wo mat-vec and
//   optionally written as f32 (merge-heads layout [E]).
// K: [C][Egqa] f16 (per layer), V: [Egqa][C] f16 (per layer).  Dynamic LDS: (Clds + D) floats + Clds halves, Clds >= the
// live length the launch can meet (the plan sizes it by the context, or by the split threshold when longer rows go to
// kernels/decode_attn_split.h).
//
// At short context the kernel is a pure latency chain (a few tens of KB per head), so its shape is dictated by
// round trips and instruction issue, not bandwidth (in-kernel timeline at 135 positions: position -> K/V loads
// 5.3 us, V·P with 8 channels per wave and 3/4 of the lanes idle 2.6 us):
//   * the FIRST 256 positions of K and V, q and the position itself are all requested at kernel entry — the
//     loads do not wait for n_past (rows past the current position are read and ignored; they exist, the cache is
//     allocated for C positions), so short contexts cost ONE memory round trip;
//   * scores : 64 groups of 16 lanes, a lane holds 8 dims (one 16-byte load per position), 4 positions per group
//              in flight → 256 positions per pass;
//   * V·P    : wave w owns channels 8w..8w+7, 8 lanes per channel, a lane covers 8 consecutive positions per
//              64-position pass → all lanes busy from 64 positions on, 8-lane DPP reduction at the end.
// ---------------------------------------------------------------------------------------------------
template <bool F16_D>
__global__ void __launch_bounds__(1024) k_attn_decode(const float *__restrict__ q, const __half *__restrict__ mem_k,
                                                      const __half *__restrict__ mem_v, const DecParams *prm,
                                                      float scale, int D, int n_rep /* H / Hkv */, int64_t Egqa,
                                                      int64_t C, float *out_f32, int8_t *lo, int8_t *hi, float *dq,
                                                      int *sumq, long long *ts, int n_head,
                                                      int64_t Clds /* positions the LDS arrays hold (<= C, % 8 == 0) */,
                                                      float *dT = nullptr, int *sT = nullptr /* as k_rmsnorm_quant */) {
    const long long t_entry = ts ? (long long)wall_clock64() : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_s = (float *)smem;  // Clds scores
    float *s_o = s_s + Clds;     // D outputs
    _Float16 *s_p = (_Float16 *)(s_o + D);  // Clds probabilities as f16 (src1 of the V matmul); Clds % 8 == 0
    __shared__ float s_red[16];
    __shared__ double s_redd[16];
    const int h = blockIdx.x, hk = h / n_rep;
    const int qn = blockIdx.y;  // query token of a prompt chunk (0 for decode): position n_past + qn, row qn of q / outputs
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n_past = prm->n_past + qn;  // requested first; nothing below waits for it until the masks are needed
    const int64_t Eq = (int64_t)n_head * D;
    const float *qh = q + qn * Eq + (int64_t)h * D;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- speculative first pass (positions 0..255 of K and V) + q
    const int g = tid >> 4, gl = tid & 15;  // scores: 64 groups of 16 lanes, lane gl owns dims gl*8 .. gl*8+7
    const int d0 = gl * 8;                  // D <= 128 and D % 8 == 0 (checked by the plan builder)
    const bool act = d0 < D;
    const __half *kbase = mem_k + (int64_t)hk * D + d0;
    f16x8 kv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int t = g + 64 * u;
        kv[u] = zero8;
        if (act && t < C) kv[u] = *(const f16x8 *)(kbase + (int64_t)t * Egqa);
    }
    const int cv = wave * 8 + (lane >> 3), pj = (lane & 7) * 8;  // V·P: channel, first position inside a 64-pass
    const bool vact = cv < D;
    const __half *vbase = mem_v + ((int64_t)hk * D + cv) * C + pj;
    f16x8 vv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        vv[u] = zero8;
        if (vact && 64 * u + pj + 8 <= C) vv[u] = *(const f16x8 *)(vbase + 64 * u);
    }
    f32x4 q0 = {0.0f, 0.0f, 0.0f, 0.0f}, q1 = q0;
    if (act) {
        q0 = *(const f32x4 *)(qh + d0);
        q1 = *(const f32x4 *)(qh + d0 + 4);
    }
    f16x2 qh2[4];  // ggml rounds src1 (Q) to f16; the products below are exact in f32 (v_dot2_f32_f16)
#pragma unroll
    for (int j = 0; j < 2; j++) {
        qh2[j] = f16x2{(_Float16)q0[2 * j], (_Float16)q0[2 * j + 1]};
        qh2[2 + j] = f16x2{(_Float16)q1[2 * j], (_Float16)q1[2 * j + 1]};
    }
    const int T = n_past + 1;
    const int T8 = (T + 7) & ~7;
    const long long t_loaded = ts ? (long long)wall_clock64() : 0;

    // ---- scores ----
#pragma unroll 1
    for (int t0 = g; t0 < T; t0 += 256) {
        if (t0 != g) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = t0 + 64 * u;
                kv[u] = zero8;
                if (act && t < T) kv[u] = *(const f16x8 *)(kbase + (int64_t)t * Egqa);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = t0 + 64 * u;
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; j++) s = __builtin_amdgcn_fdot2(f16x2{kv[u][2 * j], kv[u][2 * j + 1]}, qh2[j], s, false);
            s = g16_sum_f32(s);
            if (gl == 0 && t < T) s_s[t] = s * scale;
        }
    }
    __syncthreads();
    const long long t_scores = ts ? (long long)wall_clock64() : 0;
    // ---- softmax over T entries (ggml: max, f16-rounded exp of f16-rounded (x-max), f64 sum, scale by 1/sum) ----
    // Up to 256 positions one wave does it alone (4 per lane, DPP reductions): no exchange through LDS, no barrier between the
    // passes — attn_consumer's form (kernels/decode_fused.h).  The maximum and the f64 sum of f16-valued terms are exact, so
    // their order is immaterial: the same bits as the 16-wave form below.
    if (T <= 256) {
        if (wave == 0) {
            float sv[4], e[4];
            float mx1 = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int t = lane + 64 * i;
                sv[i] = t < T ? s_s[t] : -INFINITY;
                mx1 = fmaxf(mx1, sv[i]);
            }
            mx1 = wave_max_f32(mx1);
            double sum1 = 0.0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                e[i] = lane + 64 * i < T ? round_f16(expf(round_f16(sv[i] - mx1))) : 0.0f;
                sum1 += (double)e[i];
            }
            sum1 = wave_sum_f64(sum1);
            const float inv1 = (float)(1.0 / sum1);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int t = lane + 64 * i;
                if (t < T8) s_p[t] = t < T ? (_Float16)(e[i] * inv1) : (_Float16)0.0f;
            }
        }
    } else {
    float mx = -INFINITY;
    for (int t = tid; t < T; t += 1024) mx = fmaxf(mx, s_s[t]);
    mx = wave_max_f32(mx);
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = s_red[0];
#pragma unroll
    for (int i = 1; i < 16; i++) mx = fmaxf(mx, s_red[i]);
    double sum = 0.0;
    for (int t = tid; t < T; t += 1024) {
        const float e = round_f16(expf(round_f16(s_s[t] - mx)));
        s_s[t] = e;
        sum += (double)e;
    }
    sum = wave_sum_f64(sum);
    if (lane == 0) s_redd[wave] = sum;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) tot += s_redd[i];
    const float inv = (float)(1.0 / tot);
    for (int t = tid; t < T8; t += 1024)
        s_p[t] = t < T ? (_Float16)(s_s[t] * inv) : (_Float16)0.0f;  // probabilities as f16 (src1 of V·P); padding = 0
    }
    __syncthreads();
    const long long t_softmax = ts ? (long long)wall_clock64() : 0;
    // ---- V·P ----
    {
        float acc = 0.0f;
#pragma unroll 1
        for (int p0 = 0; p0 < T8; p0 += 256) {
            if (p0 != 0) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    vv[u] = zero8;
                    if (vact && p0 + 64 * u + pj < T8) vv[u] = *(const f16x8 *)(vbase + p0 + 64 * u);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int pos = p0 + 64 * u + pj;
                if (pos < T8) {  // 8-position chunks past the context are never touched (their V is not ours)
                    const f16x8 pp = *(const f16x8 *)(s_p + pos);
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc = __builtin_amdgcn_fdot2(f16x2{vv[u][2 * j], vv[u][2 * j + 1]}, f16x2{pp[2 * j], pp[2 * j + 1]}, acc, false);
                }
            }
        }
        acc = g8_sum_f32(acc);
        if ((lane & 7) == 0 && vact) s_o[cv] = acc;
    }
    __syncthreads();
    const long long t_vp = ts ? (long long)wall_clock64() : 0;
    // ---- outputs: f32 (merged heads) + Q8 blocks (D/32 blocks per head) ----
    const int nblk = D / 32, l = tid & 31, b = tid >> 5;
    if (b < nblk) {
        const float v = s_o[b * 32 + l];
        if (out_f32) out_f32[qn * Eq + (int64_t)h * D + b * 32 + l] = v;
        float amax = fabsf(v);
        amax = g32_max_f32(amax);
        const float d = amax / 127.0f;
        const float id = act_id(amax, d, aq_scalar());
        const int qv = act_q(v * id, aq_scalar());
        int sq = qv;
        sq = g32_sum_i32(sq);
        const int64_t gb = qn * (Eq / 32) + (int64_t)h * nblk + b;
        (l < 16 ? lo : hi)[gb * 16 + (l & 15)] = (int8_t)qv;
        if (l == 0) {
            dq[gb] = F16_D ? round_f16(d) : d;
            sumq[gb] = sq;
            if (dT) {
                const int64_t o = (int64_t)(qn >> 3) * (Eq / 32) * 8 + ((int64_t)h * nblk + b) * 8 + (qn & 7);
                dT[o] = F16_D ? round_f16(d) : d;
                sT[o] = sq;
            }
        }
    }
    if (ts && tid == 0) {
        const int q4 = (int)gridDim.x / 4;
        if (q4 > 0 && h % q4 == 0 && h / q4 < 4) {
            long long *o = ts + (h / q4) * 8;
            o[0] = t_entry; o[1] = t_loaded; o[2] = t_scores; o[3] = t_softmax; o[4] = t_vp;
            o[5] = (long long)wall_clock64(); o[6] = T; o[7] = h;
        }
    }
}
