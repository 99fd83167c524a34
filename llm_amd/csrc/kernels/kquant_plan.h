// kquant_plan.h — the helper launches of the K plan (llama_plan.inc plan_launch_k): a LLaMA whose matrices are Q2_K … Q6_K
// (file types crates/llm-base/src/loader.rs:80-93, block structs crates/ggml/sys/src/lib.rs:2977-3303) decodes — and takes prompt
// chunks of up to 8 tokens, the reference's default n_batch — as 10-13 launches per layer from a captured hipGraph instead of ~34
// eager ones on the node-by-node executor.
//
// Every kernel here performs the executor's floating-point operations in the executor's order on the same values — the
// rms_norm of k_rms_norm<true> (thread t adds elements t, t + 256, … in f64, waves by DPP, (s0 + s1) + (s2 + s3)), the Q8_K
// quantizer of k_quant_q8k (quantize_row_q8_K: first value of largest magnitude, iscale = -128 / max, nearest_int, bsums), the
// f16-table SiLU of k_unary, the RoPE table of k_rope_table (ggml's iterated f32 product), RNE f16 cache stores.  The mat-vecs
// are the executor's own k_mmvq_k / k_mmvq_k2 (a row's sum does not depend on the grid or on the column chunking), with the
// residual add in their epilogue.  Only the attention differs (k_attn_decode sums a head's scores and V.P in another order than
// the executor's three generic launches): plan and executor agree to ~5e-7 of the logits' scale (tests/test_kquant_plan_gpu.py);
// the executor's parity with the oracle is tests/test_kquant_gpu.py's subject.
#pragma once
#include "decode.h"
#include "kquant2.h"

// one super-block (256 values, one per thread of a 256-thread workgroup) -> Q8_K: k_quant_q8k's arithmetic on a register value
__device__ __forceinline__ void q8k_quant_block(const float v, const int tid, float *s_v /* 256 floats of LDS */, int8_t *q8,
                                                float *d8, int16_t *bs) {
    __shared__ float s_a[4];
    __shared__ int s_i[4];
    const int wave = tid >> 6, lane = tid & 63;
    s_v[tid] = v;
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
    const float mx = s_v[idx & 255];
    int q = 0;
    float dd = 0.0f;
    if (am != 0.0f) {
        const float iscale = -128.0f / mx;
        q = min(127, __float2int_rn(iscale * v));
        dd = 1.0f / iscale;
    }
    q8[tid] = (int8_t)q;
    const int s16 = g16_sum_i32(q);
    if ((tid & 15) == 0) bs[tid >> 4] = (int16_t)s16;
    if (tid == 0) d8[0] = dd;
}

// The helper kernels take blockIdx.y = the activation row (token of a prompt chunk; one row for decode); Q8_K rows are laid out as
// KAct wants them: q8 [N][K], d8 [N][nsb], bs [N][nsb][16], nsb = gridDim.x.
//
// rms_norm(x) * w -> Q8_K, one workgroup per super-block; every workgroup sums the whole row (E floats out of L2) the way
// k_rms_norm does, so all of them hold the same scale.  y (nullable): f32 copy of the normed rows (the embeddings output).
__global__ void __launch_bounds__(256) k_k_norm_quant(const float *__restrict__ x, const float *__restrict__ w, float eps, int E,
                                                      float *__restrict__ y, int8_t *q8, float *d8, int16_t *bs) {
    __shared__ double s_part[4];
    __shared__ float s_v[256];
    const int tid = threadIdx.x, sb = blockIdx.x, nsb = gridDim.x;
    const int64_t n = blockIdx.y;
    x += n * E;
    if (y) y += n * E;
    q8 += n * E;
    d8 += n * nsb;
    bs += n * nsb * 16;
    double s = 0.0;
    for (int i = tid; i < E; i += 256) {
        const float v = x[i];
        s += (double)(v * v);
    }
    s = wave_sum_f64(s);
    if ((tid & 63) == 0) s_part[tid >> 6] = s;
    __syncthreads();
    const double tot = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    const float mean = (float)(tot / (double)E);
    const float scale = 1.0f / sqrtf(mean + eps);
    const int i = sb * 256 + tid;
    float v = x[i] * scale;
    v = v * w[i];
    if (y) y[i] = v;
    q8k_quant_block(v, tid, s_v, q8 + (size_t)sb * 256, d8 + sb, bs + (size_t)sb * 16);
}

// plain f32 row -> Q8_K (the attention output before wo)
__global__ void __launch_bounds__(256) k_k_quant(const float *__restrict__ x, int8_t *q8, float *d8, int16_t *bs) {
    __shared__ float s_v[256];
    const int tid = threadIdx.x, sb = blockIdx.x, nsb = gridDim.x;
    const int64_t n = blockIdx.y, K = (int64_t)nsb * 256;
    q8k_quant_block(x[n * K + sb * 256 + tid], tid, s_v, q8 + n * K + (size_t)sb * 256, d8 + n * nsb + sb, bs + (n * nsb + sb) * 16);
}

// silu(g1) * g3 -> Q8_K (ggml_silu's f16 table, then the ggml_mul: crates/models/llama/src/lib.rs:328-330)
__global__ void __launch_bounds__(256) k_k_silu_mul_quant(const float *__restrict__ g1, const float *__restrict__ g3, int8_t *q8,
                                                          float *d8, int16_t *bs) {
    __shared__ float s_v[256];
    const int tid = threadIdx.x, sb = blockIdx.x, nsb = gridDim.x;
    const int64_t n = blockIdx.y, K = (int64_t)nsb * 256;
    const int64_t i = n * K + sb * 256 + tid;
    float v = silu_table(g1[i]);
    v = v * g3[i];
    q8k_quant_block(v, tid, s_v, q8 + n * K + (size_t)sb * 256, d8 + n * nsb + sb, bs + (n * nsb + sb) * 16);
}

// RoPE (mode 0, adjacent pairs, the token's (cos, sin) table of k_rope_table) on Q in place and on K; K -> f16 run at the
// token's position, V -> f16 scatter into the transposed cache (crates/models/llama/src/lib.rs:191-244).
struct KRopeStoreArgs {
    float *q;               // [N][E] rotated in place
    const float *k, *v;     // [N][Egqa] each
    const float *rope;      // per token (blockIdx.y): D/2 (cos, sin) pairs, 128 floats apart (k_rope_table)
    const DecParams *prm;
    __half *mem_k, *mem_v;  // + layer offset
    int64_t E, Egqa, C;
    int D;
};
__global__ void __launch_bounds__(256) k_k_rope_store(const KRopeStoreArgs a) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nq = a.E >> 1, nk = a.Egqa >> 1;
    const int64_t n = blockIdx.y;  // token of the chunk: position n_past + n, row n of q / k / v, table n
    const int p = a.prm->n_past + (int)n;
    float *const q = a.q + n * a.E;
    const float *const kr = a.k + n * a.Egqa, *const vr = a.v + n * a.Egqa, *const rope = a.rope + n * 128;
    if (t < nq + nk) {
        const bool is_k = t >= nq;
        const int64_t m0 = 2 * (is_k ? t - nq : t);
        const float *src = is_k ? kr : q;
        const int kk = (int)(m0 % a.D) >> 1;
        const float c = rope[2 * kk], s = rope[2 * kk + 1];
        const float v0 = src[m0], v1 = src[m0 + 1];
        const float r0 = v0 * c - v1 * s, r1 = v0 * s + v1 * c;
        if (is_k) {
            a.mem_k[(int64_t)p * a.Egqa + m0] = __float2half_rn(r0);
            a.mem_k[(int64_t)p * a.Egqa + m0 + 1] = __float2half_rn(r1);
        } else {
            q[m0] = r0;
            q[m0 + 1] = r1;
        }
    } else if (t < nq + nk + a.Egqa) {
        const int64_t m = t - nq - nk;
        a.mem_v[m * a.C + p] = __float2half_rn(vr[m]);
    }
}
