// prompt_attn.h — attention of a prompt batch in ONE launch: K.Q, scale + causal mask + softmax and V.P with the scores kept
// in LDS (crates/models/llama/src/lib.rs:246-299: mul_mat(K, Q) -> scale -> diag_mask_inf -> soft_max -> mul_mat(V, P) ->
// permute -> cpy, 7 graph nodes per layer).  The prompt plan ran them as three launches that moved the [H][N][T] scores
// through HBM twice (k_gemm_f16, k_p_soft_max, k_gemm_f16_b16: ~70 us of a 7B layer's ~480 at N = 512).
//
// ggml's softmax fixes its rounding points (row max first, e = f16(exp(f16(s*scale - max))), exact f64 sum, p = e * (1/sum),
// src1 of the V mat-mul rounded to f16), so this is not an online-softmax kernel: a workgroup owns 32 queries of one head,
// computes ALL their scores into LDS (32 rows x T f32), runs the row softmax there exactly as k_p_soft_max does, overwrites
// each row in place with its f16 probabilities and multiplies by V.  Both products use v_mfma_f32_32x32x16_f16 with the
// operand roles and k order of k_gemm_f16 (A = queries, B = keys / value channels, 16 k per instruction, ascending), so
// the result is bit-identical to the three-launch path (tests/test_prompt_plan_gpu.py).
//
//   grid = (ceil(N / 32) * H), 256 threads.  Dynamic LDS = 32 x (Tp * 4 + 16) bytes, Tp = (n_past + N) rounded up to 64.
//   S phase: wave w takes key tiles w, w+4, ... of 32 keys (8 MFMAs for D = 128), K fragments straight from global / L2,
//            the next tile's fragments requested before the current tile's MFMAs.
//   softmax: wave w takes rows 8w .. 8w+7.
//   V.P    : wave w takes value channels 32w .. 32w+31 (D / 32 waves), V^T fragments from global, P fragments from LDS.
#pragma once
#include "gemm_f16.h"
#include "mmq.h"

struct PAttnArgs {
    const float *q;      // [N][E] f32, RoPE applied
    const __half *mem_k; // this layer: [C][Egqa]
    const __half *mem_v; // this layer: [Egqa][C]
    float *out;          // [N][E] f32, merged heads
    int N, E, Egqa, H, r, n_past;
    int64_t C;
    float scale;
    int row_bytes;       // LDS bytes per score row
    const uint16_t *exp_tab;  // != nullptr: ggml's table_exp_f16 (65536 halves: f16(expf(f32(h))) for every f16 bit pattern h,
                         // filled by the host's expf): the softmax looks its exponentials up exactly as ggml does
    const float *rope;   // != nullptr: q is the raw wq product; RoPE (cos, sin per pair: 128 floats per token, k_rope_table) is
                         // applied while its fragments are loaded — k_p_qkv_post's operations on the same values
    int64_t q_part;      // != 0: ... and q is the first partial of a K-split GEMM, the second lies q_part floats on
    _Float16 *x16;       // != nullptr: the output goes out as wo's GEMM operand instead — every 32-channel block re-quantized to Q8
                         // and written as f16(d * q) in the GEMM's k order (what k_p_quant4 makes of `out`); `out` is not written
    int f16d;            // ... with the block scale rounded to f16 first (weight types whose vec_dot_type is Q8_0)
    long long *ts;       // INSTR build (option "timeline"): 8 x int64 per workgroup, see tests/tools/pattn_timeline.py
};

#define PATTN_Q 32

template <int D, bool INSTR = false>
__global__ void __launch_bounds__(256, 2) k_p_attn(const PAttnArgs a) {
    long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    if constexpr (INSTR) t0 = (long long)wall_clock64();
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int KS = D / 16;      // MFMA k steps of a K.Q tile
    constexpr int NWV = D / 32;     // waves that take part in V.P
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int ntile = (a.N + PATTN_Q - 1) / PATTN_Q;
    const int h = (int)blockIdx.x % a.H, qt = ntile - 1 - (int)blockIdx.x / a.H;  // the longest rows first
    const int hk = h / a.r;
    const int q0 = qt * PATTN_Q;
    const int Ttot = a.n_past + a.N;                        // keys written so far
    const int T_hi = min(a.n_past + q0 + PATTN_Q, Ttot);    // keys 0 .. T_hi - 1 are visible to some query of the tile
    const int nkt = (T_hi + 31) >> 5;
    const int rb = a.row_bytes;

    // ---- Q fragments (A operand): row fr of the tile, 8 channels per k step and lane half
    f16x8 qa[KS];
    {
        const int qn = min(q0 + fr, a.N - 1);
        const float *qp = a.q + (int64_t)qn * a.E + h * D + fh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            f32x4_u x0 = *(const f32x4_u *)(qp + ks * 16), x1 = *(const f32x4_u *)(qp + ks * 16 + 4);
            if (a.rope) {  // uniform
                if (a.q_part) {
                    const f32x4_u y0 = *(const f32x4_u *)(qp + a.q_part + ks * 16), y1 = *(const f32x4_u *)(qp + a.q_part + ks * 16 + 4);
                    x0 = x0 + y0;
                    x1 = x1 + y1;
                }
                // dims fh*8 + ks*16 + 0..7 of the head = pairs fh*4 + ks*8 + 0..3: the table offset equals the dim offset
                const float *tp = a.rope + (int64_t)qn * 128 + fh * 8 + ks * 16;
                const f32x4_u c0 = *(const f32x4_u *)tp, c1 = *(const f32x4_u *)(tp + 4);
                f32x4_u o0, o1;
                o0[0] = x0[0] * c0[0] - x0[1] * c0[1];
                o0[1] = x0[0] * c0[1] + x0[1] * c0[0];
                o0[2] = x0[2] * c0[2] - x0[3] * c0[3];
                o0[3] = x0[2] * c0[3] + x0[3] * c0[2];
                o1[0] = x1[0] * c1[0] - x1[1] * c1[1];
                o1[1] = x1[0] * c1[1] + x1[1] * c1[0];
                o1[2] = x1[2] * c1[2] - x1[3] * c1[3];
                o1[3] = x1[2] * c1[3] + x1[3] * c1[2];
                x0 = o0;
                x1 = o1;
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                qa[ks][e] = (_Float16)x0[e];
                qa[ks][4 + e] = (_Float16)x1[e];
            }
        }
    }
    if constexpr (INSTR) t1 = (long long)wall_clock64();
    // ---- S phase
    auto load_k = [&](int kt, f16x8 (&kb)[KS]) {
        const int64_t t = min((int64_t)kt * 32 + fr, a.C - 1);
        const __half *kp = a.mem_k + t * a.Egqa + hk * D + fh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) kb[ks] = *(const f16x8 *)(kp + ks * 16);
    };
    {
        // K fragments of up to three of the wave's key tiles in flight (96 VGPRs for D = 128; 2 waves per SIMD leave 256):
        // with one tile ahead the phase ran at one L2 / HBM round trip per tile (in-kernel timeline, 2.1 us per tile)
        constexpr int KR = 3;
        f16x8 kb[KR][KS];
#pragma unroll
        for (int j = 0; j < KR - 1; j++)
            if (wave + 4 * j < nkt) load_k(wave + 4 * j, kb[j]);
        for (int kt0 = wave; kt0 < nkt; kt0 += 4 * KR) {
#pragma unroll
            for (int j = 0; j < KR; j++) {
                const int kt = kt0 + 4 * j;
                if (kt < nkt) {  // wave-uniform
                    if (kt + 4 * (KR - 1) < nkt) load_k(kt + 4 * (KR - 1), kb[(j + KR - 1) % KR]);
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[ks], kb[j][ks], acc, 0, 0, 0);
                    float *sp = (float *)(lds + 4 * fh * rb) + kt * 32 + fr;
#pragma unroll
                    for (int r = 0; r < 16; r++) *(float *)((char *)sp + ((r & 3) + 8 * (r >> 2)) * rb) = acc[r];
                }
            }
        }
    }
    __syncthreads();
    if constexpr (INSTR) t2 = (long long)wall_clock64();
    // ---- softmax: the operations of k_p_soft_max, row by row; the row is then overwritten with its f16 probabilities
    const int npad = ((T_hi + 15) >> 4) << 4;  // V.P reads whole 16-key chunks: zeros behind the last visible key
    {
        // The wave's 8 rows side by side: each row's three passes are chains of LDS read -> exp -> LDS write that ran at one
        // latency per step when the rows went one after the other (20 of the kernel's 35 us); per element and per row the
        // operations and their order are unchanged.
        constexpr int RW = 8;
        const int row0 = wave * RW;
        const int nrow = min(RW, a.N - q0 - row0);  // rows of this wave that exist (ragged last tile), wave-uniform, may be <= 0
        const int lim0 = a.n_past + q0 + row0;      // row rr sees keys 0 .. lim0 + rr
        const int lim_hi = lim0 + nrow - 1;
        float mx[RW];
#pragma unroll
        for (int rr = 0; rr < RW; rr++) mx[rr] = -INFINITY;
        for (int i = lane; i <= lim_hi; i += 64) {
#pragma unroll
            for (int rr = 0; rr < RW; rr++)
                if (rr < nrow && i <= lim0 + rr) mx[rr] = fmaxf(mx[rr], ((const float *)(lds + (row0 + rr) * rb))[i] * a.scale);
        }
#pragma unroll
        for (int rr = 0; rr < RW; rr++) mx[rr] = wave_max_f32(mx[rr]);
        double sum[RW];
#pragma unroll
        for (int rr = 0; rr < RW; rr++) sum[rr] = 0.0;
        for (int i = lane; i <= lim_hi; i += 64) {
#pragma unroll
            for (int rr = 0; rr < RW; rr++)
                if (rr < nrow && i <= lim0 + rr) {
                    float *p = (float *)(lds + (row0 + rr) * rb);
                    const float t = p[i] * a.scale - mx[rr];
                    float e;
                    if (a.exp_tab)  // uniform: a 2-byte gather from a 128 KB table that lives in L1 / L2 instead of ~20 VALU operations
                        e = __half2float(__ushort_as_half(a.exp_tab[__half_as_ushort(__float2half_rn(t))]));
                    else
                        e = round_f16(expf(round_f16(t)));
                    sum[rr] += (double)e;
                    p[i] = e;
                }
        }
        float inv[RW];
#pragma unroll
        for (int rr = 0; rr < RW; rr++) {
            sum[rr] = wave_sum_f64(sum[rr]);
            inv[rr] = (float)(1.0 / sum[rr]);
        }
        // each row overwritten in place with its f16 probabilities: f16 element i lands on f32 element i / 2, which this wave
        // read in an earlier (or this) iteration — LDS operations of a wave execute in order
        for (int i0 = 0; i0 < npad; i0 += 64) {
            const int i = i0 + lane;
#pragma unroll
            for (int rr = 0; rr < RW; rr++)
                if (rr < nrow) {
                    float *p = (float *)(lds + (row0 + rr) * rb);
                    const float e = i <= lim0 + rr ? p[i] : 0.0f;
                    if (i < npad) ((_Float16 *)p)[i] = (_Float16)(e * inv[rr]);
                }
        }
    }
    __syncthreads();
    if constexpr (INSTR) t3 = (long long)wall_clock64();
    // ---- V.P
    if (wave < NWV) {
        const int nch = npad >> 4;
        const int d0 = wave * 32;
        const __half *vp = a.mem_v + ((int64_t)hk * D + d0 + fr) * a.C + fh * 8;
        const char *pa = lds + fr * rb + fh * 16;
        auto load_v = [&](int c) {
            const int valid = Ttot - (c * 16 + fh * 8);  // the cache beyond the last written key may hold anything,
            u32x4 v = {0, 0, 0, 0};                      // and a row ends at C (C % 8 == 0: a group of 8 is inside or outside)
            if (valid > 0) v = *(const u32x4 *)(vp + c * 16);
            if (valid < 8) v = gf16_mask_tail(v, valid < 0 ? 0 : valid);
            return __builtin_bit_cast(f16x8, v);
        };
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
        constexpr int PF = 8;  // chunks in flight
        f16x8 vb[PF];
#pragma unroll
        for (int k = 0; k < PF; k++)
            if (k < nch) vb[k] = load_v(k);
        for (int c0 = 0; c0 < nch; c0 += PF) {
#pragma unroll
            for (int k = 0; k < PF; k++) {
                if (c0 + k < nch) {
                    const f16x8 pf = *(const f16x8 *)(pa + (c0 + k) * 32);
                    const f16x8 v = vb[k];
                    if (c0 + k + PF < nch) vb[k] = load_v(c0 + k + PF);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf, v, acc, 0, 0, 0);
                }
            }
        }
        if (a.x16) {
            // the wave's 32 channels are one Q8 block of the merged row: amax over the 32 lanes of a half, k_p_quant4's formula
            _Float16 *xp = a.x16 + (int64_t)(q0 + 4 * fh) * a.E + h * D + d0 + mmq_kperm_inv(fr);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2);
                const float v = acc[r];
                float amax = fabsf(v);
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
                const float d = amax / 127.0f;
                const float id = d != 0.0f ? 1.0f / d : 0.0f;
                const float dq = a.f16d ? round_f16(d) : d;
                const int qv = (int)roundf(v * id);
                float rq = dq * (float)qv;
                rq = fminf(fmaxf(rq, -65504.0f), 65504.0f);
                if (q0 + 4 * fh + row < a.N) xp[(int64_t)row * a.E] = (_Float16)rq;
            }
        } else {
            float *op = a.out + (int64_t)(q0 + 4 * fh) * a.E + h * D + d0 + fr;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2);
                if (q0 + 4 * fh + row < a.N) op[(int64_t)row * a.E] = acc[r];
            }
        }
    }
    if constexpr (INSTR) {
        t4 = (long long)wall_clock64();
        if (a.ts && tid == 0) {
            long long *o = a.ts + (size_t)blockIdx.x * 8;
            o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4; o[5] = T_hi; o[6] = qt; o[7] = h;
        }
    }
}
