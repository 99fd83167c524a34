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
//   grid = (ceil(N / 32) * H), 256 threads.  Dynamic LDS = 32 x (Tp * 4 + 16) bytes, Tp = (n_past + N) rounded up to 64,
//   (the staged Q tile, 32 x (2 D + 16) bytes, borrows the score rows before the S phase).
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
    const float *rope;   // != nullptr: q is the raw wq product; RoPE (cos, sin per pair: 128 floats per token, k_rope_table) is
                         // applied while its fragments are loaded — k_p_qkv_post's operations on the same values
    int64_t q_part;      // != 0: ... and q is the first partial of a K-split GEMM, the second lies q_part floats on
    _Float16 *x16;       // != nullptr: the output goes out as wo's GEMM operand instead — every 32-channel block re-quantized to Q8
                         // and written as f16(d * q) in the GEMM's k order (what k_p_quant4 makes of `out`); `out` is not written
    int f16d;            // ... with the block scale rounded to f16 first (weight types whose vec_dot_type is Q8_0)
    int r1;              // query-tile ranks (blockIdx / H) dealt longest-first; the ranks behind them go shortest-first (see the kernel)
    long long *ts;       // INSTR build (option "timeline"): 8 x int64 per workgroup, see tests/tools/pattn_timeline.py
};

#define PATTN_Q 32

// expf for the softmax's arguments: x = f16(score - row maximum), so x <= 0 (or NaN for a masked column's garbage, which is selected
// away).  The operations of the device library's expf (ph = x * log2e rounded; pl = its error by two fmas with log2e's head and tail;
// e = rint(ph); 2^((ph - e) + pl) by v_exp_f32; ldexp by e) with ONE clamp in place of its two range checks (x < -103.28 -> 0,
// x > 88.72 -> inf: two compares, two selects and their wait states per element): the second cannot fire; below -104 (f16's -inf
// included: inf - inf inside the sequence would make a NaN of it) the argument is held at -104, whose result (7e-46) rounds to the
// same f16 zero the library's branch returns.  A NaN stays a NaN.  Same f16 bits as f16(expf(x)) for every f16 x <= 0 and every
// NaN: tests/test_prompt_plan_gpu.py checks all 2^15 + 1 of them on the device (ggml_hip_debug_exp_le0).
__device__ __forceinline__ float exp_le0(const float x0) {
    const float c = 0x1.715476p+0f, cc = 0x1.4ae0bep-26f;  // log2(e): head and tail
    const float x = x0 < -104.0f ? -104.0f : x0;
    const float ph = x * c;
    float pl = __builtin_fmaf(x, c, -ph);
    pl = __builtin_fmaf(x, cc, pl);
    const float e = __builtin_rintf(ph);
    const float a_ = (ph - e) + pl;
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(a_), (int)e);
}

// QR = queries per workgroup: 32, or 16 for rows too long for 32 score rows in LDS (T up to ~2300 keys: the last chunks of a
// 2048-token context; the MFMA tiles stay 32 rows high, their upper half computes on duplicated queries and is dropped)
template <int D, bool INSTR = false, int QR = PATTN_Q>
__global__ void __launch_bounds__(256, 2) k_p_attn(const PAttnArgs a) {
    long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    if constexpr (INSTR) t0 = (long long)wall_clock64();
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int KS = D / 16;      // MFMA k steps of a K.Q tile
    constexpr int NWV = D / 32;     // waves that take part in V.P
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int ntile = (a.N + QR - 1) / QR;
    // Which query tile: the longest rows first — but only for the workgroups that fill the CUs' FIRST slots (ranks < a.r1).  Two
    // workgroups share a CU (66 KB of scores each at 512 keys) and the second slots fill in the same CU order, so the ranks behind
    // them go SHORTEST first: the CU that got tile 15 gets tile 0, the one with tile 14 tile 1, ... — every CU holds the same number
    // of keys (544 at 512 tokens instead of 768 on the first CUs and 320 on the last ones; the softmax is VALU-bound per CU).
    const int h = (int)blockIdx.x % a.H, rank = (int)blockIdx.x / a.H;
    const int qt = rank < a.r1 ? ntile - 1 - rank : rank - a.r1;
    const int hk = h / a.r;
    const int q0 = qt * QR;
    const int Ttot = a.n_past + a.N;                        // keys written so far
    const int T_hi = min(a.n_past + q0 + QR, Ttot);    // keys 0 .. T_hi - 1 are visible to some query of the tile
    const int nkt = (T_hi + 31) >> 5;
    const int rb = a.row_bytes;

    auto load_k = [&](int kt, f16x8 (&kb)[KS]) {
        const int64_t t = min((int64_t)kt * 32 + fr, a.C - 1);
        const __half *kp = a.mem_k + t * a.Egqa + hk * D + fh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) kb[ks] = *(const f16x8 *)(kp + ks * 16);
    };
    // the wave's first K tiles are requested before anything else: they depend on nothing, and the Q staging below is two
    // barriers and a cold round trip long
    constexpr int KR = 3;
    f16x8 kb[KR][KS];
#pragma unroll
    for (int j = 0; j < KR - 1; j++)
        if (wave + 4 * j < nkt) load_k(wave + 4 * j, kb[j]);
    // ---- Q tile -> LDS once per workgroup (f16, rotated if a.rope), then every wave's A fragments from there.  All four
    // waves need the whole tile: fetched per wave it is 64 KB of f32 (128 KB with the RoPE table) through an L1 that delivers
    // ~17 B per clock — the in-kernel timeline showed 10.8 us of Q load for every workgroup of the prompt plan.
    constexpr int QROW = D * 2 + 16;  // bytes per staged row (+16: fragment reads of consecutive rows spread over the banks)
    char *s_q = lds;  // the score rows are not in use yet (32 x QROW <= 32 x row_bytes for every T the launcher accepts)
    for (int idx = tid; idx < QR * (D / 16); idx += 256) {
        const int row = idx / (D / 16), c = idx % (D / 16);
        const int qn = min(q0 + row, a.N - 1);
        const float *qp = a.q + (int64_t)qn * a.E + h * D + c * 16;
        f32x4_u x[4];
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = *(const f32x4_u *)(qp + 4 * k);
        if (a.rope) {  // uniform
            if (a.q_part) {
#pragma unroll
                for (int k = 0; k < 4; k++) x[k] = x[k] + *(const f32x4_u *)(qp + a.q_part + 4 * k);
            }
            // dims c*16 + 0..15 of the head = pairs c*8 + 0..7: the table offset (2 floats per pair) equals the dim offset
            const float *tp = a.rope + (int64_t)qn * 128 + c * 16;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const f32x4_u cs = *(const f32x4_u *)(tp + 4 * k);
                f32x4_u o;
                o[0] = x[k][0] * cs[0] - x[k][1] * cs[1];
                o[1] = x[k][0] * cs[1] + x[k][1] * cs[0];
                o[2] = x[k][2] * cs[2] - x[k][3] * cs[3];
                o[3] = x[k][2] * cs[3] + x[k][3] * cs[2];
                x[k] = o;
            }
        }
        f16x8 h0, h1;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            h0[e] = (_Float16)x[0][e];
            h0[4 + e] = (_Float16)x[1][e];
            h1[e] = (_Float16)x[2][e];
            h1[4 + e] = (_Float16)x[3][e];
        }
        *(f16x8 *)(s_q + row * QROW + c * 32) = h0;
        *(f16x8 *)(s_q + row * QROW + c * 32 + 16) = h1;
    }
    __syncthreads();
    f16x8 qa[KS];  // A operand: row fr of the tile, 8 channels per k step and lane half
#pragma unroll
    for (int ks = 0; ks < KS; ks++) qa[ks] = *(const f16x8 *)(s_q + (fr & (QR - 1)) * QROW + (fh * 8 + ks * 16) * 2);
    __syncthreads();  // the fragments are in registers: the S phase may overwrite the staging area
    if constexpr (INSTR) t1 = (long long)wall_clock64();
    // ---- S phase
    {
        // K fragments of up to three of the wave's key tiles in flight (96 VGPRs for D = 128; 2 waves per SIMD leave 256):
        // with one tile ahead the phase ran at one L2 / HBM round trip per tile (in-kernel timeline, 2.1 us per tile)
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
                    for (int r = 0; r < 16; r++)  // row (r & 3) + 8 (r >> 2) + 4 fh: below 16 exactly for r < 8
                        if (QR == 32 || r < 8) *(float *)((char *)sp + ((r & 3) + 8 * (r >> 2)) * rb) = acc[r];
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
        constexpr int RW = QR / 4;  // rows per wave
        const int row0 = wave * RW;
        const int nrow = min(RW, a.N - q0 - row0);  // rows of this wave that exist (ragged last tile), wave-uniform, may be <= 0
        const int lim0 = a.n_past + q0 + row0;      // row rr sees keys 0 .. lim0 + rr
        const int lim_hi = lim0 + nrow - 1;
        // rows that do not exist (ragged last tile) alias the wave's first row for their reads and never write
        const char *rp[RW];
#pragma unroll
        for (int rr = 0; rr < RW; rr++) rp[rr] = lds + (row0 + (rr < nrow ? rr : 0)) * rb;
        // every pass reads its eight rows UNCONDITIONALLY (index i < Tp is inside every row) and predicates the update: with
        // the loads under per-row branches the compiler waited for each LDS read before issuing the next row's
        float mx[RW];
#pragma unroll
        for (int rr = 0; rr < RW; rr++) mx[rr] = -INFINITY;
        for (int i = lane; i <= lim_hi; i += 64) {
            float x[RW];
#pragma unroll
            for (int rr = 0; rr < RW; rr++) x[rr] = ((const float *)rp[rr])[i];
#pragma unroll
            for (int rr = 0; rr < RW; rr++) {
                const float v = x[rr] * a.scale;
                mx[rr] = (rr < nrow && i <= lim0 + rr) ? fmaxf(mx[rr], v) : mx[rr];
            }
        }
#pragma unroll
        for (int rr = 0; rr < RW; rr++) mx[rr] = wave_max_f32(mx[rr]);
        double sum[RW];
#pragma unroll
        for (int rr = 0; rr < RW; rr++) sum[rr] = 0.0;
        if (nrow == RW) {
            // All rows exist (every tile but a ragged last one): no branch in the body.  A masked column's value (garbage, possibly
            // NaN: a score against a key behind the row's last) is SELECTED away from the sum and may be stored — the third pass
            // replaces masked columns by zero without reading them.  With a branch per row the compiler ran the eight dependent
            // chains one after the other (softmax 12.9 of tile 15's 24.5 us: r06_pattn_timeline.txt).
            for (int i = lane; i <= lim_hi; i += 64) {
                float x[RW], e[RW];
#pragma unroll
                for (int rr = 0; rr < RW; rr++) x[rr] = ((const float *)rp[rr])[i];
#pragma unroll
                for (int rr = 0; rr < RW; rr++) e[rr] = round_f16(exp_le0(round_f16(x[rr] * a.scale - mx[rr])));
#pragma unroll
                for (int rr = 0; rr < RW; rr++) {
                    sum[rr] += (double)(i <= lim0 + rr ? e[rr] : 0.0f);
                    ((float *)rp[rr])[i] = e[rr];
                }
            }
        } else {
            for (int i = lane; i <= lim_hi; i += 64) {
                float x[RW], e[RW];
#pragma unroll
                for (int rr = 0; rr < RW; rr++) x[rr] = ((const float *)rp[rr])[i];
#pragma unroll
                for (int rr = 0; rr < RW; rr++) e[rr] = round_f16(exp_le0(round_f16(x[rr] * a.scale - mx[rr])));
#pragma unroll
                for (int rr = 0; rr < RW; rr++)
                    if (rr < nrow && i <= lim0 + rr) {
                        sum[rr] += (double)e[rr];
                        ((float *)(lds + (row0 + rr) * rb))[i] = e[rr];
                    }
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
        if (nrow == RW) {  // all rows exist: one lane predicate around the eight stores instead of a branch per row
            for (int i0 = 0; i0 < npad; i0 += 64) {
                const int i = i0 + lane;
                float x[RW];
#pragma unroll
                for (int rr = 0; rr < RW; rr++) x[rr] = ((const float *)rp[rr])[min(i, npad - 1)];
                _Float16 pr[RW];
#pragma unroll
                for (int rr = 0; rr < RW; rr++) pr[rr] = (_Float16)((i <= lim0 + rr ? x[rr] : 0.0f) * inv[rr]);
                if (i < npad) {
#pragma unroll
                    for (int rr = 0; rr < RW; rr++) ((_Float16 *)rp[rr])[i] = pr[rr];
                }
            }
        } else {
            for (int i0 = 0; i0 < npad; i0 += 64) {
                const int i = i0 + lane;
                float x[RW];
#pragma unroll
                for (int rr = 0; rr < RW; rr++) x[rr] = ((const float *)rp[rr])[min(i, npad - 1)];
#pragma unroll
                for (int rr = 0; rr < RW; rr++)
                    if (rr < nrow && i < npad) {
                        const float e = i <= lim0 + rr ? x[rr] : 0.0f;
                        ((_Float16 *)(lds + (row0 + rr) * rb))[i] = (_Float16)(e * inv[rr]);
                    }
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
        const char *pa = lds + (fr & (QR - 1)) * rb + fh * 16;
        constexpr int NR = QR / 2;  // accumulator registers that hold real query rows (row (r & 3) + 8 (r >> 2) + 4 fh < QR)
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
            for (int r = 0; r < NR; r++) {
                const int row = (r & 3) + 8 * (r >> 2);
                const float v = acc[r];
                // max over the 32 lanes of this half: DPP inside each row of 16, then the two rows' results through scalar
                // registers (a __shfl_xor is an LDS round trip: five of them per row made this epilogue cost 15 us per launch)
                float amax = g16_max_f32(fabsf(v));
                {
                    const int ai = __builtin_bit_cast(int, amax);
                    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ai, 0));
                    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ai, 16));
                    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ai, 32));
                    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ai, 48));
                    amax = fh ? fmaxf(r2, r3) : fmaxf(r0, r1);
                }
                const float d = amax / 127.0f;
                const float id = act_id(amax, d, aq_scalar());
                const float dq = a.f16d ? round_f16(d) : d;
                const int qv = act_q(v * id, aq_scalar());
                float rq = dq * (float)qv;
                rq = fminf(fmaxf(rq, -65504.0f), 65504.0f);
                if (q0 + 4 * fh + row < a.N) xp[(int64_t)row * a.E] = (_Float16)rq;
            }
        } else {
            float *op = a.out + (int64_t)(q0 + 4 * fh) * a.E + h * D + d0 + fr;
#pragma unroll
            for (int r = 0; r < NR; r++) {
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
