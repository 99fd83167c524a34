// decode_big.h — the decode mat-vec as ONE wave of big workgroups: grid = #CUs x 1024 threads, the workgroups walk the
// output rows together (unit u -> workgroup (u / 16) mod G, so at any moment the chip streams one contiguous window of
// the matrix), the activation's loads are the first memory operations of the kernel, the weight stream is prefetched
// into a register ring (2 steps before the activation is staged, the rest after), and the activation's norm /
// re-quantization is done by each workgroup while those loads are in flight.  Same arithmetic as k_mmvq_dec
// (decode.h) / k_mmvq (mmvq.h); same epilogues.
//
// Why (measured on MI355X, profiles/r01_run8 -> r01_run33): with 256-thread workgroups a LLaMA-7B mat-vec needs
// 512..4000 workgroups, each of which re-stages x (4.6 KB from L2, ~1 us of latency before its first dot) and has one
// K-step of weights in flight; the E x E mat-vec ran at 1.9 TB/s, wq|wk|wv at 3.0 TB/s, and the separate
// rms_norm+quantize / quantize launches in front of them cost ~6 us each (a 1-workgroup latency chain plus a
// kernel boundary).  Here:
//   * 256 workgroups = one per CU, resident at once: no dispatch tail, x staged 256 times instead of 4000;
//   * 16 waves x up to 8 steps x 16 B x rows-per-step in flight per lane: the HBM pipe is busy from the start;
//   * staging x = rms_norm (f64 sum of squares) -> weight -> Q8 blocks in LDS overlaps the weight prefetch and
//     removes 3 launches per layer (8 -> 5).
// The pitfalls that made the first version slower than what it replaced are listed in DESIGN.md section 4 (in-order
// load return, vector loads of kernarg arrays, uncountable loads in flight, epilogue loads, x queued behind the
// prefetch, 64-bit index math).
// A "unit" is what one wave reduces together: a pair of adjacent rows for wq|wk|wv (RoPE rotates the pair),
// row m of w1 and of w3 for the gate, one row otherwise.  A "step" is one 64-block column of a unit.
#pragma once
#include "decode.h"

template <int QT, int NR>
struct BigStep {
    u32x4 q[NR];
    u32x4 p[QT == QT_Q8_0 ? NR : 1];
    uint32_t h[(QT == QT_Q5_0 || QT == QT_Q5_1) ? NR : 1];
    // the f16 scales travel as zero-extended 32-bit words, one VGPR each: as `__half dw[2]` the compiler keeps the two halves of
    // a step in ONE register and assembles it (v_perm_b32) right behind the loads — behind an s_waitcnt vmcnt(0) that drained the
    // whole ring at every step of the two-row kernels (w1|w3, wq|wk|wv): tests/tools/disasm.py, round 6
    uint32_t dw[NR];
    uint32_t mw[(QT == QT_Q4_1 || QT == QT_Q5_1) ? NR : 1];
};
__device__ __forceinline__ float big_h2f(uint32_t raw) { return __half2float(__ushort_as_half((unsigned short)raw)); }

// Ring depth = weight steps in flight per lane.  NOT "as many as the registers hold": a CU accepts only so many
// outstanding requests, and a wave whose next load is not accepted sits in ISSUE — it reaches neither the staging
// barrier nor its dots.  Round 2 settled on 5 steps of one row / 3 of two rows — with waits that, as the disassembly showed in
// round 6, drained the whole ring at the head of every pass (a 16-bit scale load that hipcc packed behind an s_waitcnt, a
// refill under a branch, a store that might be pending: see BigStep, `issue`, big_stage_x), so the ring was never that deep in
// flight.  With counted waits (PF steps really in flight at every wait) the sweep over -DBIG_PF1 / -DBIG_PF2 builds on one box
// (gpurun_out/r6/run9, all mat-vec launches of a 7B token, ms): 5|3 1.212, 4|3 1.195, 4|2 1.213, 3|2 1.207, 3|3 1.188,
// 6|3 1.211, 6|4 1.237, 2|2 1.232 — w2 (6 steps per wave at 7B) 7.7 us per launch at 5, 7.1 at 4, 6.9 at 3.
#ifndef BIG_PF1
#define BIG_PF1 3  // steps of 1 row  (1 KB of Q4/Q5 quants each)
#endif
#ifndef BIG_PF2
#define BIG_PF2 3  // steps of 2 rows
#endif
template <int QT>
__device__ __forceinline__ constexpr int big_pf(int NR) {
    // at most ~64 VGPRs of weight data in flight per lane (Q8_0 and the 5-bit types carry more per row)
    const int per = NR * (4 + (QT == QT_Q8_0 ? 4 : 0) + ((QT == QT_Q5_0 || QT == QT_Q5_1) ? 1 : 0) + 1 +
                          ((QT == QT_Q4_1 || QT == QT_Q5_1) ? 1 : 0));
    const int fit = 64 / per >= 2 ? 64 / per : 2;
    int pf = NR == 1 ? BIG_PF1 : NR == 2 ? BIG_PF2 : 2;
    if (QT == QT_Q8_0) pf = (pf + 1) / 2;  // two 16-byte planes per row step
    return pf < 2 ? 2 : pf > fit ? fit : pf;
}

#ifndef BIG_T
#define BIG_T 1024  // threads per workgroup of k_mmvq_big; 512 (-DBIG_T=512) starts faster (wo: barrier at 1.5 us instead
                    // of 2.3) but streams slower with half the waves (w1|w3 10.3 us vs 9.2, lm_head 18.6 vs 12.7): 600 vs 612 tok/s
#endif
#define BIG_W (BIG_T / 64)
// k_mmvq_big itself runs with any multiple of 64 threads up to BIG_T (blockDim.x): the launcher picks the number of
// waves per workgroup that deals the launch's units most evenly (launch_big).  12 waves x 256 workgroups give every
// wave exactly 2 of the 6144 row pairs of a 7B wq|wk|wv; with 16 waves half of them get 2 and half 1, and the
// launch lasts as long as the workgroups with the 2s.

struct BigArgs {
    DecMmvqArgs d;
    float *y_out;  // XSRC_NORM: optional f32 copy of the normed row (final norm -> OutputRequest.embeddings)
    long long *ts;  // optional timeline slot (ggml_hip_set_option("timeline", n)): 8 x int64 per sampled workgroup
    int ts_wgs;     // workgroups that record: 0, G/n, 2G/n, ... (n = 4 for "timeline" = 1, else the option's value)
    const float *rope;  // EPI_QKV: (cos, sin) of this token's RoPE angle per pair of a head, from k_rope_table
    int probe;          // measurement only (ggml_hip_set_option("probe", n), tests/tools/launch_probe.py): 1 = return before
                        // the first weight request, 2 = return once x is staged and the ring requested, 3 = no epilogue stores, 4 = every
                        // other step's dots skipped, 5 = no wave reductions; 0 = normal
    // EPI_QKV inside k_qkv_attn (kernels/decode_fused.h): the epilogue also PUBLISHES every row pair to the attention workgroups
    // of the same launch, as one 8-byte {tag = epoch, two f16} granule per pair (index = the pair's index over wq|wk|wv); Q is
    // not stored as f32 then (nothing else reads it).  nullptr = plain launch.
    unsigned long long *gran;
    const unsigned *epoch;  // device word, bumped once per token by k_rope_table: this token's tag
    int wdeal;              // waves of a workgroup that take units (0 = all of blockDim); the rest only help staging
    // 256 bytes that the DUMMY ring steps read (a wave's slots past its last real step): one line for the whole chip, fetched with
    // plain loads, so it sits in every CU's L1 — a dummy step costs its issue slots and nothing else (see `issue`).  nullptr = the
    // first line of the matrix's scales.
    const void *hot;
    // EPI_QKV inside k_qkv_attn, XCD-affine dealing (aff_hpl > 0): the row pairs of head h's Q / K / V go to the mat-vec workgroups
    // whose blockIdx mod 8 equals h mod 8 — the XCD head h's attention workgroup (blockIdx h) runs on — and travel through that
    // XCD's L2 (gran_store_l2).  aff_hpl = heads per XCD (n_head / 8), aff_shift = log2(D / 2).  MHA only (wk, wv as tall as wq).
    int aff_hpl, aff_shift;
};
__device__ __forceinline__ long long big_now() { return (long long)wall_clock64(); }  // 100 MHz, chip-wide

// The activation's global loads, issued as the FIRST memory operations of the kernel: a wave's loads return in
// order, so anything issued after the weight prefetch would only become usable after the whole prefetch landed
// (measured: norm staging behind the prefetch made the kernels additive, 16.6 us for wq|wk|wv instead of ~8).
template <int XSRC>
struct BigX;
template <>
struct BigX<XSRC_Q8> {
    i32x4 lo, hi;
    float d;
    int sum;
    __device__ __forceinline__ void load(const BigArgs &a, int nb, int tid, int T) {
        const int i = tid < nb ? tid : 0;  // nb <= T (checked by the launcher)
        lo = a.d.x.lo[i];
        hi = a.d.x.hi[i];
        d = a.d.x.d[i];
        sum = a.d.x.sum[i];
    }
};
template <>
struct BigX<XSRC_F32> {
    static constexpr int MAXIT = 6;  // rows up to 24 * T wide (24576 at 1024 threads; checked by the launcher)
    f32x4 v[MAXIT];
    __device__ __forceinline__ void load(const BigArgs &a, int nb, int tid, int T) {
        const int n4 = nb * 8;
#pragma unroll
        for (int it = 0; it < MAXIT; it++) {
            const int i4 = it * T + tid;
            v[it] = ((const f32x4 *)a.d.xf)[i4 < n4 ? i4 : 0];
        }
    }
};
template <>
struct BigX<XSRC_NORM> {
    // The norm is staged by ALL 16 waves of a 1024-thread workgroup (the launcher gives every XSRC_NORM launch 1024 threads; waves
    // beyond BigArgs::wdeal take no units), one f32x4 of the row per thread at the 7B width.  Round 2 staged on 8 waves ("the fixed
    // per-thread cost is paid 512 times instead of 1024"): the in-kernel timeline then showed x staged 3.3-4.2 us after entry
    // against 1.4-1.7 for the plain re-quantization of w2's input — two dependent passes of ~115 VALU on two waves per SIMD are a
    // LATENCY chain (DPP reductions, two IEEE divisions per block), not an issue-bound one, and with the next launch's first rows
    // warm in L2 (NextWarm, decode_fused.h) the staging, not the first weight byte, is what the first dot waits for.
    static constexpr int NT = 1024, MAXIT = 2;  // rows up to 8192 wide
    f32x4 v[MAXIT], w[MAXIT];
    __device__ __forceinline__ void load(const BigArgs &a, int nb, int tid, int T) {
        const int n4 = nb * 8;
#pragma unroll
        for (int it = 0; it < MAXIT; it++) {
            const int i4 = it * NT + tid;
            const int ic = i4 < n4 ? i4 : 0;
            v[it] = ((const f32x4 *)a.d.xf)[ic];
            w[it] = ((const f32x4 *)a.d.xw)[ic];
        }
    }
};

// registers -> LDS as padded planar Q8 (nbp = nbl*64 blocks; blocks >= nb are zero so tail steps contribute 0)
template <bool F16_D, int XSRC, bool YOUT = false>
__device__ __forceinline__ void big_stage_x(const BigArgs &a, const BigX<XSRC> &xr, int nb, int nbp, int tid, int T,
                                            i32x4 *s_lo, i32x4 *s_hi, float *s_d, int *s_sum, double *s_part, f32x4 *y_keep = nullptr) {
    const DecMmvqArgs &d = a.d;
    (void)s_part;
    for (int i = nb + tid; i < nbp; i += T) {
        s_lo[i] = i32x4{0, 0, 0, 0};
        s_hi[i] = i32x4{0, 0, 0, 0};
        s_d[i] = 0.0f;
        s_sum[i] = 0;
    }
    const int n4 = nb * 8;
    if constexpr (XSRC == XSRC_Q8) {
        if (tid < nb) {
            s_lo[tid] = xr.lo;
            s_hi[tid] = xr.hi;
            s_d[tid] = xr.d;
            s_sum[tid] = xr.sum;
        }
    } else if constexpr (XSRC == XSRC_F32) {
#pragma unroll
        for (int it = 0; it < BigX<XSRC_F32>::MAXIT; it++) {
            const int i4 = it * T + tid;
            if (it * T >= n4) break;  // uniform
            const f32x4 v = i4 < n4 ? xr.v[it] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            quant4_to_lds<F16_D>(v, i4, nb, tid, s_lo, s_hi, s_d, s_sum);
        }
    } else {
        constexpr int MAXIT = BigX<XSRC_NORM>::MAXIT, NT = BigX<XSRC_NORM>::NT;  // T == NT (launcher)
        {
            double ss = 0.0;
#pragma unroll
            for (int it = 0; it < MAXIT; it++) {
                const int i4 = it * NT + tid;
                if (i4 < n4) {
                    ss += (double)(xr.v[it][0] * xr.v[it][0]);
                    ss += (double)(xr.v[it][1] * xr.v[it][1]);
                    ss += (double)(xr.v[it][2] * xr.v[it][2]);
                    ss += (double)(xr.v[it][3] * xr.v[it][3]);
                }
            }
            ss = wave_sum_f64(ss);
            if ((tid & 63) == 0) s_part[tid >> 6] = ss;
        }
        __syncthreads();
        {
            double tot = 0.0;
#pragma unroll
            for (int i = 0; i < NT / 64; i++) tot += s_part[i];
            // tot / n in f64: for a power-of-two row width (4096, 8192) the division is an exact scaling — the same bits as the
            // division, without its ~15 dependent f64 instructions on every wave of every workgroup
            const int n_el = nb * 32;
            const bool pow2 = (n_el & (n_el - 1)) == 0;  // uniform
            const float mean = pow2 ? (float)__builtin_ldexp(tot, -(31 - __builtin_clz((unsigned)n_el))) : (float)(tot / (double)n_el);
            const float scale = 1.0f / sqrtf(mean + d.eps);
#pragma unroll
            for (int it = 0; it < MAXIT; it++) {
                const int i4 = it * NT + tid;
                if (it * NT >= n4) break;  // uniform
                f32x4 y = {0.0f, 0.0f, 0.0f, 0.0f};
                if (i4 < n4) {
                    y[0] = (xr.v[it][0] * scale) * xr.w[it][0];
                    y[1] = (xr.v[it][1] * scale) * xr.w[it][1];
                    y[2] = (xr.v[it][2] * scale) * xr.w[it][2];
                    y[3] = (xr.v[it][3] * scale) * xr.w[it][3];
                    // (only the lm_head launch has the tap, and it stores the row at the END of the kernel: a store that MAY be pending
                    // makes every later wait a vmcnt(0) — gfx9 counts loads and stores in one counter and they return out of order
                    // with respect to each other)
                    if constexpr (YOUT) y_keep[it] = y;
                }
                quant4_to_lds<F16_D>(y, i4, nb, tid, s_lo, s_hi, s_d, s_sum);
            }
        }
    }
}

// INSTR: the measurement build (BigArgs::probe early exits, BigArgs::ts timeline stamps), launched only while option
// "probe" or "timeline" is set; the production instantiation (INSTR = false) carries none of those branches.
// `bid` of `G` workgroups run the launch (blockIdx.x / gridDim.x for k_mmvq_big itself; the producer workgroups of
// k_qkv_attn pass their index among the producers).
template <int QT, int EPI, int XSRC, bool INSTR>
__device__ __forceinline__ void big_body(const BigArgs &ba, const int bid, const int G) {
    const DecMmvqArgs &a = ba.d;
    const int probe = INSTR ? ba.probe : 0;
    long long *const ts = INSTR ? ba.ts : nullptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double s_part[16];
    __shared__ float s_rope[EPI == EPI_QKV ? 256 : 2];  // cos/sin of the RoPE angle of every pair of a head (D <= 256)
    constexpr bool F16_D = QT == QT_Q4_0 || QT == QT_Q5_0 || QT == QT_Q8_0;
    constexpr int RU = EPI == EPI_QKV ? 2 : 1, NW = EPI == EPI_GATE ? 2 : 1, NR = RU * NW;
    constexpr int PF = big_pf<QT>(NR);
#ifndef BIG_PF0
#define BIG_PF0 2
#endif
    constexpr int PF0 = PF < BIG_PF0 ? PF : BIG_PF0;  // steps requested before x is staged (see step 2)
    // all index arithmetic is 32-bit (rows <= 2^17, blocks per matrix < 2^27): 64-bit divides and multiplies in
    // the prologue cost ~1 us of VALU time per launch
    const int nb = (int)a.nb;
    const int nbl = (nb + 63) >> 6;
    const int nbp = nbl * 64;
    i32x4 *s_lo = (i32x4 *)smem;
    i32x4 *s_hi = s_lo + nbp;
    float *s_d = (float *)(s_hi + nbp);
    int *s_sum = (int *)(s_d + nbp);
    const int tid = threadIdx.x, lane = tid & 63;
    const int T = (int)blockDim.x;
    const int W = ba.wdeal > 0 ? ba.wdeal : T >> 6;  // waves per workgroup that take units: chosen per launch (launch_big)
    // wave-uniform values must be uniform FOR THE COMPILER too (scalar registers, scalar selects of the matrix
    // pointers): a kernarg array indexed by a "divergent" segment id is fetched with vector loads, and waiting for
    // those drains the whole in-order load queue at every step
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const long long t_entry = ts ? big_now() : 0;
    // ---- 1. the activation's loads go first (see BigX); so does the position (needed by the QKV epilogue only)
    int n_past = 0, store_at = 0;
    unsigned epoch = 0;
    if constexpr (EPI == EPI_QKV) {
        n_past = a.prm->n_past;
        store_at = a.prm->store_at;
        if (ba.gran) epoch = *ba.epoch;
    }
    BigX<XSRC> xr;
    xr.load(ba, nb, tid, T);
    const uint8_t *hotp = ba.hot ? (const uint8_t *)ba.hot : (const uint8_t *)a.w[0].d;
    const unsigned warm_v = *(const unsigned *)hotp;  // every wave touches the dummies' line with a plain load at entry: it stays cached
    // EPI_QKV: the last two waves fetch the token's RoPE table (k_rope_table); every wave issues the load so that all
    // load queues keep one compile-time shape
    f32x2 rope_pre = {0.0f, 0.0f};
    if constexpr (EPI == EPI_QKV) {
        const int kk = tid - (T - 128);
        rope_pre = ((const f32x2 *)ba.rope)[(kk >= 0 && kk < (a.D >> 1)) ? kk : 0];
    }

    // this wave's units: ((i * G + g) * 16 + wave), i < nu — at any moment the G workgroups together stream ONE
    // contiguous window of G*16 units of the matrix.  Lane i of the wave owns unit i's epilogue.
    const int M0 = (int)a.w[0].M, M1 = EPI == EPI_QKV ? (int)a.w[1].M : 0, M2 = EPI == EPI_QKV ? (int)a.w[2].M : 0;
    const int Utot = (M0 + M1 + M2) / RU;
    // wave-major within a round: the units of the last, partial round go to waves 0..k of EVERY workgroup, so all CUs
    // stream the same number of rows (11008 w1|w3 rows: 43 per CU instead of 45 on 222 CUs and 30 on 34)
    // XCD-affine dealing (BigArgs::aff_hpl): the same formula over the units of THIS workgroup's XCD label (blockIdx mod 8 = bid mod 8:
    // the attention workgroups in front are a multiple of 8) dealt to the G / 8 workgroups that carry the label
    bool affine = false;
    if constexpr (EPI == EPI_QKV) affine = ba.aff_hpl > 0;
    const int Gd = affine ? G >> 3 : G, bd = affine ? bid >> 3 : bid;
    const int Ucnt = affine ? (ba.aff_hpl * 3) << ba.aff_shift : Utot;
    const int u_first = wave * Gd + bd, u_stride = Gd * W;
    const int nu = (wave < W && u_first < Ucnt) ? (Ucnt - u_first + u_stride - 1) / u_stride : 0;  // <= 64 (launcher)
    // unit of the dealing -> unit of the launch (row pair index over wq|wk|wv): the identity unless the dealing is XCD-affine
    auto unit_of = [&](int lu) -> int {
        if constexpr (EPI == EPI_QKV) {
            if (affine) {
                const int q = lu >> ba.aff_shift, pr = lu & ((1 << ba.aff_shift) - 1);  // (head of the label, matrix), pair of the head
                const int hh = q / 3, mat = q - 3 * hh;
                return mat * (M0 >> 1) + (((bid & 7) + 8 * hh) << ba.aff_shift) + pr;
            }
        }
        return lu;
    };
    const int S = nu * nbl;
    // EPI_ADD: lane i preloads the residual of unit i (a load issued in the epilogue would drain the queue)
    float res_pre = 0.0f;
    if constexpr (EPI == EPI_ADD) {
        const int m = u_first + u_stride * lane;
        res_pre = a.res[(lane < nu && m < Utot) ? m : 0];
    }

    // unit -> (matrix, first row)
    auto resolve = [&](int i, int &sg, int &m0) {
        const int lu = u_first + u_stride * i;
        const int r = lu < Ucnt ? unit_of(lu) * RU : 0;  // (dummy prefetch steps of a wave without (enough) units: any valid row)
        sg = 0;
        m0 = r;
        if constexpr (EPI == EPI_QKV) {
            if (r >= M0 + M1) {
                sg = 2;
                m0 = r - M0 - M1;
            } else if (r >= M0) {
                sg = 1;
                m0 = r - M0;
            }
        }
    };
    // The loads of a step are UNCONDITIONAL (addresses clamped; lanes past the row end read block nb-1 and meet
    // zero x blocks in LDS): the number of memory operations in flight is a compile-time constant at every
    // wait, so hipcc waits for exactly the step it needs (s_waitcnt vmcnt(N)) instead of draining the queue.
    // The row bases of the producer's current unit live in scalar registers and change only when the producer moves
    // to the next unit: a step then costs two vector instructions of address math (block index, clamp) — the
    // per-step resolve + pointer selects + 64-bit vector adds it replaces were ~12 VALU and, for wq|wk|wv, ~50 SALU
    // per step, on a CU whose 16 waves share one scalar unit (in-kernel timeline: loads issued 1.6 us after entry).
    const uint8_t *ub_qs[NR], *ub_qs2[NR];
    const uint32_t *ub_qh[NR];
    const __half *ub_d[NR], *ub_m[NR];
    auto set_unit = [&](int i) {
        int sg, m0;
        resolve(i, sg, m0);
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const uint8_t *qs = a.w[0].qs, *qs2 = a.w[0].qs2;
            const uint32_t *qh = a.w[0].qh;
            const __half *wd = a.w[0].d, *wm = a.w[0].m;
            if constexpr (EPI == EPI_GATE) {
                if (k == 1) {
                    qs = a.w[1].qs; qs2 = a.w[1].qs2; qh = a.w[1].qh; wd = a.w[1].d; wm = a.w[1].m;
                }
            } else if constexpr (EPI == EPI_QKV) {  // scalar selects (sg is wave-uniform)
                qs = sg == 0 ? a.w[0].qs : sg == 1 ? a.w[1].qs : a.w[2].qs;
                qs2 = sg == 0 ? a.w[0].qs2 : sg == 1 ? a.w[1].qs2 : a.w[2].qs2;
                qh = sg == 0 ? a.w[0].qh : sg == 1 ? a.w[1].qh : a.w[2].qh;
                wd = sg == 0 ? a.w[0].d : sg == 1 ? a.w[1].d : a.w[2].d;
                wm = sg == 0 ? a.w[0].m : sg == 1 ? a.w[1].m : a.w[2].m;
            }
            const size_t ro = (size_t)(uint32_t)(m0 + (EPI == EPI_QKV ? k : 0)) * (uint32_t)nb;  // first block of the row
            ub_qs[k] = qs + ro * 16;
            ub_qs2[k] = qs2 + ro * 16;
            ub_qh[k] = qh + ro;
            ub_d[k] = wd + ro;
            ub_m[k] = wm + ro;
        }
    };
    // A step past the wave's last real one is a DUMMY: it exists so that every slot is refilled UNCONDITIONALLY and the number of
    // loads in flight is the same compile-time constant at every wait (a load under a branch makes hipcc take the smallest count
    // over the paths: the ring drained at the head of every pass, tests/tools/disasm.py).  A dummy reads BigArgs::hot — one
    // line for the whole chip, touched with a plain load at kernel entry, so a cache hit — through scalar-selected base pointers
    // (dummy is wave-uniform).  Re-reading the row's first block with the stream's non-temporal policy (round 2's dummy) is a
    // second HBM miss per load: with counted waits five such steps at the end of a wq|wk|wv wave (4 real steps on a ring of 3)
    // held every wave's exit back by ~1 us.
    auto issue = [&](BigStep<QT, NR> &st, int j, bool dummy) {
        const int b = lane + 64 * j;
        const uint32_t bc = dummy ? 0u : (uint32_t)(b < nb ? b : nb - 1);
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const uint8_t *pq = dummy ? hotp : ub_qs[k], *pq2 = dummy ? hotp : ub_qs2[k];
            const uint32_t *ph = dummy ? (const uint32_t *)hotp : ub_qh[k];
            const __half *pd = dummy ? (const __half *)hotp : ub_d[k], *pm = dummy ? (const __half *)hotp : ub_m[k];
            st.q[k] = __builtin_nontemporal_load((const u32x4 *)(pq + (size_t)bc * 16));
            if constexpr (QT == QT_Q8_0) st.p[k] = __builtin_nontemporal_load((const u32x4 *)(pq2 + (size_t)bc * 16));
            if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) st.h[k] = __builtin_nontemporal_load(ph + bc);
            // the f16 scale travels as the aligned 32-bit word that holds it (blocks 2i, 2i + 1) and is picked by block parity at
            // its use: a 16-bit load is a value hipcc packs two of into one register (v_perm_b32) right behind the loads
            st.dw[k] = *(const uint32_t *)(pd + (bc & ~1u));
            if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1) st.mw[k] = *(const uint32_t *)(pm + (bc & ~1u));
        }
    };

    if (probe == 1) return;  // here, not at entry: a check at entry costs every launch one more scalar-cache round trip
    // ---- 2. weight prologue, part 1: PF0 steps.  Enough to cover the latency of x, little enough that x is not
    //         queued behind tens of MB of weight requests in the fabric (measured with the in-kernel timeline:
    //         with the full ring requested up front x took 3..6 us to arrive)
    BigStep<QT, NR> ring[PF];
    int pi = 0, pj = 0;  // producer position (unit, column); steps past the wave's last one are dummies
    auto advance = [&](int k) {
        if (k + 1 < S && ++pj == nbl) {
            pj = 0;
            set_unit(++pi);
        }
    };
    set_unit(0);
#pragma unroll
    for (int k = 0; k < PF0; k++) {
        issue(ring[k], pj, k >= S);
        advance(k);
    }
    const long long t_issued = ts ? big_now() : 0;

    // ---- 3. norm / re-quantization of x into LDS; meanwhile the last two waves (not stagers) park the RoPE table
    if constexpr (EPI == EPI_QKV) {
        if (tid >= T - 128 && tid - (T - 128) < (a.D >> 1)) {
            const int kk = tid - (T - 128);
            s_rope[2 * kk] = rope_pre[0];
            s_rope[2 * kk + 1] = rope_pre[1];
        }
    }
    f32x4 y_keep[XSRC == XSRC_NORM && EPI == EPI_STORE ? BigX<XSRC_NORM>::MAXIT : 1];  // the normed row for the embedding tap (stored in the epilogue)
    big_stage_x<F16_D, XSRC, XSRC == XSRC_NORM && EPI == EPI_STORE>(ba, xr, nb, nbp, tid, T, s_lo, s_hi, s_d, s_sum, s_part, y_keep);
    const long long t_staged = ts ? big_now() : 0;
    // ---- 2b. the rest of the ring
#pragma unroll
    for (int k = PF0; k < PF; k++) {
        issue(ring[k], pj, k >= S);
        advance(k);
    }
    __syncthreads();
    const long long t_barrier = ts ? big_now() : 0;
    long long t_first = 0;
    if (probe == 2) return;

    // ---- 4. dots.  The unrolled body only accumulates; when a unit's last column is done its NR sums are reduced
    //         across the wave and parked in lane `unit index` (myv), the epilogues run afterwards, one lane each.
    float acc[NR], myv[NR];
#pragma unroll
    for (int k = 0; k < NR; k++) acc[k] = myv[k] = 0.0f;
    int ci = 0, cj = 0;  // consumer position
    auto dots = [&](const BigStep<QT, NR> &st, const int k, const bool first) {
        const int b = lane + 64 * cj;  // < nbp: the padded LDS blocks are zero
        const i32x4 lo = s_lo[b], hi = s_hi[b];
        const float xd = s_d[b];
        const int xs = s_sum[b];
        // which half of a scale word is this lane's block: the parity of the (clamped) block index the word was fetched
        // for — a lane past the row's end must pick the row's LAST scale, not its neighbour (the next row's first one, or
        // for the last row whatever lies behind the array: 0 * NaN is not 0)
        const uint32_t sh16 = (uint32_t)((b < nb ? b : nb - 1) & 1) << 4;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            if (probe == 4 && (k & 1)) continue;  // measurement: half the dots
            u32x4 p2 = st.q[r];
            uint32_t hh = 0;
            float mw = 0.0f;
            if constexpr (QT == QT_Q8_0) p2 = st.p[r];
            if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) hh = st.h[r];
            if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1) mw = big_h2f(st.mw[r] >> sh16);
            acc[r] += block_dot<QT>(st.q[r], p2, hh, big_h2f(st.dw[r] >> sh16), mw, lo, hi, xd, xs);
        }
        if (ts && first) t_first = acc[0] != 12345.678f ? big_now() : 1;  // first step's weights landed
        if (++cj == nbl) {
            cj = 0;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const float v = probe == 5 ? acc[r] : wave_sum_f32(acc[r]);  // 5: measurement, no reduction
                myv[r] = lane == ci ? v : myv[r];
                acc[r] = 0.0f;
            }
            ci++;
        }
    };
    // Full passes over the ring while a pass still has a step to request: every slot's dots are followed by that slot's refill,
    // UNCONDITIONALLY (a dummy where the wave has no step left, see `issue`) — PF steps are in flight at every wait, hipcc counts
    // them (s_waitcnt vmcnt((PF - 1) x loads per step)).  The last PF (or fewer) steps are drained by a pass that requests
    // nothing: its waits are static too, and the only dummies ever issued are the < PF ones that fill up the last full pass.
    int s = 0;
    for (; s + PF < S; s += PF) {
#pragma unroll
        for (int k = 0; k < PF; k++) {
            dots(ring[k], k, s + k == 0);
            const bool more = s + k + PF < S;
            issue(ring[k], pj, !more);
            if (more && ++pj == nbl) {
                pj = 0;
                if (s + k + PF + 1 < S) set_unit(++pi);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PF; k++)
        if (s + k < S) dots(ring[k], k, s + k == 0);  // wave-uniform
    const long long t_dots = ts ? big_now() : 0;

    // ---- 5. epilogues: lane i finishes unit i
    if (lane < nu && probe != 3) {
        int sg, m0;
        resolve(lane, sg, m0);
        if constexpr (EPI == EPI_STORE) {
            a.dst[m0] = myv[0];
        } else if constexpr (EPI == EPI_ADD) {
            a.dst[m0] = myv[0] + res_pre;
        } else if constexpr (EPI == EPI_GATE) {
            a.dst[m0] = silu_table(myv[0]) * myv[1];
        } else {  // EPI_QKV, see k_mmvq_dec
            const int p = store_at ? store_at : n_past;
            __half h0, h1;  // the pair as f16: what the K/V cache holds, and what ggml's F16 mat-mul makes of Q (src1 -> f16)
            if (sg == 2) {
                h0 = __float2half_rn(myv[0]);
                h1 = __float2half_rn(myv[1]);
                a.mem_v[(int64_t)m0 * a.C + p] = h0;
                a.mem_v[(int64_t)(m0 + 1) * a.C + p] = h1;
            } else {
                const int kk = (m0 % a.D) >> 1;
                const float c = s_rope[2 * kk], sn = s_rope[2 * kk + 1];
                const float r0 = myv[0] * c - myv[1] * sn, r1 = myv[0] * sn + myv[1] * c;
                h0 = __float2half_rn(r0);
                h1 = __float2half_rn(r1);
                if (sg == 0) {
                    if (!ba.gran) {
                        a.dst[m0] = r0;
                        a.dst[m0 + 1] = r1;
                    }
                } else {
                    a.mem_k[(int64_t)p * a.Egqa + m0] = h0;
                    a.mem_k[(int64_t)p * a.Egqa + m0 + 1] = h1;
                }
            }
            if (ba.gran) {  // one aligned 8-byte agent-scope (write-through) store: the data is the flag
                const unsigned v2 = (unsigned)__half_as_ushort(h0) | ((unsigned)__half_as_ushort(h1) << 16);
                unsigned long long *gp = ba.gran + unit_of(u_first + u_stride * lane);
                if (affine)
                    gran_store_l2(gp, epoch, v2);
                else
                    gran_store(gp, epoch, v2);
            }
        }
    }
    if constexpr (XSRC == XSRC_NORM && EPI == EPI_STORE) {  // the embedding tap: workgroup 0's copy of the normed row
        if (ba.y_out && bid == 0) {
#pragma unroll
            for (int it = 0; it < BigX<XSRC_NORM>::MAXIT; it++) {
                const int i4 = it * BigX<XSRC_NORM>::NT + tid;
                if (i4 < nb * 8) ((f32x4 *)ba.y_out)[i4] = y_keep[it];
            }
        }
    }
    if (warm_v == 0x7fc0dead && nb < 0) a.dst[0] = 0.0f;  // (never: keeps the warm-up load alive; its wait falls here, long after it landed)
    if (ts && wave == 0 && lane == 0) {
        const int q = G / ba.ts_wgs;
        if (q > 0 && bid % q == 0 && bid / q < ba.ts_wgs) {
            long long *o = ts + (bid / q) * 8;
            o[0] = t_entry; o[1] = t_issued; o[2] = t_staged; o[3] = t_barrier; o[4] = t_first; o[5] = big_now();
            o[6] = S | ((long long)(t_dots - t_entry) << 32); o[7] = bid;
        }
    }
}
template <int QT, int EPI, int XSRC, bool INSTR = false>
__global__ void __launch_bounds__(BIG_T) k_mmvq_big(const BigArgs ba) {
    big_body<QT, EPI, XSRC, INSTR>(ba, (int)blockIdx.x, (int)gridDim.x);
}
