// decode_fused.h — wq|wk|wv and the attention of a decode token in ONE launch (k_qkv_attn).
//
// The two launches it replaces are a dependent pair: k_mmvq_big<EPI_QKV> streams 28 MB on all CUs, then k_attn_decode runs a
// latency chain on n_head CUs with HBM idle — its K/V round trip, a kernel boundary and a cold start for ~100 KB of work per
// head (7.3-8.4 us of a 39 us layer at 7B).  Here the n_head attention workgroups are part of the mat-vec's launch:
//   * workgroups 0 .. n_head-1 (dispatched first): one head each.  They request the first 256 positions of the head's K and V
//     at kernel entry — that traffic overlaps the weight stream instead of following it — and then wait for the token's own
//     Q / K / V rows of their head;
//   * the other G - n_head workgroups run the mat-vec (big_body) over all row pairs, dealt exactly as in k_mmvq_big, and
//     publish every finished pair as an 8-byte {epoch, f16 x 2} granule besides storing K / V into the cache.  Q travels as
//     f16 because that is what ggml's F16 mat-mul makes of its src1 (k_attn_decode rounds the f32 Q the same way); the K / V
//     halves are the very halves the cache gets, so this token's row of the scores / of V.P comes from the granules and the
//     cache row that is being written concurrently is never read.
// The hand-off is Guideline 16's form R2 (the data is the flag: one aligned 8-byte agent-scope store per pair, agent-scope
// polling loads, no fence, no counter); it does not depend on which XCD or in which order workgroups run: producers never
// wait, consumers wait only for producers, and all G workgroups fit the chip at once (one 1024-thread workgroup per CU).  The
// tag is a device word bumped once per token (k_rope_table), so a replayed hipGraph never sees its own previous granules as
// current.  Every spin is bounded (GRAN_SPIN_MAX polls; FusedAttnArgs::err is raised and read back with the token's results: the host
// re-runs the token on the two-launch pair or aborts with a message, llama_plan.inc token_finish).
// Arithmetic = k_attn_decode's, hence ggml's: f32 dot of f16 K with f16 Q, scale, row max, f16-rounded exp of the f16-rounded
// difference, f64 sum, f16 probabilities, f32 V.P, Q8 re-quantization for wo.
#pragma once
#include "decode_big.h"
#include "decode_attn_split.h"  // attn_one_wait and the hand-off protocol of k_attn_split_one
#ifndef GRAN_SLEEP
#define GRAN_SLEEP 2  // s_sleep between two polls of a granule sweep (64-cycle units)
#endif

struct FusedAttnArgs {
    const __half *mem_k, *mem_v;  // + layer offset, layouts of DecMmvqArgs
    const DecParams *prm;
    const unsigned *epoch;
    const unsigned long long *gran;  // this layer's granules: one per row pair of wq|wk|wv
    int k_pair0, v_pair0;            // index of the first K / V pair (= rows of wq / 2, rows of wq|wk / 2)
    float scale;
    int D, n_rep, n_head;
    int64_t Egqa, C, Clds;
    int8_t *lo, *hi;  // the head's D outputs re-quantized for wo
    float *dq;
    int *sumq;
    long long *ts;  // optional timeline slot (as k_attn_decode)
    int ts_heads;   // heads the slot has room for: all of them (slot h) or four (every n_head / 4 th)
    int local_rows; // 1: the token's rows come from mat-vec workgroups of this workgroup's own XCD through its L2 (BigArgs::aff_hpl)
    unsigned *err;  // raised when a wait gave up
    // S > 1: S attention workgroups per head, workgroup s takes positions [512 s, 512 (s + 1)) (attn_consumer_split below)
    int S, layer;
    unsigned long long *mx_g, *sum_g, *part_g;  // this LAYER's hand-off granules of k_attn_split_one (decode_attn_split.h): [n_head][S], [..][S][2], [..][S][D]
    unsigned *cnt;                              // [n_head] arrival counters
    // WO form (k_qkv_attn<.., WO = true>): the heads' re-quantized outputs are PUBLISHED to the mat-vec workgroups of the same
    // launch instead of stored for a following wo launch: this layer's E/32 x OGRAN granules (publish_head_q8 below)
    unsigned long long *ogran;
    // K plan (k_qkv_attn_k, kernels/kquant_big.h): the merged heads as f32 [n_head * D] instead of Q8_0 blocks (wo stages its own
    // Q8_K row from them); nullptr = the Q8_0 forms above
    float *out_f32;
};

// ---------------------------------------------------------------------------------------------------
// wo under the attention's tail.  After its last wq|wk|wv row pair (about 7 us into a 7B launch) a mat-vec workgroup is idle
// while the attention workgroups run their latency chain (4 us), and the wo launch that follows pays a kernel boundary, a cold
// start and 9.4 MB of weights on top.  In the WO form every mat-vec workgroup instead pulls ITS rows of wo (row m of wo belongs
// to mat-vec workgroup m mod G; 42 KB at 7B) into registers while the attention runs, then gathers the heads' outputs — E int8
// quants + E/32 scales and sums = 1280 granules at 7B, published by the n_head attention workgroups as they finish — and computes
// its rows of wo.x + residual.  Same per-row arithmetic as k_mmvq_big<EPI_ADD, XSRC_Q8> (lane-strided blocks, steps in
// order, DPP wave reduction, + residual): BIT-IDENTICAL to the separate launch.  Needs the whole launch resident like every
// in-launch hand-off here, and now in BOTH directions (attention waits for the mat-vec's rows, the mat-vec's second phase for the
// attention): the launcher takes it only while this slot has the GPU's CUs to itself (fused_qkv_shape).
// ---------------------------------------------------------------------------------------------------
#define OGRAN 10  // granules per Q8 block of the attention output: 4 words of quants 0..15, 4 of quants 16..31, d, sum

// ---------------------------------------------------------------------------------------------------
// L2 warm-up of the NEXT launch from the window in which HBM idles.  Between a mat-vec workgroup's last wq|wk|wv row pair and the
// first head output it can gather (6.1 -> 13.5 us of a 17 us launch at 7B, profiles/r06_wo_timeline_128.txt) nothing streams but
// 42 KB of wo per workgroup.  One wave per workgroup spends that window touching the first rows of w1|w3 — one 4-byte LDS-DMA
// load per 128-byte line, no VGPR, nothing waits for the data — so that the next launch finds them in the L2 OF THE XCD THAT
// READS THEM: k_mmvq_big deals row m to workgroup m mod G, and a launch's workgroups go to XCDs round robin by their index
// (observed, never relied on for anything but speed: MI355X_MICROARCH.md "Workgroup dispatch"), so rows = x mod 8 are read by the
// workgroups with blockIdx = x mod 8 of the next launch — and warmed by the workgroups with blockIdx = x mod 8 of this one.
// (NOT by the hardware's XCC_ID register: round 6 first keyed the rows on s_getreg(HW_REG_XCC_ID) and warmed the wrong L2s — the
// register numbers the dies physically, the dispatcher's round robin does not follow that order; the memory-side counters showed
// w1|w3 fetching all its bytes again, and the gain was a tenth of what the right mapping gives: all mat-vec launches of a 7B
// token 1.183 ms without, 1.175 with XCC_ID, 1.123 with blockIdx, gpurun_out/r6/run33-34.)  An XCD's L2 keeps its lines
// across the kernel boundary (tests/tools/overlap_probe3.hip).  Round 2 tried this from the spare workgroups of the then separate
// attention launch and lost what it won (the heads' K/V round trips queued behind the warm-up, and the launch ended when the
// warm-up did); here the heads' K/V are requested at kernel entry, the warming wave's own polls are the only thing behind its
// requests, and the launch cannot end before the gather.  Nothing is computed from the warmed bytes: results cannot change.
// ---------------------------------------------------------------------------------------------------
constexpr int WARM_MAX = 10;  // arrays: qs, d (+ qs2 | qh | m) of w1 and of w3
struct NextWarm {
    const uint8_t *base[WARM_MAX];
    uint32_t row_bytes[WARM_MAX];  // bytes of one matrix row in that array
    int rows_of[WARM_MAX];         // rows 0 .. rows_of[a] - 1 of array a are warmed (w1|w3: the next launch; w2: the one behind it)
    int n;                         // arrays in use
    int rows;                      // > 0: the warm-up is on
    const uint8_t *bcast;          // a small array EVERY workgroup of the next launch reads (the norm weights): one workgroup per XCD takes it
    int bcast_bytes;
};
// called by ONE wave of mat-vec workgroup `bid` of `G`; `junk` = 256 bytes of LDS nobody reads
__device__ __forceinline__ void warm_next(const NextWarm &nw, const int bid, const int G, const int lane, unsigned *junk) {
    typedef const __attribute__((address_space(1))) void *wg_ptr;
    typedef __attribute__((address_space(3))) void *wl_ptr;
    const int x = (int)blockIdx.x & 7;  // this workgroup's place in the dispatcher's round robin = that of the workgroups it warms for
    const int j = bid >> 3, nj = (G + 7) >> 3;   // this workgroup among its XCD's (round-robin placement assumed, for speed only)
    for (int a = 0; a < nw.n; a++) {
        const int K = (nw.rows_of[a] + 7 - x) >> 3;  // rows x, x + 8, ... of the first rows_of[a]
        const int c = (K + nj - 1) / nj;
        const int k0 = j * c, cnt = K - k0 < c ? K - k0 : c;
        const uint32_t rb = nw.row_bytes[a];
        const int lpr = (int)((rb + 127) >> 7);  // samples per row, 128 bytes apart + the row's last word: every line it touches
        const uint8_t *base = nw.base[a];
        const float inv = 1.0f / (float)lpr;
        for (int i = lane; i < cnt * lpr; i += 64) {
            const int r = (int)(((float)i + 0.5f) * inv), l = i - r * lpr;  // exact: i < 2^16, lpr <= 2^8
            uint32_t off = (uint32_t)l << 7;
            off = off + 4 > rb ? rb - 4 : off;
            const uint8_t *p = base + (size_t)(uint32_t)(x + 8 * (k0 + r)) * rb + off;
            __builtin_amdgcn_global_load_lds((wg_ptr)p, (wl_ptr)junk, 4, 0, 0);
        }
    }
    if (j == 0 && nw.bcast)
        for (int i = lane * 128; i < nw.bcast_bytes; i += 64 * 128)
            __builtin_amdgcn_global_load_lds((wg_ptr)(nw.bcast + i), (wl_ptr)junk, 4, 0, 0);
}

// the XCD every workgroup of a num_cus x 1024-thread launch runs on (llama_plan.inc xcd_labels_ok: does blockIdx mod 8 name the XCD?)
__global__ void __launch_bounds__(1024) k_xcc_ids(unsigned *out) {
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        out[blockIdx.x] = v & 0xf;
    }
}

struct WoTailArgs {
    QWeight w;            // wo
    const float *res;     // residual (the layer's input row)
    float *dst;           // wo.x + residual
    int cap;              // (unused)
    long long *ts;        // optional timeline slot (option "timeline"): 8 x int64 per sampled workgroup, as big_body's
    int ts_wgs;
    NextWarm warm;        // the next launch's first rows (w1|w3), warmed into L2 while the attention runs
    int warm_wave;        // the wave that issues the warm-up
};
constexpr int WO_NG_MAX = 2;  // granules per thread it may gather (E / 32 * OGRAN <= 2048)

// the D outputs of head h (already in LDS s_o[D]) as Q8 blocks: plain stores for a following wo launch, or granules for the WO form
template <bool F16_D>
__device__ __forceinline__ void publish_head_q8(const FusedAttnArgs &f, const int h, const float v, const int nblk, const int l, const int b,
                                                const unsigned epoch) {
    float amax = fabsf(v);
    amax = g32_max_f32(amax);
    const float d = amax / 127.0f;
    const float id = act_id(amax, d, aq_scalar());
    const int qv = act_q(v * id, aq_scalar());
    int sq = qv;
    sq = g32_sum_i32(sq);
    const int64_t gb = (int64_t)h * nblk + b;
    if (f.ogran) {
        // four neighbouring lanes' bytes -> one word (disjoint bits: the quad's OR by two DPP steps), published by the quad's first lane
        int wv = (qv & 0xFF) << (8 * (l & 3));
        wv |= dpp_i32<DPP_QUAD_XOR1>(wv);
        wv |= dpp_i32<DPP_QUAD_XOR2>(wv);
        unsigned long long *go = f.ogran + gb * OGRAN;
        if ((l & 3) == 0) gran_store(go + (l >> 2), epoch, (unsigned)wv);
        if (l == 0) {
            gran_store(go + 8, epoch, __float_as_uint(F16_D ? round_f16(d) : d));
            gran_store(go + 9, epoch, (unsigned)sq);
        }
        return;
    }
    (l < 16 ? f.lo : f.hi)[gb * 16 + (l & 15)] = (int8_t)qv;
    if (l == 0) {
        f.dq[gb] = F16_D ? round_f16(d) : d;
        f.sumq[gb] = sq;
    }
}

// second phase of a mat-vec workgroup in the WO form (bid of G, as in big_body); smem = the dynamic LDS of the launch, whose first
// nbp * 40 bytes (the staged activation of wq|wk|wv) are re-used for the gathered attention output.  Wave w owns rows
// w, w + 16 (of the workgroup's rows bid + G i) and keeps their blocks in REGISTERS, unpacked to int8 codes while the attention
// still runs: behind the gather only 8 v_dot4 + the scale formula per block are left (block_dot_codes = block_dot's arithmetic).
constexpr int WO_RW = 2;  // rows per wave (the workgroup holds at most 32 rows: ceil(M / G) <= 32)
template <int QT, int NBLT>
__device__ __forceinline__ void wo_tail(const WoTailArgs &t, const FusedAttnArgs &f, const int bid, const int G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nb = (int)t.w.nb, M = (int)t.w.M;
    const int nbl = (nb + 63) >> 6, nbp = nbl * 64;  // nbl <= NBLT (launcher)
    i32x4 *s_lo = (i32x4 *)smem;
    i32x4 *s_hi = s_lo + nbp;
    float *s_d = (float *)(s_hi + nbp);
    int *s_sum = (int *)(s_d + nbp);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned epoch = *f.epoch;
    const int nrows = bid < M ? (M - bid + G - 1) / G : 0;
    const int nrw = wave < nrows ? (nrows - wave + 15) >> 4 : 0;  // rows of this wave: <= WO_RW
    const long long t0 = t.ts ? (long long)wall_clock64() : 0;
    // ---- this wave's rows of wo: requested now, they land while the attention runs
    u32x4 q[WO_RW][NBLT], q2[QT == QT_Q8_0 ? WO_RW : 1][QT == QT_Q8_0 ? NBLT : 1];
    uint32_t qh[(QT == QT_Q5_0 || QT == QT_Q5_1) ? WO_RW : 1][(QT == QT_Q5_0 || QT == QT_Q5_1) ? NBLT : 1];
    __half dw[WO_RW][NBLT], mw[(QT == QT_Q4_1 || QT == QT_Q5_1) ? WO_RW : 1][(QT == QT_Q4_1 || QT == QT_Q5_1) ? NBLT : 1];
#pragma unroll
    for (int r = 0; r < WO_RW; r++) {
        const int m = bid + G * (wave + 16 * r);
        const size_t ro = (size_t)(uint32_t)(r < nrw ? m : 0) * (uint32_t)nb;
#pragma unroll
        for (int j = 0; j < NBLT; j++) {
            const int b = lane + 64 * j;
            const size_t o = ro + (uint32_t)(b < nb ? b : nb - 1);  // clamped: the loads are unconditional, invalid ones are not used
            q[r][j] = __builtin_nontemporal_load((const u32x4 *)(t.w.qs + o * 16));
            if constexpr (QT == QT_Q8_0) q2[r][j] = __builtin_nontemporal_load((const u32x4 *)(t.w.qs2 + o * 16));
            if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) qh[r][j] = __builtin_nontemporal_load(t.w.qh + o);
            dw[r][j] = t.w.d[o];
            if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1) mw[r][j] = t.w.m[o];
        }
    }
    // the residual of this wave's rows: lane r preloads row (wave + 16 r) of the workgroup
    const float res_pre = t.res[lane < nrw ? bid + G * (wave + 16 * lane) : 0];
    const long long t1 = t.ts ? (long long)wall_clock64() : 0;
    __syncthreads();  // every wave is done with the staged activation of wq|wk|wv: its LDS is re-used from here on
    // ---- this workgroup's share of HBM is idle from here until the heads are done: one wave warms the next launch's first rows
    //      (see NextWarm) — behind its own rows of wo in its queue, ahead of its polls by microseconds
    if (t.warm.rows > 0 && wave == t.warm_wave) {
        __shared__ unsigned s_junk[64];
        warm_next(t.warm, bid, G, lane, s_junk);
    }
    for (int i = nb + tid; i < nbp; i += 1024) {  // padded blocks (a row that does not fill its last 64-block step) stay zero
        s_lo[i] = i32x4{0, 0, 0, 0};
        s_hi[i] = i32x4{0, 0, 0, 0};
        s_d[i] = 0.0f;
        s_sum[i] = 0;
    }
    uint32_t wl[WO_RW][NBLT][4], wh[WO_RW][NBLT][4];
#pragma unroll
    for (int r = 0; r < WO_RW; r++)
#pragma unroll
        for (int j = 0; j < NBLT; j++) {
            u32x4 p2 = q[r][j];
            uint32_t hh = 0;
            if constexpr (QT == QT_Q8_0) p2 = q2[r][j];
            if constexpr (QT == QT_Q5_0 || QT == QT_Q5_1) hh = qh[r][j];
            block_unpack<QT>(q[r][j], p2, hh, wl[r][j], wh[r][j]);
        }
    const long long t2 = t.ts ? (long long)wall_clock64() : 0;
    // ---- the heads' outputs: nb * OGRAN granules, swept by all threads until every tag is this token's epoch
    const int ngran = nb * OGRAN;
#pragma unroll
    for (int k = 0; k < WO_NG_MAX; k++) {
        const int idx = tid + 1024 * k;
        if (1024 * k + 64 * wave >= ngran) break;  // wave-uniform: a wave with no granule of this batch polls nothing (clamped to granule 0,
                                                   // twelve waves of every workgroup used to hammer the line head 0 publishes into)
        const unsigned long long *gp = f.ogran + (idx < ngran ? idx : 0);
        unsigned long long x;
        for (int spin = 0;; spin++) {
            x = gran_load(gp);
            const bool ok = (unsigned)(x >> 32) == epoch;
            if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
            __builtin_amdgcn_s_sleep(GRAN_SLEEP);
            if (spin > GRAN_SPIN_MAX) {
                if (lane == 0) __hip_atomic_store(f.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if (idx < ngran) {
            const int blk = idx / OGRAN, j = idx - blk * OGRAN;
            const unsigned v = (unsigned)x;
            if (j < 4)
                ((unsigned *)s_lo)[blk * 4 + j] = v;
            else if (j < 8)
                ((unsigned *)s_hi)[blk * 4 + (j - 4)] = v;
            else if (j == 8)
                s_d[blk] = __uint_as_float(v);
            else
                s_sum[blk] = (int)v;
        }
    }
    const long long t3 = t.ts ? (long long)wall_clock64() : 0;
    __syncthreads();
    const long long t4 = t.ts ? (long long)wall_clock64() : 0;
    // ---- rows: one row = nbl steps of 64 blocks, accumulated per lane in step order, DPP wave reduction: as in big_body
    float myv = 0.0f;
#pragma unroll
    for (int r = 0; r < WO_RW; r++) {
        if (r >= nrw) break;  // uniform
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < NBLT; j++) {
            const int b = lane + 64 * j;
            if (b < nb) {
                float m_ = 0.0f;
                if constexpr (QT == QT_Q4_1 || QT == QT_Q5_1) m_ = __half2float(mw[r][j]);
                acc += block_dot_codes<QT>(wl[r][j], wh[r][j], __half2float(dw[r][j]), m_, s_lo[b], s_hi[b], s_d[b], s_sum[b]);
            }
        }
        const float v = wave_sum_f32(acc);
        myv = lane == r ? v : myv;
    }
    if (lane < nrw) t.dst[bid + G * (wave + 16 * lane)] = myv + res_pre;
    if (t.ts && tid == 0) {
        const int qq = G / t.ts_wgs;
        if (qq > 0 && bid % qq == 0 && bid / qq < t.ts_wgs) {
            long long *o = t.ts + (bid / qq) * 8;
            o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4; o[5] = (long long)wall_clock64(); o[6] = nrows; o[7] = bid;
        }
    }
}

__device__ __forceinline__ f16x2 u32_as_h2(unsigned u) { return __builtin_bit_cast(f16x2, u); }

template <bool F16_D>
__device__ __forceinline__ void attn_consumer(const FusedAttnArgs &f, const int h) {
    const long long t_entry = f.ts ? (long long)wall_clock64() : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t Clds = f.Clds, C = f.C, Egqa = f.Egqa;
    const int D = f.D;
    float *s_s = (float *)smem;             // Clds scores
    float *s_o = s_s + Clds;                // D outputs
    _Float16 *s_p = (_Float16 *)(s_o + D);  // Clds probabilities as f16
    __shared__ float s_red[16];
    __shared__ double s_redd[16];
    __shared__ __attribute__((aligned(16))) unsigned s_new[3 * 64];  // this token's Q | K | V of the head, f16 pairs (D <= 128)
    const int hk = h / f.n_rep;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_past = f.prm->n_past;
    const unsigned epoch = *f.epoch;
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // ---- the head's cache: up to 512 positions (the split-attention threshold) go into registers now, while the mat-vec
    //      workgroups stream — exactly the rows the context has (the position is known after one scalar round trip; the
    //      requests have microseconds of slack), so a short context costs the weight stream next to nothing
    constexpr int NPRE = 8;
    const int T = n_past + 1;
    const int T8 = (T + 7) & ~7;
    const int g = tid >> 4, gl = tid & 15;
    const int d0 = gl * 8;
    const bool act = d0 < D;
    const __half *kbase = f.mem_k + (int64_t)hk * D + d0;
    f16x8 kv[NPRE];
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        const int t = g + 64 * u;
        kv[u] = zero8;
        if (act && t < n_past) kv[u] = *(const f16x8 *)(kbase + (int64_t)t * Egqa);
    }
    const int cv = wave * 8 + (lane >> 3), pj = (lane & 7) * 8;
    const bool vact = cv < D;
    const __half *vbase = f.mem_v + ((int64_t)hk * D + cv) * C + pj;
    f16x8 vv[NPRE];
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        vv[u] = zero8;
        if (vact && 64 * u + pj < T8) vv[u] = *(const f16x8 *)(vbase + 64 * u);  // T8 <= C: the chunk lies inside the cache
    }

    // ---- this token's rows of the head: waves 0 / 1 / 2 sweep the D/2 granules of Q / K / V until every tag is the epoch
    if (wave < 3) {
        const int half_d = D >> 1;
        const int base = wave == 0 ? h * half_d : wave == 1 ? f.k_pair0 + hk * half_d : f.v_pair0 + hk * half_d;
        const unsigned long long *gp = f.gran + base + (lane < half_d ? lane : 0);
        unsigned long long x;
        for (int spin = 0;; spin++) {
            x = f.local_rows ? gran_load_l2(gp) : gran_load(gp);
            const bool ok = (unsigned)(x >> 32) == epoch;
            if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
            __builtin_amdgcn_s_sleep(GRAN_SLEEP);
            if (spin > GRAN_SPIN_MAX) {  // a producer never arrived (see GRAN_SPIN_MAX, kernels/common.h)
                if (lane == 0) __hip_atomic_store(f.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        s_new[wave * 64 + lane] = (unsigned)x;
    }
    __syncthreads();
    const long long t_loaded = f.ts ? (long long)wall_clock64() : 0;
    f16x2 qh2[4];
    f16x8 knew = zero8;
    if (act) {
        const u32x4 q4 = *(const u32x4 *)(s_new + (d0 >> 1));
        const u32x4 k4 = *(const u32x4 *)(s_new + 64 + (d0 >> 1));
#pragma unroll
        for (int j = 0; j < 4; j++) {
            qh2[j] = u32_as_h2(q4[j]);
            const f16x2 kk = u32_as_h2(k4[j]);
            knew[2 * j] = kk[0];
            knew[2 * j + 1] = kk[1];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) qh2[j] = f16x2{(_Float16)0.0f, (_Float16)0.0f};
    }

    // ---- scores ----
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        if (64 * u >= T) break;  // the slab starts beyond the context (uniform over the workgroup)
        const int t = g + 64 * u;
        const f16x8 kr = t == n_past ? knew : kv[u];  // the token's own row comes from the granules
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) s = __builtin_amdgcn_fdot2(f16x2{kr[2 * j], kr[2 * j + 1]}, qh2[j], s, false);
        s = g16_sum_f32(s);
        if (gl == 0 && t < T) s_s[t] = s * f.scale;
    }
#pragma unroll 1
    for (int t0 = 64 * NPRE; t0 < T; t0 += 64) {  // contexts beyond the register window (option attn_split raised): from the cache
        const int t = t0 + g;
        f16x8 kr = zero8;
        if (act && t < n_past) kr = *(const f16x8 *)(kbase + (int64_t)t * Egqa);
        if (t == n_past) kr = knew;
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) s = __builtin_amdgcn_fdot2(f16x2{kr[2 * j], kr[2 * j + 1]}, qh2[j], s, false);
        s = g16_sum_f32(s);
        if (gl == 0 && t < T) s_s[t] = s * f.scale;
    }
    __syncthreads();
    const long long t_scores = f.ts ? (long long)wall_clock64() : 0;
    // ---- softmax (ggml: max, f16-rounded exp of the f16-rounded difference, f64 sum, scale by 1/sum) ----
    // Up to 256 positions one wave does it alone (4 per lane, DPP reductions): no exchange through LDS, no barrier between the
    // passes.  The f64 sum of f16-valued terms is exact, so its order is immaterial.
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
    for (int t = tid; t < T8; t += 1024) s_p[t] = t < T ? (_Float16)(s_s[t] * inv) : (_Float16)0.0f;
    }
    __syncthreads();
    const long long t_softmax = f.ts ? (long long)wall_clock64() : 0;
    // ---- V.P ----
    {
        // this token's V of channel cv: element cv of the V granules
        const unsigned vpair = vact ? s_new[128 + (cv >> 1)] : 0u;
        const _Float16 vnew = u32_as_h2(vpair)[cv & 1];
        float acc = 0.0f;
        auto chunk = [&](f16x8 vr, const int pos) {  // 8 positions of channel cv against their probabilities
            if ((pos >> 6) == (n_past >> 6)) {  // the 64-slab of the token's own position (uniform)
                const int e = n_past - pos;
#pragma unroll
                for (int j = 0; j < 8; j++) vr[j] = e == j ? vnew : vr[j];
            }
            const f16x8 pp = *(const f16x8 *)(s_p + pos);
#pragma unroll
            for (int j = 0; j < 4; j++)
                acc = __builtin_amdgcn_fdot2(f16x2{vr[2 * j], vr[2 * j + 1]}, f16x2{pp[2 * j], pp[2 * j + 1]}, acc, false);
        };
#pragma unroll
        for (int u = 0; u < NPRE; u++) {
            if (64 * u >= T8) break;  // uniform
            if (64 * u + pj < T8) chunk(vv[u], 64 * u + pj);
        }
#pragma unroll 1
        for (int p0 = 64 * NPRE; p0 < T8; p0 += 64)
            if (p0 + pj < T8) chunk(vact ? *(const f16x8 *)(vbase + p0) : zero8, p0 + pj);
        acc = g8_sum_f32(acc);
        if ((lane & 7) == 0 && vact) s_o[cv] = acc;
    }
    __syncthreads();
    const long long t_vp = f.ts ? (long long)wall_clock64() : 0;
    // ---- the head's D outputs as Q8 blocks for wo ----
    const int nblk = D / 32, l = tid & 31, b = tid >> 5;
    if (b < nblk) {
        if (f.out_f32)
            f.out_f32[(int64_t)h * D + b * 32 + l] = s_o[b * 32 + l];
        else
            publish_head_q8<F16_D>(f, h, s_o[b * 32 + l], nblk, l, b, epoch);
    }
    if (f.ts && tid == 0) {
        const int q4 = f.n_head / 4;
        const bool all = f.ts_heads >= f.n_head;
        if (all || (q4 > 0 && h % q4 == 0 && h / q4 < 4)) {
            long long *o = f.ts + (all ? h : h / q4) * 8;
            o[0] = t_entry; o[1] = t_loaded; o[2] = t_scores; o[3] = t_softmax; o[4] = t_vp;
            o[5] = (long long)wall_clock64();
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's granule stores are acknowledged (write-through: by the memory side)
            o[6] = (long long)wall_clock64(); o[7] = h;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Contexts beyond the 512-position register window: S attention workgroups per head instead of one (S = 2 up to 1024
// positions, 4 up to 2048), still inside the wq|wk|wv launch.  Workgroup (h, s) owns positions [512 s, 512 (s + 1)): it
// requests exactly those K / V rows at entry (they stream under the weights), waits for the token's Q (and, if the token's own
// position falls into its range, K / V) granules, and then runs k_attn_split_one's protocol with its S - 1 peers
// (decode_attn_split.h): range maximum -> row maximum, range sum of the f16-rounded exps -> row sum (exact in any order), partial
// V.P -> the last workgroup of the head to arrive adds the partials (s ascending) and re-quantizes for wo.  ggml's rounding points
// are kept; only the f32 association of the V.P sum differs from the one-workgroup form (as for every split of a head).
// ---------------------------------------------------------------------------------------------------
template <bool F16_D>
__device__ __forceinline__ void attn_consumer_split(const FusedAttnArgs &f, const int h, const int s) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t C = f.C, Egqa = f.Egqa;
    const int D = f.D, S = f.S;
    constexpr int WIN = 512;
    float *s_s = (float *)smem;               // WIN scores, then exps, of the range
    float *s_o = s_s + WIN;                   // D floats (unused here, keeps the layout of attn_consumer's dynamic LDS in mind)
    _Float16 *s_p = (_Float16 *)(s_o + 128);  // WIN probabilities
    __shared__ float s_red[16];
    __shared__ double s_redd[16];
    __shared__ float s_mx;
    __shared__ double s_tot;
    __shared__ int s_last;
    __shared__ __attribute__((aligned(16))) unsigned s_new[3 * 64];
    const int hk = h / f.n_rep;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_past = f.prm->n_past;
    const unsigned epoch = *f.epoch;
    const unsigned tag = epoch;  // all 32 bits of the token's epoch: mx_g / sum_g / part_g are this layer's own buffers
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int NPRE = 8;
    const int T = n_past + 1;
    const int t0 = s * WIN;
    const int n = T > t0 ? (T - t0 < WIN ? T - t0 : WIN) : 0;  // positions of this range (0: the context ends before it)
    const int n8 = (n + 7) & ~7;
    const bool own = n_past >= t0 && n_past < t0 + WIN;        // the token's own position lies in this range
    const int g = tid >> 4, gl = tid & 15;
    const int d0 = gl * 8;
    const bool act = d0 < D;
    const __half *kbase = f.mem_k + (int64_t)hk * D + d0;
    f16x8 kv[NPRE];
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        const int t = t0 + g + 64 * u;
        kv[u] = zero8;
        if (act && t < n_past) kv[u] = *(const f16x8 *)(kbase + (int64_t)t * Egqa);
    }
    const int cv = wave * 8 + (lane >> 3), pj = (lane & 7) * 8;
    const bool vact = cv < D;
    const __half *vbase = f.mem_v + ((int64_t)hk * D + cv) * C + t0 + pj;
    f16x8 vv[NPRE];
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        vv[u] = zero8;
        if (vact && 64 * u + pj < n8) vv[u] = *(const f16x8 *)(vbase + 64 * u);  // t0 + n8 <= C: the chunk lies inside the cache
    }
    // ---- Q first: the scores of the cached positions and the exchange of their range maxima need nothing else.  (The mat-vec
    //      waves hand all their rows over at the end of their work — epilogues after the dots — so Q, K and V become visible
    //      almost together; what this order buys is that the row maximum is settled by ONE exchange among values every range
    //      already holds, plus a new-position score every workgroup computes itself: no exchange has to wait for the K row.
    //      Handing Q over inside the mat-vec loop was measured: profiles/r04_negative_results.txt item 5.)
    auto wait_rows = [&](int which /* 0 Q, 1 K, 2 V */) {  // one wave: the D/2 granules of this head's row -> s_new[which * 64 ..]
        const int half_d = D >> 1;
        const int base = which == 0 ? h * half_d : which == 1 ? f.k_pair0 + hk * half_d : f.v_pair0 + hk * half_d;
        const unsigned long long *gp = f.gran + base + (lane < half_d ? lane : 0);
        unsigned long long x;
        for (int spin = 0;; spin++) {
            x = f.local_rows ? gran_load_l2(gp) : gran_load(gp);  // (all S workgroups of head h sit at blockIdx = h mod n_head: one XCD)
            const bool ok = (unsigned)(x >> 32) == epoch;
            if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) break;
            __builtin_amdgcn_s_sleep(GRAN_SLEEP);
            if (spin > GRAN_SPIN_MAX) {
                if (lane == 0) __hip_atomic_store(f.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        s_new[which * 64 + lane] = (unsigned)x;
    };
    if (wave == 0) wait_rows(0);
    for (int i = tid; i < WIN; i += 1024) s_p[i] = (_Float16)0.0f;
    __syncthreads();
    f16x2 qh2[4];
    if (act) {
        const u32x4 q4 = *(const u32x4 *)(s_new + (d0 >> 1));
#pragma unroll
        for (int j = 0; j < 4; j++) qh2[j] = u32_as_h2(q4[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) qh2[j] = f16x2{(_Float16)0.0f, (_Float16)0.0f};
    }
    // scores of the range's CACHED positions (t < n_past) -> LDS; their maximum -> peers
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        if (64 * u >= n) break;  // uniform
        const int tl = g + 64 * u;
        float sc = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) sc = __builtin_amdgcn_fdot2(f16x2{kv[u][2 * j], kv[u][2 * j + 1]}, qh2[j], sc, false);
        sc = g16_sum_f32(sc);
        if (t0 + tl < n_past) {
            sc *= f.scale;
            if (gl == 0) s_s[tl] = sc;
            mx = fmaxf(mx, sc);
        }
    }
    mx = wave_max_f32(mx);
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    if (wave == 0) {
        float m = lane < 16 ? s_red[lane] : -INFINITY;
        m = wave_max_f32(m);
        if (lane == 0) gran_store(f.mx_g + (int64_t)h * S + s, tag, __float_as_uint(m));  // -inf: no cached position here
        float pm = -INFINITY;
        if (lane < S) pm = __uint_as_float((unsigned)attn_one_wait(f.mx_g + (int64_t)h * S + lane, tag, f.err));
        pm = wave_max_f32(pm);
        if (lane == 0) s_mx = pm;
    } else if (wave == 1) {
        // ---- the token's K row: EVERY workgroup of the head takes it and scores the new position itself, so the row maximum
        //      needs no second exchange behind the mat-vec; its V row only where the position lives
        wait_rows(1);
    } else if (wave == 2 && own) {
        wait_rows(2);
    }
    __syncthreads();
    {
        f16x8 knew = zero8;
        if (act) {
            const u32x4 k4 = *(const u32x4 *)(s_new + 64 + (d0 >> 1));
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const f16x2 kk = u32_as_h2(k4[j]);
                knew[2 * j] = kk[0];
                knew[2 * j + 1] = kk[1];
            }
        }
        float sc = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) sc = __builtin_amdgcn_fdot2(f16x2{knew[2 * j], knew[2 * j + 1]}, qh2[j], sc, false);
        sc = g16_sum_f32(sc) * f.scale;  // every 16-lane group holds the same value
        mx = fmaxf(s_mx, sc);
        if (own && tid == 0) s_s[n_past - t0] = sc;
    }
    __syncthreads();
    // ---- f16-rounded exps of the range, range sum -> peers, row sum (exact in any order)
    double sum = 0.0;
    for (int i = tid; i < n; i += 1024) {
        const float e = round_f16(expf(round_f16(s_s[i] - mx)));
        s_s[i] = e;
        sum += (double)e;
    }
    sum = wave_sum_f64(sum);
    if (lane == 0) s_redd[wave] = sum;
    __syncthreads();
    if (wave == 0) {
        if (lane == 0) {
            double loc = 0.0;
#pragma unroll
            for (int i = 0; i < 16; i++) loc += s_redd[i];
            const unsigned long long bits = (unsigned long long)__double_as_longlong(loc);
            gran_store(f.sum_g + ((int64_t)h * S + s) * 2, tag, (unsigned)(bits >> 32));
            gran_store(f.sum_g + ((int64_t)h * S + s) * 2 + 1, tag, (unsigned)bits);
        }
        double part = 0.0;
        if (lane < S) {
            const unsigned hi = (unsigned)attn_one_wait(f.sum_g + ((int64_t)h * S + lane) * 2, tag, f.err);
            const unsigned lo = (unsigned)attn_one_wait(f.sum_g + ((int64_t)h * S + lane) * 2 + 1, tag, f.err);
            part = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        }
        part = wave_sum_f64(part);
        if (lane == 0) s_tot = part;
    }
    __syncthreads();
    const float inv = (float)(1.0 / s_tot);
    for (int i = tid; i < n; i += 1024) s_p[i] = (_Float16)(s_s[i] * inv);
    __syncthreads();
    // ---- partial V.P of the range
    {
        const unsigned vpair = (vact && own) ? s_new[128 + (cv >> 1)] : 0u;
        const _Float16 vnew = u32_as_h2(vpair)[cv & 1];
        const int np = n_past - t0;  // the token's position inside the range (if own)
        float acc = 0.0f;
#pragma unroll
        for (int u = 0; u < NPRE; u++) {
            if (64 * u >= n8) break;  // uniform
            const int pos = 64 * u + pj;
            if (pos < n8) {
                f16x8 vr = vv[u];
                if (own && (pos >> 3) == (np >> 3)) {
                    const int e = np - pos;
#pragma unroll
                    for (int j = 0; j < 8; j++) vr[j] = e == j ? vnew : vr[j];
                }
                const f16x8 pp = *(const f16x8 *)(s_p + pos);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc = __builtin_amdgcn_fdot2(f16x2{vr[2 * j], vr[2 * j + 1]}, f16x2{pp[2 * j], pp[2 * j + 1]}, acc, false);
            }
        }
        acc = g8_sum_f32(acc);
        if ((lane & 7) == 0 && vact) gran_store(f.part_g + ((int64_t)h * S + s) * D + cv, tag, __float_as_uint(acc));
    }
    // ---- the last workgroup of the head to arrive adds the partials and re-quantizes for wo
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(f.cnt + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == (unsigned)(S - 1);
        if (s_last) __hip_atomic_store(f.cnt + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    const int nblk = D / 32, l = tid & 31, b = tid >> 5;
    if (b >= nblk) return;
    float v = 0.0f;
    for (int s2 = 0; s2 < S; s2++)
        v += __uint_as_float((unsigned)attn_one_wait(f.part_g + ((int64_t)h * S + s2) * D + b * 32 + l, tag, f.err));
    publish_head_q8<F16_D>(f, h, v, nblk, l, b, epoch);
}

template <int QT, bool INSTR = false>
__global__ void __launch_bounds__(1024) k_qkv_attn(const BigArgs ba, const FusedAttnArgs fa) {
    constexpr bool F16_D = QT == QT_Q4_0 || QT == QT_Q5_0 || QT == QT_Q8_0;
    const int H = fa.n_head, A = H * (fa.S > 1 ? fa.S : 1);  // attention workgroups (dispatched first)
    if ((int)blockIdx.x < A) {
        if (fa.S > 1)
            attn_consumer_split<F16_D>(fa, (int)blockIdx.x % H, (int)blockIdx.x / H);
        else
            attn_consumer<F16_D>(fa, (int)blockIdx.x);
        return;
    }
    big_body<QT, EPI_QKV, XSRC_NORM, INSTR>(ba, (int)blockIdx.x - A, (int)gridDim.x - A);
}
// the WO form: wq|wk|wv + attention + wo + residual in one launch (see wo_tail)
template <int QT, int NBLT>
__global__ void __launch_bounds__(1024) k_qkv_attn_wo(const BigArgs ba, const FusedAttnArgs fa, const WoTailArgs wt) {
    constexpr bool F16_D = QT == QT_Q4_0 || QT == QT_Q5_0 || QT == QT_Q8_0;
    const int H = fa.n_head, A = H * (fa.S > 1 ? fa.S : 1);
    if ((int)blockIdx.x < A) {
        if (fa.S > 1)
            attn_consumer_split<F16_D>(fa, (int)blockIdx.x % H, (int)blockIdx.x / H);
        else
            attn_consumer<F16_D>(fa, (int)blockIdx.x);
        return;
    }
    big_body<QT, EPI_QKV, XSRC_NORM, false>(ba, (int)blockIdx.x - A, (int)gridDim.x - A);
    wo_tail<QT, NBLT>(wt, fa, (int)blockIdx.x - A, (int)gridDim.x - A);
}
