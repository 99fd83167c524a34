// decode_dma.h — the decode mat-vec with the weight stream DECOUPLED from everything else (option "big" = 3, Q4_0):
// one workgroup per CU, 8 waves with fixed roles:
//   wave 0           LOADER: streams this CU's slice of the matrix (rows dealt contiguously per CU) into a 96 KiB LDS ring
//                    with LDS-DMA (global_load_lds_dwordx4, 1 KiB per request, non-temporal), starting at its first
//                    instruction.  It never waits for the activation and never touches a weight byte with the VALU.
//   waves 1-3, 5-7   CONSUMERS: stage the activation (rms_norm / re-quantization to Q8, exactly the arithmetic and the
//                    summation order of k_mmvq_big's staging), then dot the rows out of the ring as they land, group by
//                    group, and run the epilogues (store / +residual / silu*mul / RoPE + K,V store) per group.
//   wave 4           helps staging, then exits: it would share the loader's SIMD, and a lone loader wave streams 6.8 TB/s
//                    while one that shares its SIMD with a busy wave does not reach 4 (tests/tools/dma_rate.hip,
//                    tests/tools/engine_probe.hip; measured on MI355X).
// Why: k_mmvq_big keeps the weights in flight in registers of the waves that also stage x and do the dots, so its
// launches run as phases — requests, staging, arrival, then ~2.7 us of dots/reductions/epilogue during which HBM idles
// (tests/tools/launch_probe.py: the w1|w3 launch takes 9.5 us when it returns once everything is requested and 12.8 us
// in full).  Here the stream runs from entry to its last byte at the rate of a plain copy, the dots of group g overlap
// the arrival of group g+1.., and the launch ends one group's arithmetic after the last byte.
//
// Ring protocol (all inside one workgroup, LDS only): the loader numbers its 1 KiB requests; the requests of a group
// (RS consecutive rows: RS*nbl chunks of 64 blocks, then one chunk with the RS rows' scales; twice that for w1|w3) occupy
// contiguous ring slots (a group that would wrap skips to slot 0).  After every group it READS how many of its requests
// are still outstanding (s_getreg IB_STS.VM_CNT; requests land in order) and publishes `landed`; it waits only to keep
// at most ~50 requests in flight.  Consumer c owns groups c, c+6, ...; all consumers replay the loader's walk (a handful
// of scalar instructions per group), wait for `landed`, and publish in done[c] the first chunk they still need, which is
// what lets the loader reuse slots.  Results are bit-identical to k_mmvq_big: same block_dot, same per-lane block order
// (lane + 64 j), same wave reduction, same staging arithmetic (the 512 staging threads of k_mmvq_big are emulated as 8
// virtual waves by the 7 staging waves).
#pragma once
#include <type_traits>

#include "decode_big.h"

#define DMA_T 512
#define DMA_NC 6
#define DMA_NCH 96
#define DMA_RING_B (DMA_NCH * 1024)

struct DmaCtl {
    unsigned landed;  // requests landed (loader -> consumers)
    unsigned bar;     // staging barrier arrivals (monotonic)
    unsigned pad[2];
    unsigned done[8];  // per consumer: first chunk index it still needs (consumers -> loader)
    double part[8];    // sum of squares of the 8 virtual staging waves (XSRC_NORM)
};
__device__ unsigned g_dma_err;  // set when an intra-workgroup wait gave up (a logic error, never expected): ggml_hip_get_stat("dma_err")
#define DMA_SPIN_LIMIT 400000u  // polls of ~0.1 us: tens of ms, orders of magnitude above any legitimate wait

// geometry of one launch for one CU
struct DmaGeo {
    int nb, nbl, RS, NW, nseg;
    int r0a, r0b, r0c, na, nbr, nc;  // this CU's rows of each segment (QKV: wq, wk, wv; otherwise one segment); scalars,
                                     // selected by compares: a dynamically indexed array would live in scratch memory
    // (a blend by arithmetic: a select between members becomes a select of their ADDRESSES and one load, which pins the
    // struct in scratch memory — the loader must not issue scratch loads, they share its request counter)
    __device__ __forceinline__ int row0(int s) const { return nseg == 1 ? r0a : (s == 0) * r0a + (s == 1) * r0b + (s == 2) * r0c; }
    __device__ __forceinline__ int nrows(int s) const { return nseg == 1 ? na : (s == 0) * na + (s == 1) * nbr + (s == 2) * nc; }
    __device__ __forceinline__ int groups(int s) const { return (nrows(s) + RS - 1) / RS; }
    __device__ __forceinline__ int rows_of(int s, int g) const { const int r = nrows(s) - g * RS; return r < RS ? r : RS; }
    __device__ __forceinline__ unsigned chunks(int rows) const { return (unsigned)(NW * (rows * nbl + 1)); }
};
// rows of a matrix dealt contiguously over the G workgroups, in units of UR rows
__device__ __forceinline__ void dma_slice(int M, int UR, int cu, int G, int &r0, int &n) {
    const int U = M / UR, per = U / G, rem = U % G;
    r0 = (cu * per + (cu < rem ? cu : rem)) * UR;
    n = (per + (cu < rem ? 1 : 0)) * UR;
}
// The loader's walk over the ring, replayed by every consumer.
struct DmaWalk {
    int s, g;                 // segment, group inside it (in front of the walk)
    unsigned k, slot, req, gi;  // chunk index, k mod NCH, requests issued so far, global group index
    __device__ __forceinline__ void init(const DmaGeo &ge) {
        s = 0; g = 0; k = 0; slot = 0; req = 0; gi = 0;
        while (s < ge.nseg && ge.groups(s) == 0) s++;
    }
    __device__ __forceinline__ bool end(const DmaGeo &ge) const { return s >= ge.nseg; }
    __device__ __forceinline__ unsigned k0(const DmaGeo &ge) const {  // first chunk of the group in front
        const unsigned n = ge.chunks(ge.rows_of(s, g));
        return slot + n > DMA_NCH ? k + (DMA_NCH - slot) : k;
    }
    __device__ __forceinline__ void step(const DmaGeo &ge) {
        const unsigned n = ge.chunks(ge.rows_of(s, g));
        if (slot + n > DMA_NCH) { k += DMA_NCH - slot; slot = 0; }
        k += n; slot += n; req += n; gi++;
        if (++g == ge.groups(s)) {
            g = 0;
            do { s++; } while (s < ge.nseg && ge.groups(s) == 0);
        }
    }
};

template <int NBL>
struct DmaXFrag {  // this lane's activation blocks lane, lane + 64, ... in registers for the whole launch
    i32x4 lo[NBL], hi[NBL];
    float d[NBL];
    int s[NBL];
};

__device__ __forceinline__ unsigned long long dma_uniform(const void *p) {  // an "s" operand must be uniform for the compiler too
    const unsigned long long q = (unsigned long long)(uintptr_t)p;
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(q >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)q);
}
// up to 4 consecutive 1 KiB requests: one M0 write, the instruction's immediate offset moves both the global and the LDS
// address (checked on hardware: tests/tools/dma_rate.hip)
__device__ __forceinline__ void dma_req4(unsigned voff, unsigned long long base_, unsigned dst_) {
    const unsigned long long base = dma_uniform((const void *)(uintptr_t)base_);
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)dst_);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:2048 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072 nt"
                 ::"v"(voff), "s"(base), "s"(dst) : "memory");
}
__device__ __forceinline__ void dma_req2(unsigned voff, unsigned long long base_, unsigned dst_) {
    const unsigned long long base = dma_uniform((const void *)(uintptr_t)base_);
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)dst_);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024 nt"
                 ::"v"(voff), "s"(base), "s"(dst) : "memory");
}
__device__ __forceinline__ void dma_req1(unsigned voff, unsigned long long base_, unsigned dst_) {
    const unsigned long long base = dma_uniform((const void *)(uintptr_t)base_);
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)dst_);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(base), "s"(dst) : "memory");
}
// n consecutive KiB from `base` into consecutive slots from `dst`; every lane reads base + voff + i * 1024
__device__ __forceinline__ void dma_run(int n, unsigned voff, unsigned long long &base, unsigned &dst) {
    while (n >= 4) { dma_req4(voff, base, dst); base += 4096; dst += 4096; n -= 4; }
    if (n >= 2) { dma_req2(voff, base, dst); base += 2048; dst += 2048; n -= 2; }
    if (n) { dma_req1(voff, base, dst); base += 1024; dst += 1024; }
}

// The consumer side of k_mmvq_dma for rows of NBL column steps: a function taking its state BY VALUE (a lambda capturing by
// reference made the launch geometry live in scratch memory).
template <int EPI, int NBL>
__device__ __forceinline__ bool dma_consume(const BigArgs ba, const DmaGeo ge, char *smem, DmaCtl *ctl, const i32x4 *s_lo,
                                            const i32x4 *s_hi, const float *s_d, const int *s_sum, const float *s_rope,
                                            const int c, const int lane, const int n_past) {
    constexpr int QT = QT_Q4_0;
    constexpr int NW = EPI == EPI_GATE ? 2 : 1;
    const DecMmvqArgs &a = ba.d;
    const int nb = ge.nb;
    bool dead = false;
    DmaXFrag<NBL> xf;
#pragma unroll
    for (int j = 0; j < NBL; j++) {
        const int b = j * 64 + lane;  // < nbp
        xf.lo[j] = s_lo[b]; xf.hi[j] = s_hi[b]; xf.d[j] = s_d[b]; xf.s[j] = s_sum[b];
    }
    DmaWalk cur, nx;
    cur.init(ge);
    nx.init(ge);
    while (!cur.end(ge) && !dead) {
        if ((int)(cur.gi % DMA_NC) == c) {
            const int s = cur.s, g = cur.g, rows = ge.rows_of(s, g);
            const unsigned n = ge.chunks(rows), k0 = cur.k0(ge), req_end = cur.req + n;
            const int m = ge.row0(s) + g * ge.RS + lane;  // lane r finishes row r of the group
            float res = 0.0f;
            if constexpr (EPI == EPI_ADD) res = a.res[lane < rows ? m : ge.row0(s)];  // in flight while the group lands
            {
                unsigned spins = 0;
                while (__hip_atomic_load(&ctl->landed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < req_end) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > DMA_SPIN_LIMIT) { dead = true; break; }
                }
                if (dead) break;
            }
            const char *gp = smem + (k0 % DMA_NCH) * 1024;
            const char *sc = gp + NW * rows * NBL * 1024;
            float myv[NW];
#pragma unroll
            for (int mtx = 0; mtx < NW; mtx++) myv[mtx] = 0.0f;
            for (int r = 0; r < rows; r += 2) {
                const bool two = r + 1 < rows;
                u32x4 q[NW][2][NBL];
                float dw[NW][2][NBL];
#pragma unroll
                for (int mtx = 0; mtx < NW; mtx++)
#pragma unroll
                    for (int rr = 0; rr < 2; rr++)
#pragma unroll
                        for (int j = 0; j < NBL; j++) {
                            const int ro = r + ((rr == 1 && !two) ? 0 : rr);  // no second row: re-read the first
                            const int b = j * 64 + lane, bc = b < nb ? b : nb - 1;
                            q[mtx][rr][j] = *(const u32x4 *)(gp + ((mtx * rows + ro) * NBL + j) * 1024 + lane * 16);
                            dw[mtx][rr][j] = __half2float(*(const __half *)(sc + mtx * 1024 + (ro * nb + bc) * 2));
                        }
#pragma unroll
                for (int mtx = 0; mtx < NW; mtx++)
#pragma unroll
                    for (int rr = 0; rr < 2; rr++) {
                        float acc = 0.0f;
#pragma unroll
                        for (int j = 0; j < NBL; j++) {
                            const int b = j * 64 + lane;
                            const float t = block_dot<QT>(q[mtx][rr][j], q[mtx][rr][j], 0u, dw[mtx][rr][j], 0.0f, xf.lo[j], xf.hi[j], xf.d[j], xf.s[j]);
                            acc += b < nb ? t : 0.0f;  // past the row end the ring holds other rows' bytes
                        }
                        if (rr == 0 || two) {
                            const float v = wave_sum_f32(acc);
                            myv[mtx] = lane == r + rr ? v : myv[mtx];
                        }
                    }
            }
            // nothing below this wave's next group is needed any more
            while (!nx.end(ge) && nx.gi < cur.gi + DMA_NC) nx.step(ge);
            const unsigned nk = nx.end(ge) ? 0xffffffffu : nx.k0(ge);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&ctl->done[c], nk, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            // ---- epilogue of the group's rows, one lane each ----
            if (ba.probe != 3) {
                if constexpr (EPI == EPI_STORE) {
                    if (lane < rows) a.dst[m] = myv[0];
                } else if constexpr (EPI == EPI_ADD) {
                    if (lane < rows) a.dst[m] = myv[0] + res;
                } else if constexpr (EPI == EPI_GATE) {
                    if (lane < rows) a.dst[m] = silu_table(myv[0]) * myv[1];
                } else {  // EPI_QKV: rows (2p, 2p+1) of a matrix are a RoPE pair; lanes 2p, 2p+1 hold them
                    const float other = dpp_f32<DPP_QUAD_XOR1>(myv[0]);
                    const float v0 = (lane & 1) ? other : myv[0], v1 = (lane & 1) ? myv[0] : other;
                    if (lane < rows) {
                        if (s == 2) {
                            a.mem_v[(int64_t)m * a.C + n_past] = __float2half_rn(myv[0]);
                        } else {
                            const int kk = (m % a.D) >> 1;
                            const float cs = s_rope[2 * kk], sn = s_rope[2 * kk + 1];
                            const float rv = (lane & 1) ? v0 * sn + v1 * cs : v0 * cs - v1 * sn;
                            if (s == 0) a.dst[m] = rv;
                            else a.mem_k[(int64_t)n_past * a.Egqa + m] = __float2half_rn(rv);
                        }
                    }
                }
            }
        }
        cur.step(ge);
    }
    return dead;
}

template <int EPI, int XSRC>
__global__ void __launch_bounds__(DMA_T) k_mmvq_dma(const BigArgs ba) {
    constexpr int QT = QT_Q4_0;
    constexpr bool F16_D = true;
    constexpr int NW = EPI == EPI_GATE ? 2 : 1;
    const DecMmvqArgs &a = ba.d;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cu = blockIdx.x, G = gridDim.x;
    const int nb = (int)a.nb, nbl = (nb + 63) >> 6, nbp = nbl * 64;
    i32x4 *s_lo = (i32x4 *)(smem + DMA_RING_B);
    i32x4 *s_hi = s_lo + nbp;
    float *s_d = (float *)(s_hi + nbp);
    int *s_sum = (int *)(s_d + nbp);
    DmaCtl *ctl = (DmaCtl *)(s_sum + nbp);
    float *s_rope = (float *)(ctl + 1);  // EPI_QKV: (cos, sin) per pair of a head, D <= 256
    const unsigned lds0 = __builtin_amdgcn_groupstaticsize();

    if (ba.probe == 1) return;
    int RS_ = 512 / nb;
    RS_ = RS_ < 1 ? 1 : RS_ > 4 ? 4 : RS_;
    if (EPI == EPI_QKV && RS_ > 1) RS_ &= ~1;  // a RoPE pair stays in one group (the launcher refuses RS = 1 for QKV)
    int r0a = 0, r0b = 0, r0c = 0, na = 0, nbr = 0, nc = 0;
    dma_slice((int)a.w[0].M, EPI == EPI_QKV ? 2 : 1, cu, G, r0a, na);
    if constexpr (EPI == EPI_QKV) {
        dma_slice((int)a.w[1].M, 2, cu, G, r0b, nbr);
        dma_slice((int)a.w[2].M, 2, cu, G, r0c, nc);
    }
    const DmaGeo ge = {nb, nbl, RS_, NW, EPI == EPI_QKV ? 3 : 1, r0a, r0b, r0c, na, nbr, nc};
    if (tid < (int)(sizeof(DmaCtl) / 4)) ((unsigned *)ctl)[tid] = 0;
    __syncthreads();

    if (wave == 0) {
        // ================================ LOADER ================================
        unsigned k = 0, slot = 0, req = 0, min_done = 0;
        const unsigned v_lane16 = (unsigned)lane * 16u;
        const int last = (nbl - 1) * 64 + lane;
        const unsigned v_last16 = (unsigned)((last < nb ? last : nb - 1) - (nbl - 1) * 64) * 16u;  // last column step of a row
        const bool aligned = (nb & 63) == 0;  // rows are whole chunks: a group's rows are one contiguous run
        auto read_min = [&]() {
            unsigned m = 0xffffffffu;
#pragma unroll
            for (int c = 0; c < DMA_NC; c++) {
                const unsigned v = __hip_atomic_load(&ctl->done[c], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                m = v < m ? v : m;
            }
            return (unsigned)__builtin_amdgcn_readfirstlane((int)m);  // uniform for the compiler: the loop's control flow
        };                                                            // (and with it the "s" operands of the requests) depends on it
        auto publish = [&](unsigned outstanding) {
            __hip_atomic_store(&ctl->landed, req - outstanding, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        bool dead = false;
        for (int s = 0; s < ge.nseg && !dead; s++) {
            const int ng = __builtin_amdgcn_readfirstlane(ge.groups(s));
            unsigned long long qb[2], db[2];
#pragma unroll
            for (int m = 0; m < NW; m++) {
                const int wi = EPI == EPI_GATE ? m : s;  // scalar selects: a kernarg array indexed by a variable is read with vector loads
                const uint8_t *wq = wi == 0 ? a.w[0].qs : wi == 1 ? a.w[1].qs : a.w[2].qs;
                const __half *wdp = wi == 0 ? a.w[0].d : wi == 1 ? a.w[1].d : a.w[2].d;
                qb[m] = dma_uniform(wq + (size_t)ge.row0(s) * nb * 16);
                db[m] = dma_uniform(wdp + (size_t)ge.row0(s) * nb);
            }
            for (int g = 0; g < ng && !dead; g++) {
                const int rows = __builtin_amdgcn_readfirstlane(ge.rows_of(s, g));
                const unsigned n = ge.chunks(rows);
                if (slot + n > DMA_NCH) { k += DMA_NCH - slot; slot = 0; }
                if (k + n > min_done + DMA_NCH) min_done = read_min();
                if (k + n > min_done + DMA_NCH) {  // ring full: everything in flight lands meanwhile
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    publish(0);
                    unsigned spins = 0;
                    for (;;) {
                        min_done = read_min();
                        if (k + n <= min_done + DMA_NCH) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > DMA_SPIN_LIMIT) { dead = true; break; }
                    }
                    if (dead) break;
                }
                unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + slot * 1024u);
#pragma unroll
                for (int m = 0; m < NW; m++) {
                    if (aligned) {
                        dma_run(rows * nbl, v_lane16, qb[m], dst);
                    } else {
                        for (int r = 0; r < rows; r++) {
                            unsigned long long b = qb[m];
                            if (nbl > 1) dma_run(nbl - 1, v_lane16, b, dst);
                            dma_req1(v_last16, b, dst);
                            dst += 1024;
                            qb[m] += (unsigned)nb * 16u;
                        }
                    }
                }
#pragma unroll
                for (int m = 0; m < NW; m++) {  // the rows' scales: rows * nb f16, contiguous, 8 per lane
                    const int n16 = (rows * nb) >> 3;
                    if (lane < n16) dma_req1(v_lane16, db[m], dst);
                    dst += 1024;
                    db[m] += (unsigned)(rows * nb * 2);
                }
                k += n; slot += n; req += n;
                unsigned vm_lo, vm_hi;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_IB_STS, 0, 4)\n\ts_getreg_b32 %1, hwreg(HW_REG_IB_STS, 22, 2)" : "=s"(vm_lo), "=s"(vm_hi));
                unsigned vm = vm_lo | (vm_hi << 4);
                if (vm > 40) {  // a group adds up to 18 requests and the counter has 6 bits
                    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                    vm = 32;
                }
                publish(vm);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish(0);
        if (dead && lane == 0) atomicOr(&g_dma_err, 1u);
        return;
    }

    // ================================ STAGING (waves 1..7) ================================
    const int st = wave - 1;  // 0..6
    const int c = wave < 4 ? wave - 1 : wave - 2;  // consumer index 0..5 (wave 4 has none)
    bool dead = false;
    if (wave != 4) {
        // what this consumer needs first: the loader may recycle everything below it.  (Left at 0, a consumer whose first
        // group lies beyond the first trip around the ring would stop the loader before it ever requests that group.)
        DmaWalk w0;
        w0.init(ge);
        while (!w0.end(ge) && (int)(w0.gi % DMA_NC) != c) w0.step(ge);
        const unsigned first = w0.end(ge) ? 0xffffffffu : w0.k0(ge);
        if (lane == 0) __hip_atomic_store(&ctl->done[c], first, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    unsigned bar_target = 0;
    auto sbarrier = [&]() {  // among the 7 staging waves
        bar_target += 7;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&ctl->bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned spins = 0;
        while (__hip_atomic_load(&ctl->bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < bar_target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > DMA_SPIN_LIMIT) { dead = true; break; }
        }
    };
    int n_past = 0;
    if constexpr (EPI == EPI_QKV) {
        n_past = a.prm->n_past;
        const int t = st * 64 + lane;
        if (t < (a.D >> 1)) {
            const f32x2 cs = ((const f32x2 *)ba.rope)[t];
            s_rope[2 * t] = cs[0];
            s_rope[2 * t + 1] = cs[1];
        }
    }
    for (int i = nb + st * 64 + lane; i < nbp; i += 7 * 64) {  // padded blocks: zero, so that tail lanes contribute 0
        s_lo[i] = i32x4{0, 0, 0, 0};
        s_hi[i] = i32x4{0, 0, 0, 0};
        s_d[i] = 0.0f;
        s_sum[i] = 0;
    }
    const int n4 = nb * 8;
    if constexpr (XSRC == XSRC_Q8) {
        for (int i = st * 64 + lane; i < nb; i += 7 * 64) {
            s_lo[i] = a.x.lo[i];
            s_hi[i] = a.x.hi[i];
            s_d[i] = a.x.d[i];
            s_sum[i] = a.x.sum[i];
        }
    } else if constexpr (XSRC == XSRC_F32) {
        constexpr int MAXIT = 14;  // rows up to 14 * 448 * 4 = 25088 wide (checked by the launcher)
        f32x4 v[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; it++) {
            const int i4 = it * 448 + st * 64 + lane;
            v[it] = ((const f32x4 *)a.xf)[i4 < n4 ? i4 : 0];
        }
#pragma unroll
        for (int it = 0; it < MAXIT; it++) {
            const int i4 = it * 448 + st * 64 + lane;
            if (it * 448 >= n4) break;  // uniform
            const f32x4 y = i4 < n4 ? v[it] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            quant4_to_lds<F16_D>(y, i4, nb, lane, s_lo, s_hi, s_d, s_sum);
        }
    } else {
        // rms_norm: k_mmvq_big stages with 512 threads = 8 waves; here virtual wave vw = st, and staging wave 0 also
        // plays virtual wave 7 — same per-thread element sets, same f64 orders, same s_part order: the same bits
        constexpr int MAXIT = BigX<XSRC_NORM>::MAXIT, NT = BigX<XSRC_NORM>::NT;
        f32x4 v[2][MAXIT], wv[2][MAXIT];
        const int nv = st == 0 ? 2 : 1;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int tv = (q == 0 ? st : 7) * 64 + lane;
#pragma unroll
            for (int it = 0; it < MAXIT; it++) {
                const int i4 = it * NT + tv;
                const int ic = (q < nv && i4 < n4) ? i4 : 0;
                v[q][it] = ((const f32x4 *)a.xf)[ic];
                wv[q][it] = ((const f32x4 *)a.xw)[ic];
            }
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (q < nv) {
                const int tv = (q == 0 ? st : 7) * 64 + lane;
                double ss = 0.0;
#pragma unroll
                for (int it = 0; it < MAXIT; it++) {
                    const int i4 = it * NT + tv;
                    if (i4 < n4) {
                        ss += (double)(v[q][it][0] * v[q][it][0]);
                        ss += (double)(v[q][it][1] * v[q][it][1]);
                        ss += (double)(v[q][it][2] * v[q][it][2]);
                        ss += (double)(v[q][it][3] * v[q][it][3]);
                    }
                }
                ss = wave_sum_f64(ss);
                if (lane == 0) ctl->part[q == 0 ? st : 7] = ss;
            }
        }
        sbarrier();
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < 8; i++) tot += ctl->part[i];
        const float mean = (float)(tot / (double)(nb * 32));
        const float scale = 1.0f / sqrtf(mean + a.eps);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (q < nv) {
                const int tv = (q == 0 ? st : 7) * 64 + lane;
#pragma unroll
                for (int it = 0; it < MAXIT; it++) {
                    const int i4 = it * NT + tv;
                    if (it * NT >= n4) break;  // uniform
                    f32x4 y = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (i4 < n4) {
                        y[0] = (v[q][it][0] * scale) * wv[q][it][0];
                        y[1] = (v[q][it][1] * scale) * wv[q][it][1];
                        y[2] = (v[q][it][2] * scale) * wv[q][it][2];
                        y[3] = (v[q][it][3] * scale) * wv[q][it][3];
                        if (ba.y_out && blockIdx.x == 0) ((f32x4 *)ba.y_out)[i4] = y;
                    }
                    quant4_to_lds<F16_D>(y, i4, nb, lane, s_lo, s_hi, s_d, s_sum);
                }
            }
        }
    }
    sbarrier();
    if (wave == 4 || ba.probe == 2 || dead) {
        if (dead && lane == 0) atomicOr(&g_dma_err, 2u);
        if (wave != 4 && lane == 0) __hip_atomic_store(&ctl->done[c], 0xffffffffu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }

    // ================================ CONSUMERS ================================
    switch (nbl) {  // the row lengths of the LLaMA family (7B: 2 and 6; 13B: 3 and 7; 65B: 4); the launcher refuses others
        case 2: dead = dma_consume<EPI, 2>(ba, ge, smem, ctl, s_lo, s_hi, s_d, s_sum, s_rope, c, lane, n_past); break;
        case 3: dead = dma_consume<EPI, 3>(ba, ge, smem, ctl, s_lo, s_hi, s_d, s_sum, s_rope, c, lane, n_past); break;
        case 4: dead = dma_consume<EPI, 4>(ba, ge, smem, ctl, s_lo, s_hi, s_d, s_sum, s_rope, c, lane, n_past); break;
        case 6: dead = dma_consume<EPI, 6>(ba, ge, smem, ctl, s_lo, s_hi, s_d, s_sum, s_rope, c, lane, n_past); break;
        default: dead = dma_consume<EPI, 7>(ba, ge, smem, ctl, s_lo, s_hi, s_d, s_sum, s_rope, c, lane, n_past); break;
    }
    if (dead) {
        if (lane == 0) {
            atomicOr(&g_dma_err, 4u);
            __hip_atomic_store(&ctl->done[c], 0xffffffffu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}
