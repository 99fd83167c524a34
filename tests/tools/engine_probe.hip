// engine_probe.hip — stand-alone prototype of the persistent decode engine (measurement tool, not product code):
// ONE launch walks a chain of quantized mat-vecs (Q4_0 weights x Q8 activations, the arithmetic of kernels/mmvq.h);
// per CU one LOADER wave streams that CU's slice of every matrix into an LDS ring by LDS-DMA (never waits for an
// activation), NC CONSUMER waves dot the rows out of the ring, publish each output element as an 8-byte
// {tag, value} granule, and gather the next activation from all 256 CUs' granules (no kernel boundary, no grid
// barrier).  Prints: us per op, the gather latency, the streaming rate, and checks every intermediate vector
// bit-for-bit against the same chain run as ordinary launches.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I. -o /tmp/engine_probe tests/tools/engine_probe.hip
//   /tmp/engine_probe [chain] [nops] [NC] [NL]     chain: wo | ffn | layer; NC consumer + NL loader waves per CU
#include "../../llm_amd/csrc/kernels/decode.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define ACQ_WG __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP
#define REL_WG __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP
#define RLX_WG __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP

constexpr int NCH = 96;          // ring chunks of 1 KiB
constexpr int RING_B = NCH * 1024;
constexpr int NBP_MAX = 384;     // padded blocks of the widest activation (11008 / 32 = 344 -> 384)
constexpr int XQ_B = NBP_MAX * 40;
constexpr int CTL_OFF = RING_B + 2 * XQ_B;  // the activation is double-buffered by op parity (see the consumers)
constexpr int OPS_OFF = CTL_OFF + 512;
constexpr int MAX_OPS = 256;
constexpr int LDS_B = OPS_OFF + MAX_OPS * 24;  // op descriptors live in LDS: a global load of one in the loader's loop
                                               // would make hipcc wait vmcnt(0) = drain the DMA pipe at every op
constexpr unsigned SPIN_LIMIT = 4000000;

struct Op {
    int K, M;            // input width, output rows
    long long qs_off;    // byte offset of this matrix's qs plane in the weight buffer (16 B per block)
    long long d_off;     // element offset of its d plane (f16 per block)
};
struct Args {
    const uint8_t *wqs;
    const __half *wd;
    const Op *ops;
    int nops;
    u64 *gran;           // [2][GMAX] granules, ping-pong by op parity
    int gmax;
    const float *x0;     // input of op 0 (f32, K0 wide)
    float *vecs;         // [nops][GMAX] plain copy of every op's output (for the check)
    unsigned *err;
    long long *ts;       // [nops][4] wall clock of workgroup 0: gather start, x staged, dots done, published
    long long *lts;      // [G][4] loader: start, end, stall polls, chunks
    unsigned epoch0;
    int mode;            // 0 = full chain; 1 = stream only (consumers free chunks without computing)
};

struct Ctl {
    unsigned landed[4];  // per loader: how many of ITS groups have landed in the ring (loader -> consumers)
    unsigned err;
    unsigned bar;        // arrivals at the consumers' barrier (monotonic)
    unsigned pad[2];
    unsigned done[12];   // per consumer: first chunk index it still needs (consumers -> loaders)
    unsigned fifo[4][16];  // per loader: cumulative DMA-instruction count at the end of each group in flight
};

// chunk bookkeeping of one op for one CU: rows [r0, r0 + nrows) in groups of RS rows; a group = its rows' qs chunks
// (nbl each) followed by ONE scale chunk (the scales of RS consecutive rows are contiguous: RS * nb * 2 <= 1024 B)
struct OpGeo {
    int nb, nbl, RS, r0, nrows, ng, cpg;  // cpg: chunks of a full group
    __device__ __forceinline__ void init(const Op &o, int cu, int ncu) {
        nb = o.K >> 5;
        nbl = (nb + 63) >> 6;
        RS = nb <= 512 ? (512 / nb > 0 ? 512 / nb : 1) : 1;
        if (RS > 8) RS = 8;
        const int per = o.M / ncu, rem = o.M % ncu;  // rows dealt contiguously, the first `rem` CUs get one more
        r0 = cu * per + (cu < rem ? cu : rem);
        nrows = per + (cu < rem ? 1 : 0);
        ng = (nrows + RS - 1) / RS;
        cpg = RS * nbl + 1;
    }
    __device__ __forceinline__ int rows_of(int g) const { return g == ng - 1 ? nrows - g * RS : RS; }
    __device__ __forceinline__ int chunks_of(int g) const { return rows_of(g) * nbl + 1; }
    __device__ __forceinline__ int total() const { return ng == 0 ? 0 : (ng - 1) * cpg + chunks_of(ng - 1); }
};
// Walks the op list forward: groups are numbered globally (gg) across ops, chunks likewise (k).
struct Cursor {
    int oi;
    unsigned gg0, kbase;  // first global group / first chunk of op oi
    OpGeo ge;
    __device__ __forceinline__ void start(const Op *ops, int nops, int cu, int ncu) {
        oi = 0; gg0 = 0; kbase = 0;
        if (nops > 0) ge.init(ops[0], cu, ncu);
    }
    // positions the cursor on the op that holds global group gg; false past the end
    __device__ __forceinline__ bool seek(const Op *ops, int nops, int cu, int ncu, unsigned gg) {
        while (oi < nops && gg >= gg0 + (unsigned)ge.ng) {
            gg0 += (unsigned)ge.ng;
            kbase += (unsigned)ge.total();
            oi++;
            if (oi < nops) ge.init(ops[oi], cu, ncu);
        }
        return oi < nops;
    }
    __device__ __forceinline__ unsigned k0(unsigned gg) const { return kbase + (gg - gg0) * (unsigned)ge.cpg; }
};

template <int NBL>
struct XFrag {  // this lane's activation blocks lane, lane + 64, ... held in registers for the whole op
    i32x4 lo[NBL], hi[NBL];
    float d[NBL];
    int s[NBL];
};

template <int NC, int NL>
__global__ void __launch_bounds__((NC + NL) * 64) k_engine(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctl *ctl = (Ctl *)(smem + CTL_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cu = blockIdx.x, ncu = gridDim.x;
    const unsigned lds0 = __builtin_amdgcn_groupstaticsize();  // LDS address of smem[0]

    const Op *s_ops = (const Op *)(smem + OPS_OFF);
    for (int i = tid; i < a.nops * 6; i += blockDim.x) ((int *)(smem + OPS_OFF))[i] = ((const int *)a.ops)[i];
    for (int i = tid; i < (int)(sizeof(Ctl) / 4); i += blockDim.x) ((unsigned *)ctl)[i] = 0;
    __syncthreads();

    if (wave >= NC) {
        // ================================ LOADER ================================
        // One wave (NL must be 1 in this version) whose whole job is to keep LDS-DMA requests in flight, so the loop is a
        // handful of scalar instructions per 1 KiB request: a row is one asm statement (M0 = ring slot, the row's address in
        // a scalar register pair, every lane a constant 32-bit offset: lane * 16, clamped to the row's last block in the last
        // column step), a group's ring slots are contiguous (a group that would wrap skips to slot 0: consumers apply the
        // same rule), and the number of requests still outstanding is READ (s_getreg IB_STS.VM_CNT), not waited for, so
        // what has landed is published at once.
        const long long t_start = wall_clock64();
        unsigned issued = 0 /* chunk index incl. skipped slots */, slot = 0, min_done = 0, req = 0 /* requests issued */;
        long long stalls = 0;
        bool dead = false;
        const unsigned v_lane16 = (unsigned)lane * 16u;
        auto read_min = [&]() {
            unsigned m = 0xffffffffu;
#pragma unroll
            for (int w = 0; w < NC; w++) {
                const unsigned v = __hip_atomic_load(&ctl->done[w], ACQ_WG);
                m = v < m ? v : m;
            }
            return m;
        };
        // chunk index up to which everything has landed: requests land in order, `outstanding` of them are in flight,
        // and request number -> chunk index is kept for the last 64 requests in an LDS table (skips make it non-trivial)
        unsigned *rq2k = ctl->fifo[0];  // [64]: chunk index AFTER request r (r mod 64)
        auto publish = [&](unsigned outstanding) {
            const unsigned r = req - outstanding;  // requests landed
            if (r == 0) return;
            const unsigned k = rq2k[(r - 1) & 63];
            __hip_atomic_store(&ctl->landed[0], k, REL_WG);
        };
        for (int oi = 0; oi < a.nops && !dead; oi++) {
            const Op o = s_ops[oi];
            OpGeo ge;
            ge.init(o, cu, ncu);
            const int last = (ge.nbl - 1) * 64 + lane;
            // second request of a pair: + 1024 relative to the pair's base; in the row's last step clamped to its last block
            const unsigned v_lane16p = v_lane16 + 1024u;
            const unsigned v_last16p = (unsigned)((last < ge.nb ? last : ge.nb - 1) - (ge.nbl - 1) * 64) * 16u + 1024u;
            const unsigned long long q0 = (unsigned long long)(uintptr_t)(a.wqs + o.qs_off + (size_t)ge.r0 * ge.nb * 16);
            unsigned long long base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(q0 >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)q0);
            const unsigned long long d0 = (unsigned long long)(uintptr_t)(a.wd + o.d_off + (size_t)ge.r0 * ge.nb);
            unsigned long long dbase = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(d0 >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)d0);
            const unsigned row_bytes = (unsigned)ge.nb * 16u;
            for (int g = 0; g < ge.ng && !dead; g++) {
                const int rows = ge.rows_of(g);
                const unsigned nchunks = (unsigned)(rows * ge.nbl + 1);
                if (slot + nchunks > NCH) {  // the group's slots are contiguous: skip the tail of the ring
                    issued += NCH - slot;
                    slot = 0;
                }
                if (issued + nchunks > min_done + NCH) min_done = read_min();
                if (issued + nchunks > min_done + NCH) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // blocked anyway: everything in flight lands
                    publish(0);
                    unsigned spins = 0;
                    for (;;) {
                        min_done = read_min();
                        if (issued + nchunks <= min_done + NCH) break;
                        stalls++;
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > SPIN_LIMIT || __hip_atomic_load(&ctl->err, RLX_WG)) { dead = true; break; }
                    }
                    if (dead) break;
                }
                unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + slot * 1024u);
                for (int r = 0; r < rows; r++) {
                    // column steps in pairs (nbl is even for the probe's shapes): requests j and j + 1 of the row
                    unsigned long long b2 = base;
                    for (int j = 0; j < ge.nbl; j += 2) {
                        const unsigned vb = j + 2 == ge.nbl ? v_last16p : v_lane16p;
                        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 nt\n\t"
                                     "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt"
                                     ::"v"(v_lane16), "v"(vb), "s"(b2), "s"(dst) : "memory");
                        b2 += 2048;
                        dst += 2048;
                    }
                    base += row_bytes;
                }
                {  // the group's scales: rows * nb f16, contiguous; 8 per lane
                    const int n16 = (rows * ge.nb) >> 3;
                    if (lane < n16)
                        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(v_lane16), "s"(dbase), "s"(dst) : "memory");
                    dbase += (unsigned)(rows * ge.nb * 2);
                }
                // request -> chunk table (one entry per request; only the group's last one is ever looked up precisely,
                // the others point at the group start so that a partially landed group is not published)
                for (unsigned i = lane; i < nchunks; i += 64) rq2k[(req + i) & 63] = i == nchunks - 1 ? issued + nchunks : issued;
                req += nchunks;
                issued += nchunks;
                slot += nchunks;
                unsigned vm_lo, vm_hi;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_IB_STS, 0, 4)\n\ts_getreg_b32 %1, hwreg(HW_REG_IB_STS, 22, 2)" : "=s"(vm_lo), "=s"(vm_hi));
                unsigned vm = vm_lo | (vm_hi << 4);
                if (vm > 44) {
                    asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
                    vm = 36;
                }
                publish(vm);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish(0);
        if (dead) {
            __hip_atomic_store(&ctl->err, 1u, RLX_WG);
            if (lane == 0) atomicOr(a.err, 1u);
        }
        if (lane == 0) {
            a.lts[cu * 4 + 0] = t_start;
            a.lts[cu * 4 + 1] = wall_clock64();
            a.lts[cu * 4 + 2] = stalls;
            a.lts[cu * 4 + 3] = req;
        }
        return;
    }

    // ================================ CONSUMERS ================================
    const int w = wave;  // 0..NC-1
    unsigned bar_target = 0;
    bool dead = false;
    auto cbarrier = [&]() {  // barrier among the NC consumer waves (the loader never joins one)
        bar_target += NC;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&ctl->bar, 1u, REL_WG);
        unsigned spins = 0;
        while (__hip_atomic_load(&ctl->bar, ACQ_WG) < bar_target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT || __hip_atomic_load(&ctl->err, RLX_WG)) { dead = true; break; }
        }
    };
    // The loader's ring walk, replayed: position in front of group (oi, g); k = chunk index, slot = k mod NCH.
    struct Walk {
        int oi, g;
        unsigned k, slot, gg;
        OpGeo ge;
    };
    auto walk_init = [&](Walk &x) {
        x.oi = 0; x.g = 0; x.k = 0; x.slot = 0; x.gg = 0;
        x.ge.init(s_ops[0], cu, ncu);
        while (x.oi < a.nops && x.ge.ng == 0) { x.oi++; if (x.oi < a.nops) x.ge.init(s_ops[x.oi], cu, ncu); }
    };
    auto walk_k0 = [&](const Walk &x) -> unsigned {  // first chunk of the group in front (after the wrap skip)
        const unsigned n = (unsigned)x.ge.chunks_of(x.g);
        return x.slot + n > NCH ? x.k + (NCH - x.slot) : x.k;
    };
    auto walk_step = [&](Walk &x) {  // pass the group in front
        const unsigned n = (unsigned)x.ge.chunks_of(x.g);
        if (x.slot + n > NCH) { x.k += NCH - x.slot; x.slot = 0; }
        x.k += n; x.slot += n; x.gg++;
        if (++x.g == x.ge.ng) {
            x.g = 0;
            do { x.oi++; if (x.oi < a.nops) x.ge.init(s_ops[x.oi], cu, ncu); } while (x.oi < a.nops && x.ge.ng == 0);
        }
    };
    Walk cur, nx;
    walk_init(cur);
    walk_init(nx);
    // rows of one group out of the ring: rows in pairs, all LDS reads of a pair issued before its arithmetic
    auto group_rows = [&](auto nbl_tag, const OpGeo &ge, unsigned slot0, int rows, const auto &xf) -> float {
        constexpr int NBL = decltype(nbl_tag)::value;
        const int nb = ge.nb;
        float myv = 0.0f;
        const char *sc = smem + (slot0 + (unsigned)(rows * NBL)) * 1024;  // the group's slots are contiguous
        const char *rowp = smem + slot0 * 1024;
        for (int r = 0; r < rows; r += 2) {
            const bool two = r + 1 < rows;
            u32x4 q[2][NBL];
            float dw[2][NBL];
#pragma unroll
            for (int rr = 0; rr < 2; rr++)
#pragma unroll
                for (int j = 0; j < NBL; j++) {
                    const int ro = (rr == 1 && !two) ? 0 : rr;  // no second row: re-read the first, result unused
                    const int b = j * 64 + lane, bc = b < nb ? b : nb - 1;
                    q[rr][j] = *(const u32x4 *)(rowp + (ro * NBL + j) * 1024 + lane * 16);
                    dw[rr][j] = __half2float(*(const __half *)(sc + ((r + ro) * nb + bc) * 2));
                }
            float acc[2] = {0.0f, 0.0f};
#pragma unroll
            for (int rr = 0; rr < 2; rr++)
#pragma unroll
                for (int j = 0; j < NBL; j++) {
                    const int b = j * 64 + lane;
                    const float t = block_dot<QT_Q4_0>(q[rr][j], q[rr][j], 0u, dw[rr][j], 0.0f, xf.lo[j], xf.hi[j], xf.d[j], xf.s[j]);
                    acc[rr] += b < nb ? t : 0.0f;  // the weight bytes past the row end are not this row's
                }
            const float v0 = wave_sum_f32(acc[0]);
            myv = lane == r ? v0 : myv;
            if (two) {
                const float v1 = wave_sum_f32(acc[1]);
                myv = lane == r + 1 ? v1 : myv;
            }
            rowp += 2 * NBL * 1024;
        }
        return myv;
    };
    for (int oi = 0; oi < a.nops; oi++) {
        const Op o = s_ops[oi];
        OpGeo ge;
        ge.init(o, cu, ncu);
        const int nb = ge.nb, nbp = ge.nbl * 64;
        // The Q8 activation of op oi lives in buffer oi & 1: a wave that is already gathering for op oi + 1 writes the
        // other buffer while a slower wave may still be loading its fragments of op oi; the gather barrier of op oi + 1
        // is passed by every wave before anyone writes buffer oi & 1 again.
        i32x4 *s_lo = (i32x4 *)(smem + RING_B + (oi & 1) * XQ_B);
        i32x4 *s_hi = s_lo + NBP_MAX;
        float *s_d = (float *)(s_hi + NBP_MAX);
        int *s_sum = (int *)(s_d + NBP_MAX);
        const unsigned tag = a.epoch0 + (unsigned)oi;
        const bool rec = cu == 0 && w == 0 && lane == 0;
        if (rec) a.ts[oi * 4 + 0] = wall_clock64();
        // ---- 1. the activation: gather the previous op's granules (or x0), re-quantize to Q8 into LDS ----
        if (a.mode == 0) {
            for (int i = nb + w * 64 + lane; i < nbp; i += NC * 64) {  // zero the padded blocks
                s_lo[i] = i32x4{0, 0, 0, 0};
                s_hi[i] = i32x4{0, 0, 0, 0};
                s_d[i] = 0.0f;
                s_sum[i] = 0;
            }
            const int npass = (o.K + 255) >> 8;  // a pass = 64 lanes x 4 consecutive elements = 8 blocks
            const u64 *src = a.gran + (size_t)((oi + 1) & 1) * a.gmax;  // written by op oi-1 with tag - 1
            if (oi == 0) {  // the chain's input is a plain f32 row
                for (int p = w; p < npass; p += NC) {
                    const int e = (p * 64 + lane) * 4;
                    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (e < o.K) v = *(const f32x4 *)(a.x0 + e);
                    quant4_to_lds<true>(v, (int64_t)p * 64 + lane, nb, lane, s_lo, s_hi, s_d, s_sum);
                }
            } else {
                for (int p0 = w; p0 < npass && !dead; p0 += NC * 4) {
                    // up to 4 passes of this wave per batch (p0, p0+NC, p0+2NC, p0+3NC): 16 granule loads per lane in
                    // flight, re-read until every tag of the batch carries the producer op's tag
                    u64 gr[4][4];
                    unsigned spins = 0;
                    for (;;) {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int p = p0 + q * NC;
                            int e = (p * 64 + lane) * 4;
                            e = (p < npass && e < o.K) ? e : 0;  // clamped: the loads are unconditional
#pragma unroll
                            for (int k = 0; k < 4; k++) gr[q][k] = __hip_atomic_load(src + e + k, RLX_AGENT);
                        }
                        bool ok = true;
#pragma unroll
                        for (int q = 0; q < 4; q++)
#pragma unroll
                            for (int k = 0; k < 4; k++) ok &= (unsigned)(gr[q][k] >> 32) == tag - 1;
                        if (__all(ok)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > SPIN_LIMIT || __hip_atomic_load(&ctl->err, RLX_WG)) { dead = true; break; }
                    }
                    if (dead) break;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int p = p0 + q * NC;
                        const int e = (p * 64 + lane) * 4;
                        f32x4 v;
#pragma unroll
                        for (int k = 0; k < 4; k++) v[k] = (p < npass && e < o.K) ? __builtin_bit_cast(float, (unsigned)gr[q][k]) : 0.0f;
                        if (p < npass) quant4_to_lds<true>(v, (int64_t)p * 64 + lane, nb, lane, s_lo, s_hi, s_d, s_sum);
                    }
                }
            }
            cbarrier();
        }
        if (rec) a.ts[oi * 4 + 1] = wall_clock64();
        // ---- 2. this wave's row groups of this op ----
        u64 *dstg = a.gran + (size_t)(oi & 1) * a.gmax;
        auto run_groups = [&](auto nbl_tag) {
            constexpr int NBL = decltype(nbl_tag)::value;
            XFrag<NBL> xf;
            if (a.mode == 0) {
#pragma unroll
                for (int j = 0; j < NBL; j++) {
                    const int b = j * 64 + lane;  // < nbp: padded blocks are zero
                    xf.lo[j] = s_lo[b]; xf.hi[j] = s_hi[b]; xf.d[j] = s_d[b]; xf.s[j] = s_sum[b];
                }
            }
            while (cur.oi == oi && !dead) {
                const int g = cur.g;
                const unsigned gg = cur.gg, k0 = walk_k0(cur), n = (unsigned)ge.chunks_of(g);
                const int rows = ge.rows_of(g);
                if (gg % NC == (unsigned)w) {
                    {  // wait until the group has landed
                        unsigned spins = 0;
                        while (__hip_atomic_load(&ctl->landed[0], ACQ_WG) < k0 + n) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > SPIN_LIMIT || __hip_atomic_load(&ctl->err, RLX_WG)) { dead = true; break; }
                        }
                        if (dead) break;
                    }
                    float myv = 0.0f;
                    if (a.mode == 0) myv = group_rows(nbl_tag, ge, k0 % NCH, rows, xf);
                    // this wave needs nothing below its next group (gg + NC) any more
                    while (nx.oi < a.nops && nx.gg < gg + NC) walk_step(nx);
                    const unsigned nk = nx.oi < a.nops ? walk_k0(nx) : 0xffffffffu;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&ctl->done[w], nk, REL_WG);
                    // publish: lane r holds row r of the group
                    if (a.mode == 0 && lane < rows) {
                        const int m = ge.r0 + g * ge.RS + lane;
                        __hip_atomic_store(dstg + m, ((u64)tag << 32) | (u64)__builtin_bit_cast(unsigned, myv), RLX_AGENT);
                        a.vecs[(size_t)oi * a.gmax + m] = myv;
                    }
                }
                walk_step(cur);
            }
        };
        switch (ge.nbl) {
            case 2: run_groups(std::integral_constant<int, 2>{}); break;
            case 6: run_groups(std::integral_constant<int, 6>{}); break;
            default: dead = true; break;  // probe: only the 7B row widths
        }
        if (rec) a.ts[oi * 4 + 2] = wall_clock64();
        if (dead) break;
        if (rec) a.ts[oi * 4 + 3] = wall_clock64();
    }
    if (dead) {
        __hip_atomic_store(&ctl->err, 1u, RLX_WG);
        if (lane == 0) atomicOr(a.err, 2u);
        if (lane == 0) __hip_atomic_store(&ctl->done[w], 0xffffffffu, REL_WG);  // unblock the loader
    }
}

// ---- the same chain as ordinary launches (reference for the bit-exact check and for the launch-based time) ----
__global__ void __launch_bounds__(1024) k_ref_quant(const float *x, int K, int8_t *lo, int8_t *hi, float *dq, int *sumq) {
    // same arithmetic as quant4_to_lds: 8 lanes per block, 4 consecutive elements per lane
    const int t = blockIdx.x * 1024 + threadIdx.x;
    const int nb = K >> 5;
    if (t * 4 >= ((K + 255) & ~255)) return;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (t * 4 < K) v = *(const f32x4 *)(x + t * 4);
    float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    amax = g8_max_f32(amax);
    const float d = amax / 127.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    const int q0 = (int)roundf(v[0] * id), q1 = (int)roundf(v[1] * id), q2 = (int)roundf(v[2] * id), q3 = (int)roundf(v[3] * id);
    int sq = (q0 + q1) + (q2 + q3);
    sq = g8_sum_i32(sq);
    const int b = t >> 3, j = t & 7;
    if (b >= nb) return;
    const int packed = (q0 & 0xFF) | ((q1 & 0xFF) << 8) | ((q2 & 0xFF) << 16) | ((int)((unsigned)q3 << 24));
    ((int *)(j < 4 ? lo : hi))[b * 4 + (j & 3)] = packed;
    if (j == 0) {
        dq[b] = round_f16(d);
        sumq[b] = sq;
    }
}
__global__ void __launch_bounds__(256) k_ref_mv(const uint8_t *qs, const __half *wd, int K, int M, const i32x4 *lo,
                                                const i32x4 *hi, const float *dq, const int *sumq, float *out) {
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int nb = K >> 5;
    float acc = 0.0f;
    for (int b = lane; b < nb; b += 64) {
        const u32x4 q = __builtin_nontemporal_load((const u32x4 *)(qs + ((size_t)m * nb + b) * 16));
        const float dw = __half2float(wd[(size_t)m * nb + b]);
        acc += block_dot<QT_Q4_0>(q, q, 0u, dw, 0.0f, lo[b], hi[b], dq[b], sumq[b]);
    }
    const float v = wave_sum_f32(acc);
    if (lane == 0) out[m] = v;
}
__global__ void k_fill(uint8_t *qs, __half *wd, size_t nblk, unsigned seed) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblk) return;
    unsigned s = (unsigned)(i * 2654435761u) ^ seed;
    u32x4 q;
    for (int k = 0; k < 4; k++) {
        s = s * 1664525u + 1013904223u;
        q[k] = s;
    }
    ((u32x4 *)qs)[i] = q;
    s = s * 1664525u + 1013904223u;
    wd[i] = __float2half(0.0015f + 0.001f * (float)(s >> 24) / 256.0f);
}

template <int NC, int NL>
static void run_engine(const Args &a, int G, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        attr = true;
        CK(hipFuncSetAttribute((const void *)k_engine<NC, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B));
    }
    hipLaunchKernelGGL((k_engine<NC, NL>), dim3(G), dim3((NC + NL) * 64), LDS_B, st, a);
}

int main(int argc, char **argv) {
    const char *chain = argc > 1 ? argv[1] : "ffn";
    const int nops = std::min(argc > 2 ? atoi(argv[2]) : 64, MAX_OPS);
    const int NC = argc > 3 ? atoi(argv[3]) : 7;
    const int NL = argc > 4 ? atoi(argv[4]) : 1;
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int G = pr.multiProcessorCount;
    // op shapes
    std::vector<Op> ops(nops);
    std::vector<std::pair<int, int>> shapes;  // (K, M) cycle
    if (!strcmp(chain, "wo")) shapes = {{4096, 4096}};
    else if (!strcmp(chain, "ffn")) shapes = {{4096, 11008}, {11008, 4096}};
    else shapes = {{4096, 4096}, {4096, 4096}, {4096, 11008}, {11008, 4096}};  // "layer": ~qkv-ish, wo, gate-ish, down
    const int GMAX = 11008;
    size_t blk = 0;
    const size_t WBLK = (size_t)1 << 27;  // 128 Mi blocks = 2 GiB of qs: every op reads fresh HBM
    for (int i = 0; i < nops; i++) {
        const auto sh = shapes[i % shapes.size()];
        ops[i].K = sh.first;
        ops[i].M = sh.second;
        const size_t n = (size_t)sh.first / 32 * sh.second;
        if (blk + n > WBLK) blk = 0;
        ops[i].qs_off = (long long)blk * 16;
        ops[i].d_off = (long long)blk;
        blk += n;
    }
    double total_bytes = 0;
    for (auto &o : ops) total_bytes += (double)o.K / 32 * o.M * 18;
    printf("device %s CUs %d | chain %s, %d ops, NC %d NL %d, %.1f MB of weights per run\n", pr.gcnArchName, G, chain, nops, NC, NL,
           total_bytes / 1e6);
    uint8_t *wqs;
    __half *wd;
    CK(hipMalloc(&wqs, WBLK * 16));
    CK(hipMalloc(&wd, WBLK * 2));
    k_fill<<<(unsigned)((WBLK + 255) / 256), 256>>>(wqs, wd, WBLK, 12345u);
    CK(hipDeviceSynchronize());
    Args a;
    memset(&a, 0, sizeof(a));
    a.wqs = wqs;
    a.wd = wd;
    Op *dops;
    CK(hipMalloc(&dops, nops * sizeof(Op)));
    CK(hipMemcpy(dops, ops.data(), nops * sizeof(Op), hipMemcpyHostToDevice));
    a.ops = dops;
    a.nops = nops;
    a.gmax = GMAX;
    CK(hipMalloc(&a.gran, 2 * GMAX * 8));
    CK(hipMemset(a.gran, 0, 2 * GMAX * 8));
    std::vector<float> x0(GMAX);
    for (int i = 0; i < GMAX; i++) x0[i] = sinf(0.37f * i) + 0.25f * cosf(0.011f * i);
    float *dx0;
    CK(hipMalloc(&dx0, GMAX * 4));
    CK(hipMemcpy(dx0, x0.data(), GMAX * 4, hipMemcpyHostToDevice));
    a.x0 = dx0;
    CK(hipMalloc(&a.vecs, (size_t)nops * GMAX * 4));
    CK(hipMalloc(&a.err, 4));
    CK(hipMemset(a.err, 0, 4));
    CK(hipMalloc(&a.ts, nops * 32));
    CK(hipMalloc(&a.lts, G * 32));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    // ---- reference chain: ordinary launches ----
    int8_t *rlo, *rhi;
    float *rd, *rvecs;
    int *rs;
    CK(hipMalloc(&rlo, NBP_MAX * 16));
    CK(hipMalloc(&rhi, NBP_MAX * 16));
    CK(hipMalloc(&rd, NBP_MAX * 4));
    CK(hipMalloc(&rs, NBP_MAX * 4));
    CK(hipMalloc(&rvecs, (size_t)nops * GMAX * 4));
    auto ref_chain = [&]() {
        for (int i = 0; i < nops; i++) {
            const float *src = i == 0 ? dx0 : rvecs + (size_t)(i - 1) * GMAX;
            const int K = ops[i].K, M = ops[i].M;
            hipLaunchKernelGGL(k_ref_quant, dim3((K / 4 + 1023) / 1024), dim3(1024), 0, st, src, K, rlo, rhi, rd, rs);
            hipLaunchKernelGGL(k_ref_mv, dim3((M + 3) / 4), dim3(256), 0, st, wqs + ops[i].qs_off, wd + ops[i].d_off, K, M,
                               (const i32x4 *)rlo, (const i32x4 *)rhi, rd, rs, rvecs + (size_t)i * GMAX);
        }
    };
    ref_chain();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    ref_chain();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ref_ms;
    CK(hipEventElapsedTime(&ref_ms, e0, e1));
    printf("reference (2 plain launches per op, eager): %.1f us per op\n", ref_ms * 1e3 / nops);

    auto launch = [&](int mode, unsigned epoch) {
        a.mode = mode;
        a.epoch0 = epoch;
        switch (NC * 10 + NL) {
            case 31: run_engine<3, 1>(a, G, st); break;
            case 51: run_engine<5, 1>(a, G, st); break;
            case 71: run_engine<7, 1>(a, G, st); break;
            default: printf("(NC, NL) must be one of (3,1) (5,1) (7,1)\n"); exit(1);
        }
        CK(hipGetLastError());
    };
    unsigned epoch = 1000;
    for (int mode : {1, 0}) {
        for (int rep = 0; rep < 3; rep++) {
            epoch += 4096;
            CK(hipMemsetAsync(a.err, 0, 4, st));
            CK(hipEventRecord(e0, st));
            launch(mode, epoch);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned err;
            CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
            std::vector<long long> lts(G * 4), ts(nops * 4);
            CK(hipMemcpy(lts.data(), a.lts, G * 32, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ts.data(), a.ts, nops * 32, hipMemcpyDeviceToHost));
            double lmax = 0, stalls = 0;
            for (int c = 0; c < G; c++) {
                lmax = std::max(lmax, (double)(lts[c * 4 + 1] - lts[c * 4]) / 100.0);
                stalls += (double)lts[c * 4 + 2];
            }
            printf("%s rep %d: %8.1f us total, %6.2f us per op, %7.1f GB/s | loader span max %.1f us, ring-full polls/CU %.0f | err %u\n",
                   mode == 1 ? "stream-only" : "full chain ", rep, ms * 1e3, ms * 1e3 / nops, total_bytes / 1e3 / (ms * 1e3), lmax,
                   stalls / G, err);
            if (mode == 0 && rep == 2) {
                double g = 0, d = 0, p = 0, per = 0;
                int n = 0;
                for (int i = 2; i < nops; i++) {
                    g += (double)(ts[i * 4 + 1] - ts[i * 4]) / 100.0;
                    d += (double)(ts[i * 4 + 2] - ts[i * 4 + 1]) / 100.0;
                    p += (double)(ts[i * 4 + 3] - ts[i * 4 + 2]) / 100.0;
                    per += (double)(ts[i * 4] - ts[(i - 1) * 4]) / 100.0;
                    n++;
                }
                printf("  workgroup 0, wave 0, per op: gather+quantize %.2f us, dots %.2f us, end barrier %.2f us, period %.2f us\n",
                       g / n, d / n, p / n, per / n);
                // per-shape split
                for (size_t s = 0; s < shapes.size(); s++) {
                    double gg = 0, dd = 0;
                    int nn = 0;
                    for (int i = 2 + (int)s; i < nops; i += (int)shapes.size())
                        if (i % (int)shapes.size() == (int)s) {
                            gg += (double)(ts[i * 4 + 1] - ts[i * 4]) / 100.0;
                            dd += (double)(ts[i * 4 + 2] - ts[i * 4 + 1]) / 100.0;
                            nn++;
                        }
                    if (nn) printf("    shape K=%d M=%d: gather %.2f us, dots %.2f us\n", shapes[s].first, shapes[s].second, gg / nn, dd / nn);
                }
                // bit-exact check of every intermediate vector
                std::vector<float> ev((size_t)nops * GMAX), rv((size_t)nops * GMAX);
                CK(hipMemcpy(ev.data(), a.vecs, ev.size() * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(rv.data(), rvecs, rv.size() * 4, hipMemcpyDeviceToHost));
                long long bad = 0;
                int first_bad_op = -1;
                for (int i = 0; i < nops; i++)
                    for (int m = 0; m < ops[i].M; m++)
                        if (memcmp(&ev[(size_t)i * GMAX + m], &rv[(size_t)i * GMAX + m], 4)) {
                            bad++;
                            if (first_bad_op < 0) first_bad_op = i;
                        }
                printf("  check vs plain launches: %lld mismatching elements (first bad op %d) -> %s\n", bad, first_bad_op,
                       bad == 0 && err == 0 ? "BIT-EXACT" : "MISMATCH");
            }
        }
    }
    return 0;
}
