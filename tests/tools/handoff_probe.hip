// Probe: what does an 8-byte {tag, value} hand-off between two workgroups of ONE launch cost when both sit on the same XCD
// (one L2) and when they do not, as a function of how the granule is stored and how it is polled?  ("the
// 1.7 us edge is a cross-XCD price, the intra-XCD one has never been measured here".)
//   stores: agent-scope atomic (what gran_store emits: sc1 write-through) | plain global_store | system scope (sc0 sc1)
//   polls : agent-scope atomic load (gran_load: sc1) | vector load sc0 | scalar load behind s_dcache_inv (K$ -> L2) | plain load nt
// Two workgroups play ping-pong over two granules (R rounds, every wait bounded); a round trip is timed with s_memrealtime
// (100 MHz) on the pinger; half of it is one hand-off.  The other 254 workgroups idle (quiet) or stream HBM (loaded).
// Placement is read from HW_REG_XCC_ID; the probe also reports whether blockIdx mod 8 names the XCD (it does for the dispatcher's
// round robin on an otherwise idle device: what warm_next relies on, kernels/decode_fused.h).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/handoff_probe tests/tools/handoff_probe.hip && /tmp/handoff_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { ST_AGENT = 0, ST_PLAIN = 1, ST_SYS = 2 };
enum { LD_AGENT = 0, LD_SC0 = 1, LD_SCALAR = 2, LD_NT = 3 };
static const char *ST_NAME[] = {"agent-scope atomic (sc1)", "plain global_store", "system scope (sc0 sc1)"};
static const char *LD_NAME[] = {"agent-scope atomic (sc1)", "vector load sc0", "s_dcache_inv + s_load", "vector load nt"};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
template <int ST>
__device__ __forceinline__ void put(u64 *p, u64 v) {
    if constexpr (ST == ST_AGENT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (ST == ST_PLAIN) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int LD>
__device__ __forceinline__ u64 get(const u64 *p) {
    u64 v;
    if constexpr (LD == LD_AGENT) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (LD == LD_SC0) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (LD == LD_NT) asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("s_dcache_inv\n\ts_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

__global__ void k_where(unsigned *xcc) {
    if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
}

struct PP {
    u64 *ping, *pong;     // one granule each, 256 bytes apart
    long long *rtt;       // [R] round trips in 10 ns ticks, -1 = gave up
    unsigned *xcc;        // placement of this launch
    const u32x4 *stream;  // loaded variant: what the bystanders read
    size_t stream_n;
    unsigned *sink;
    int a, b, R, tag0, load_iters;
};
constexpr int SPIN_MAX = 1 << 16;

template <int ST, int LD>
__global__ __launch_bounds__(1024) void k_pingpong(PP c) {
    const int bid = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) c.xcc[bid] = xcc_id();
    if (bid != c.a && bid != c.b) {  // bystanders
        if (c.load_iters > 0) {
            u32x4 acc = {0, 0, 0, 0};
            size_t i = ((size_t)bid * 1024 + tid) % c.stream_n;
            for (int it = 0; it < c.load_iters; it++) {
                const u32x4 v = __builtin_nontemporal_load(c.stream + i);
                acc ^= v;
                i += (size_t)gridDim.x * 1024;
                if (i >= c.stream_n) i -= c.stream_n;
            }
            if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) c.sink[0] = 1;
        }
        return;
    }
    if (tid >= 64) return;
    if (tid != 0) return;  // one lane plays
    const bool pinger = bid == c.a;
    for (int r = 0; r < c.R; r++) {
        const u64 tag = (u64)(unsigned)(c.tag0 + r) << 32;
        if (pinger) {
            __builtin_amdgcn_s_sleep(32);  // the ponger is polling by now
            const long long t0 = wall_clock64();
            put<ST>(c.ping, tag | 1u);
            int spin = 0;
            while ((get<LD>(c.pong) >> 32) != (tag >> 32) && ++spin < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
            const long long t1 = wall_clock64();
            c.rtt[r] = spin < SPIN_MAX ? t1 - t0 : -1;
        } else {
            int spin = 0;
            while ((get<LD>(c.ping) >> 32) != (tag >> 32) && ++spin < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
            put<ST>(c.pong, tag | 2u);
        }
    }
}

template <int ST, int LD>
static void run(const char *where, int a, int b, int load_iters, PP c, int &tag0) {
    c.a = a; c.b = b; c.load_iters = load_iters; c.tag0 = tag0;
    tag0 += c.R;
    hipLaunchKernelGGL((k_pingpong<ST, LD>), dim3(256), dim3(1024), 0, 0, c);
    CK(hipDeviceSynchronize());
    std::vector<long long> rtt(c.R);
    std::vector<unsigned> xcc(256);
    CK(hipMemcpy(rtt.data(), c.rtt, c.R * sizeof(long long), hipMemcpyDeviceToHost));
    CK(hipMemcpy(xcc.data(), c.xcc, 256 * sizeof(unsigned), hipMemcpyDeviceToHost));
    int bad = 0;
    std::vector<double> ok;
    for (int r = 4; r < c.R; r++) (rtt[r] < 0 ? (void)bad++ : ok.push_back((double)rtt[r] * 0.01));  // us
    std::sort(ok.begin(), ok.end());
    printf("%-9s %-6s  store %-26s poll %-26s XCDs %u/%u  ", where, load_iters ? "loaded" : "quiet", ST_NAME[ST], LD_NAME[LD], xcc[a], xcc[b]);
    if (ok.empty()) printf("every round gave up (%d)\n", bad);
    else printf("one way us: min %.2f  median %.2f  p90 %.2f  max %.2f   gave up %d of %d\n", ok[0] / 2, ok[ok.size() / 2] / 2, ok[ok.size() * 9 / 10] / 2,
                ok.back() / 2, bad, c.R - 4);
}

// ---- second part: the all-gather edge of k_qkv_attn_wo in isolation.  NP publisher workgroups (blocks 0 .. NP-1) publish PER granules
// each DELAY us after their entry; the other workgroups poll from their entry on (as the mat-vec workgroups do while the attention
// runs) and record when they hold all NP * PER granules.  Variants of the readers:
//   G_SWEEP : every thread polls its granule(s) with agent-scope loads until all carry the tag (what wo_tail does)
//   G_FLAGS : one wave polls NP "head done" flags (one 8-byte granule per publisher, stored behind its data), then all threads read the
//             data ONCE and check the tags (a stale tag falls back to the sweep): 256 B of poll traffic per round instead of 10 KB
//   G_ONEWAVE : wave 0 alone sweeps all granules (ngran / 64 per lane)
//   G_RELAY : per XCD one workgroup sweeps memory (as G_SWEEP) and re-publishes the granules with plain stores into a per-XCD copy;
//             the other workgroups of the XCD poll that copy with nt loads (served by their own L2)
enum { G_SWEEP = 0, G_FLAGS = 1, G_ONEWAVE = 2, G_RELAY = 3 };
static const char *G_NAME[] = {"sweep by all threads (agent-scope loads)", "flags first, then one read of the data", "one wave sweeps everything",
                               "one relay workgroup per XCD + local copies (nt loads)"};
struct GA {
    u64 *gran;        // [NP * PER]
    u64 *flags;       // [NP]
    u64 *local;       // [8][NP * PER]: per-XCD copies (G_RELAY)
    long long *t_pub; // [NP]: publish time (before the first store)
    long long *t_got; // [256]: all granules held
    unsigned *sink;
    const u32x4 *stream;
    size_t stream_n;
    int NP, PER, delay_ticks, tag, load_iters;
};
template <int GV>
__global__ __launch_bounds__(1024) void k_gather(GA c) {
    __shared__ u64 s_buf[2048];
    __shared__ int s_ok;
    const int bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ngran = c.NP * c.PER;
    const u64 tag = (u64)(unsigned)c.tag;
    const long long t_in = wall_clock64();
    if (bid < c.NP) {  // publisher: PER granules by the first PER lanes of wave 0, then its flag
        if (wave != 0) return;
        while (wall_clock64() - t_in < c.delay_ticks) __builtin_amdgcn_s_sleep(8);
        if (lane == 0) c.t_pub[bid] = wall_clock64();
        for (int i = lane; i < c.PER; i += 64) put<ST_AGENT>(c.gran + bid * c.PER + i, (tag << 32) | (unsigned)(bid * c.PER + i));
        __builtin_amdgcn_s_waitcnt(0);  // the data left this wave before the flag does
        if (lane == 0) put<ST_AGENT>(c.flags + bid, (tag << 32) | 1u);
        return;
    }
    if (c.load_iters > 0 && wave == 15) {  // background traffic from every reader, fire and forget like warm_next's LDS-DMA loads:
        __shared__ unsigned s_junk[64];    // load_iters x 64 lanes x one 128-byte line each (100 KB per workgroup at 13 iterations x 64... see main)
        typedef const __attribute__((address_space(1))) void *wg_ptr;
        typedef __attribute__((address_space(3))) void *wl_ptr;
        const char *base = (const char *)c.stream;
        size_t off = ((size_t)bid * 64 + lane) * 128;
        for (int it = 0; it < c.load_iters; it++) {
            __builtin_amdgcn_global_load_lds((wg_ptr)(base + off), (wl_ptr)s_junk, 4, 0, 0);
            off += (size_t)gridDim.x * 64 * 128;
        }
    }
    auto sweep_all = [&](const u64 *src, auto ld) {  // every thread its granules, until all carry the tag
        for (int k = 0; 1024 * k < ngran; k++) {
            const int idx = tid + 1024 * k;
            const u64 *gp = src + (idx < ngran ? idx : 0);
            u64 x;
            for (int spin = 0; spin < SPIN_MAX; spin++) {
                x = ld(gp);
                if (__builtin_amdgcn_ballot_w64((x >> 32) == tag) == ~0ull) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (idx < ngran) s_buf[idx] = x;
        }
    };
    auto ld_agent = [](const u64 *p) { return get<LD_AGENT>(p); };
    auto ld_nt = [](const u64 *p) { return get<LD_NT>(p); };
    if constexpr (GV == G_SWEEP) {
        sweep_all(c.gran, ld_agent);
    } else if constexpr (GV == G_FLAGS) {
        if (wave == 0) {
            const u64 *fp = c.flags + (lane < c.NP ? lane : 0);
            for (int spin = 0; spin < SPIN_MAX; spin++) {
                const u64 x = get<LD_AGENT>(fp);
                if (__builtin_amdgcn_ballot_w64((x >> 32) == tag) == ~0ull) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        sweep_all(c.gran, ld_agent);  // first pass normally succeeds
    } else if constexpr (GV == G_ONEWAVE) {
        if (wave == 0)
            for (int i = lane; i < ngran; i += 64) {
                u64 x;
                for (int spin = 0; spin < SPIN_MAX; spin++) {
                    x = get<LD_AGENT>(c.gran + i);
                    if (__builtin_amdgcn_ballot_w64((x >> 32) == tag) == ~0ull) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                s_buf[i] = x;
            }
    } else {
        const int x8 = bid & 7;
        u64 *loc = c.local + (size_t)x8 * ngran;
        if ((bid >> 3) == (c.NP >> 3)) {  // the first reader workgroup of this XCD relays
            sweep_all(c.gran, ld_agent);
            for (int idx = tid; idx < ngran; idx += 1024) put<ST_PLAIN>(loc + idx, s_buf[idx]);
        } else {
            sweep_all(loc, ld_nt);
        }
    }
    __syncthreads();
    if (tid == 0) {
        c.t_got[bid] = wall_clock64();
        if (s_buf[ngran - 1] == 1) c.sink[1] = 1;
    }
}
template <int GV>
static void run_gather(GA c, int load_iters, int &tag0) {
    c.tag = tag0++;
    c.load_iters = load_iters;
    hipLaunchKernelGGL((k_gather<GV>), dim3(256), dim3(1024), 0, 0, c);
    CK(hipDeviceSynchronize());
    std::vector<long long> tp(c.NP), tg(256);
    CK(hipMemcpy(tp.data(), c.t_pub, c.NP * sizeof(long long), hipMemcpyDeviceToHost));
    CK(hipMemcpy(tg.data(), c.t_got, 256 * sizeof(long long), hipMemcpyDeviceToHost));
    const long long last_pub = *std::max_element(tp.begin(), tp.end()), first_pub = *std::min_element(tp.begin(), tp.end());
    std::vector<double> d;
    for (int b = c.NP; b < 256; b++) d.push_back((double)(tg[b] - last_pub) * 0.01);
    std::sort(d.begin(), d.end());
    printf("gather %-6s %-58s publishers spread %.2f us; all granules held, us after the last publish: min %.2f  median %.2f  p90 %.2f  max %.2f\n",
           load_iters ? "loaded" : "quiet", G_NAME[GV], (double)(last_pub - first_pub) * 0.01, d[0], d[d.size() / 2], d[d.size() * 9 / 10], d.back());
}

int main() {
    unsigned *xcc_d;
    CK(hipMalloc(&xcc_d, 256 * sizeof(unsigned)));
    std::vector<unsigned> xcc(256);
    int consistent = 0;
    for (int rep = 0; rep < 4; rep++) {
        hipLaunchKernelGGL(k_where, dim3(256), dim3(1024), 0, 0, xcc_d);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(xcc.data(), xcc_d, 256 * sizeof(unsigned), hipMemcpyDeviceToHost));
        bool same = true;
        for (int b = 8; b < 256; b++) same = same && xcc[b] == xcc[b & 7];
        consistent += same;
        printf("launch %d: XCC_ID of blocks 0..15:", rep);
        for (int b = 0; b < 16; b++) printf(" %u", xcc[b]);
        printf("   blockIdx mod 8 names the XCD: %s\n", same ? "yes" : "NO");
    }
    int same_b = -1, far_b = -1;
    for (int b = 40; b < 256 && (same_b < 0 || far_b < 0); b++) {  // partners away from block 0's neighbours
        if (xcc[b] == xcc[0] && same_b < 0) same_b = b;
        if (xcc[b] != xcc[0] && far_b < 0) far_b = b;
    }
    printf("pinger = block 0 (XCD %u), same-XCD partner = block %d, other-XCD partner = block %d\n", xcc[0], same_b, far_b);
    PP c;
    u64 *gr;
    CK(hipMalloc(&gr, 4096));
    CK(hipMemset(gr, 0, 4096));
    c.ping = gr; c.pong = gr + 32;
    c.R = 68;
    CK(hipMalloc(&c.rtt, c.R * sizeof(long long)));
    c.xcc = xcc_d;
    c.stream_n = ((size_t)1 << 30) / 16;
    CK(hipMalloc((void **)&c.stream, c.stream_n * 16));
    CK(hipMemset((void *)c.stream, 1, c.stream_n * 16));
    CK(hipMalloc(&c.sink, 4));
    int tag0 = 1;
    for (int load = 0; load < 2; load++) {
        const int it = load ? 600 : 0;
        run<ST_AGENT, LD_AGENT>("same XCD", 0, same_b, it, c, tag0);
        run<ST_AGENT, LD_AGENT>("far XCD", 0, far_b, it, c, tag0);
        run<ST_PLAIN, LD_SCALAR>("same XCD", 0, same_b, it, c, tag0);
        run<ST_PLAIN, LD_SC0>("same XCD", 0, same_b, it, c, tag0);
        run<ST_PLAIN, LD_NT>("same XCD", 0, same_b, it, c, tag0);
        run<ST_AGENT, LD_SCALAR>("same XCD", 0, same_b, it, c, tag0);
        run<ST_AGENT, LD_SC0>("same XCD", 0, same_b, it, c, tag0);
        run<ST_SYS, LD_SCALAR>("same XCD", 0, same_b, it, c, tag0);
        run<ST_AGENT, LD_SCALAR>("far XCD", 0, far_b, it, c, tag0);
        run<ST_PLAIN, LD_SCALAR>("far XCD", 0, far_b, it, c, tag0);
        run<ST_SYS, LD_AGENT>("far XCD", 0, far_b, it, c, tag0);
    }
    GA g;
    g.NP = 32; g.PER = 40; g.delay_ticks = 400;  // 4 us of polling before anything is published
    CK(hipMalloc(&g.gran, 2048 * 8)); CK(hipMemset(g.gran, 0, 2048 * 8));
    CK(hipMalloc(&g.flags, 64 * 8)); CK(hipMemset(g.flags, 0, 64 * 8));
    CK(hipMalloc(&g.local, 8 * 2048 * 8)); CK(hipMemset(g.local, 0, 8 * 2048 * 8));
    CK(hipMalloc(&g.t_pub, 64 * 8)); CK(hipMalloc(&g.t_got, 256 * 8));
    g.sink = c.sink; g.stream = c.stream; g.stream_n = c.stream_n;
    int gt = 1000;
    for (int rep = 0; rep < 3; rep++)
        for (int load = 0; load < 2; load++) {
            const int it = load;  // loaded: 13 (26) iterations x 64 lines of 128 bytes per reader workgroup = 24 (48) MB requested at entry, as the warm-up does
            run_gather<G_SWEEP>(g, it ? 13 : 0, gt);
            run_gather<G_FLAGS>(g, it ? 13 : 0, gt);
            run_gather<G_SWEEP>(g, it ? 26 : 0, gt);
            run_gather<G_RELAY>(g, it ? 13 : 0, gt);
        }
    return 0;
}
