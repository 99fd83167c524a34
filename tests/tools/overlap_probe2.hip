// Probe 2: a chain of decode-like links (each streams 192 KB of weights per CU, reads the previous link's 16 KB
// activation, writes its own) — kernel boundary (linear hipGraph, today's plan) against two alternating streams with
// a flag hand-off that lets link k+1 request its first weights while link k is still running.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/overlap_probe2 tests/tools/overlap_probe2.hip && /tmp/overlap_probe2
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int STEPS = 12, PFN = 3, XN = 4096;

struct Chain {
    const u32x4 *w;        // weights: NREG regions of W*STEPS*1024 u32x4
    size_t region;         // u32x4 per region
    int nreg;
    float *act;            // [L+1][XN] activations, link k reads act[k], writes act[k+1]
    unsigned *count;       // [L] arrivals (monotonic over runs)
    unsigned *ready;       // [L][W*16] per-consumer flag words, one 64 B line each, value = run sequence number
    unsigned *err;
    unsigned *sink;        // every thread's checksum (keeps the weight loads of all lanes alive)
    long long *ts;         // [L][4]: entry, wait passed, exit of WG 0; [3] unused
};

__device__ __forceinline__ f32x4 load_sc1(const f32x4 *p) {  // agent-coherent 128-bit load (bypasses stale L2 lines)
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store_sc1(float *p, float v) {  // write-through to the agent-coherent level
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

template <int MODE>  // 0: plain loads/stores, ordering from the kernel boundary.  1: flag hand-off
__global__ __launch_bounds__(1024) void k_link(Chain c, int k, unsigned seq, int L) {
    __shared__ float s_x[XN];
    __shared__ float s_red[16];
    __shared__ int s_ok;
    const int tid = threadIdx.x, W = gridDim.x;
    if (tid == 0 && blockIdx.x == 0) c.ts[k * 4] = wall_clock64();
    // 1. first weights: independent of the previous link
    const u32x4 *w = c.w + (size_t)(k % c.nreg) * c.region + (size_t)blockIdx.x * STEPS * 1024 + tid;
    u32x4 r[PFN];
#pragma unroll
    for (int i = 0; i < PFN; i++) r[i] = __builtin_nontemporal_load(w + i * 1024);
    // 2. wait for the previous link
    if (MODE == 1) {
        if (tid == 0) {
            int ok = 1;
            if (k > 0) {
                const unsigned *f = c.ready + ((size_t)(k - 1) * W + blockIdx.x) * 16;
                long long t0 = wall_clock64();
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) {
                    __builtin_amdgcn_s_sleep(8);
                    if (wall_clock64() - t0 > 200000) { ok = 0; break; }
                }
            }
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok && tid == 0) atomicExch(c.err, 1u);
    }
    if (tid == 0 && blockIdx.x == 0) c.ts[k * 4 + 1] = wall_clock64();
    // 3. the activation
    const f32x4 *xp = (const f32x4 *)(c.act + (size_t)k * XN) + tid;
    f32x4 xv = MODE == 1 ? load_sc1(xp) : *xp;
    ((f32x4 *)s_x)[tid] = xv;
    __syncthreads();
    if (k > 0 && s_x[tid] != (float)k) atomicExch(c.err + 1, (unsigned)k);
    // 4. stream the weights
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < STEPS; i++) {
        u32x4 q = r[i % PFN];
        if (i + PFN < STEPS) r[i % PFN] = __builtin_nontemporal_load(w + (i + PFN) * 1024);
        acc += (q[0] ^ q[1]) + (q[2] ^ q[3]);
    }
    c.sink[blockIdx.x * 1024 + tid] = acc;
    float v = s_x[(tid * 5) & (XN - 1)] + 1.0f + (acc == 0x12345u ? 1.0f : 0.0f);
    // 5. this link's activation: 16 values per workgroup (XN = 16 * 256)
    if (tid < XN / W) {
        float *o = c.act + (size_t)(k + 1) * XN + blockIdx.x * (XN / W) + tid;
        if (MODE == 1) store_sc1(o, v); else *o = v;
    }
    if (MODE == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < 64) {
            unsigned old = 0;
            if (tid == 0) old = __hip_atomic_fetch_add(c.count + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __shfl(old, 0);
            if ((old + 1) % (unsigned)W == 0) {  // last arrival: wake every consumer workgroup of the next link
                for (int i = tid; i < W; i += 64)
                    __hip_atomic_store(c.ready + ((size_t)k * W + i) * 16, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (tid == 0 && blockIdx.x == 0) c.ts[k * 4 + 2] = wall_clock64();
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int W = pr.multiProcessorCount, L = 160;
    const int NREG = getenv("NREG") ? atoi(getenv("NREG")) : 8;  // 8: 400 MB (partly Infinity-Cache resident); 64: 3.2 GB like a 7B model
    printf("device %s CUs %d; link = %d KB/CU weights (%.1f MB), %d links\n", pr.gcnArchName, W, STEPS * 16,
           (double)W * STEPS * 16384 / 1e6, L);
    Chain c; c.nreg = NREG; c.region = (size_t)W * STEPS * 1024;
    u32x4 *wbuf; CK(hipMalloc(&wbuf, c.region * NREG * 16)); CK(hipMemset(wbuf, 1, c.region * NREG * 16)); c.w = wbuf;
    CK(hipMalloc(&c.act, (size_t)(L + 1) * XN * 4)); CK(hipMalloc(&c.count, L * 4)); CK(hipMemset(c.count, 0, L * 4));
    CK(hipMalloc(&c.ready, (size_t)L * W * 64)); CK(hipMemset(c.ready, 0, (size_t)L * W * 64));
    CK(hipMalloc(&c.sink, (size_t)W * 1024 * 4));
    CK(hipMalloc(&c.err, 8)); CK(hipMalloc(&c.ts, L * 32));
    hipStream_t s[2]; CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    unsigned seq = 0;

    auto reset = [&](hipStream_t st) { CK(hipMemsetAsync(c.err, 0, 8, st)); CK(hipMemsetAsync(c.act, 0, (size_t)(L + 1) * XN * 4, st)); };
    auto report = [&](const char *name, double host_us, float dev_ms) {
        unsigned err[2]; std::vector<long long> ts(L * 4); std::vector<float> act(XN);
        CK(hipMemcpy(err, c.err, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ts.data(), c.ts, L * 32, hipMemcpyDeviceToHost));
        CK(hipMemcpy(act.data(), c.act + (size_t)L * XN, XN * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < XN; i++) bad += act[i] != (float)L;
        double waits = 0, overlap = 0, body = 0;
        for (int k = 1; k < L; k++) {
            waits += (ts[k * 4 + 1] - ts[k * 4]) / 100.0;
            body += (ts[k * 4 + 2] - ts[k * 4 + 1]) / 100.0;
            overlap += (ts[(k - 1) * 4 + 2] - ts[k * 4]) / 100.0;  // >0: WG0 of link k entered before WG0 of k-1 left
        }
        printf("%-30s host %7.1f us  device %8.1f us  per link %6.2f us  timeout %u stale %u wrong %d | WG0: wait %5.2f "
               "body %5.2f  entered %5.2f us before predecessor's exit\n", name, host_us, dev_ms * 1e3, dev_ms * 1e3 / L,
               err[0], err[1], bad, waits / (L - 1), body / (L - 1), overlap / (L - 1));
    };

    printf("weights cycled: %.1f MB\n", (double)c.region * NREG * 16 / 1e6);
    for (int round = 0; round < 2; round++) {
        // A. linear hipGraph, kernel boundaries (today)
        {
            hipGraph_t gr; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < L; k++) hipLaunchKernelGGL(k_link<0>, dim3(W), dim3(1024), 0, s[0], c, k, 0u, L);
            CK(hipStreamEndCapture(s[0], &gr));
            CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
            for (int rep = 0; rep < 3; rep++) {
                reset(s[0]); CK(hipEventRecord(e0, s[0]));
                double h0 = now_us(); CK(hipGraphLaunch(ge, s[0])); double h1 = now_us();
                CK(hipEventRecord(e1, s[0])); CK(hipStreamSynchronize(s[0]));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep == 2) report("A linear hipGraph (today)", h1 - h0, ms);
            }
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
        }
        // B. one stream, direct launches, kernel boundaries
        for (int rep = 0; rep < 2; rep++) {
            reset(s[0]); CK(hipEventRecord(e0, s[0]));
            double h0 = now_us();
            for (int k = 0; k < L; k++) hipLaunchKernelGGL(k_link<0>, dim3(W), dim3(1024), 0, s[0], c, k, 0u, L);
            double h1 = now_us();
            CK(hipEventRecord(e1, s[0])); CK(hipStreamSynchronize(s[0]));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report("B one stream, direct", h1 - h0, ms);
        }
        // C. two streams alternating, flag hand-off
        for (int rep = 0; rep < 3; rep++) {
            seq++;
            reset(s[0]); CK(hipEventRecord(e0, s[0])); CK(hipEventRecord(ef, s[0])); CK(hipStreamWaitEvent(s[1], ef, 0));
            double h0 = now_us();
            for (int k = 0; k < L; k++) hipLaunchKernelGGL(k_link<1>, dim3(W), dim3(1024), 0, s[k & 1], c, k, seq, L);
            double h1 = now_us();
            CK(hipEventRecord(ej, s[1])); CK(hipStreamWaitEvent(s[0], ej, 0));
            CK(hipEventRecord(e1, s[0])); CK(hipStreamSynchronize(s[0]));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) report("C two streams + flags", h1 - h0, ms);
        }
        // D. one stream, flag protocol (its cost without any overlap)
        for (int rep = 0; rep < 2; rep++) {
            seq++;
            reset(s[0]); CK(hipEventRecord(e0, s[0]));
            double h0 = now_us();
            for (int k = 0; k < L; k++) hipLaunchKernelGGL(k_link<1>, dim3(W), dim3(1024), 0, s[0], c, k, seq, L);
            double h1 = now_us();
            CK(hipEventRecord(e1, s[0])); CK(hipStreamSynchronize(s[0]));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report("D one stream + flags", h1 - h0, ms);
        }
    }
    return 0;
}
