// Probe 3: does the L2 keep lines across a kernel boundary?  The chain of decode-like links of overlap_probe2 (192 KB of
// weights per CU and link, nothing cache-resident between repetitions), linear hipGraph.  Variant P: at its tail every
// workgroup requests the first PF ring steps of the NEXT link's weights for the same blockIdx (same XCD, hence the same
// L2) with ordinary loads, so that the next link's first requests could hit in L2 instead of paying the first-byte
// latency after the boundary.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/overlap_probe3 tests/tools/overlap_probe3.hip && NREG=64 /tmp/overlap_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int STEPS = 12, PFN = 3, XN = 4096;

struct Chain {
    const u32x4 *w;
    size_t region;
    int nreg;
    float *act;
    unsigned *sink;
    long long *ts;  // [L][4]: WG0 entry, first weights usable, exit
};

// PF: steps of the next link prefetched at the tail (0 = none).  NT_FIRST: the first PF steps of this link are read with
// ordinary loads (1) or non-temporal ones (0) — the prefetched lines must be found either way.  WHERE: 0 = prefetch after the last
// ring step was requested (under the tail of the stream), 1 = after the last dot (dead time before exit)
template <int PF, int WHERE>
__global__ __launch_bounds__(1024) void k_link(Chain c, int k) {
    __shared__ float s_x[XN];
    const int tid = threadIdx.x, W = gridDim.x;
    if (tid == 0 && blockIdx.x == 0) c.ts[k * 4] = wall_clock64();
    const u32x4 *w = c.w + (size_t)(k % c.nreg) * c.region + (size_t)blockIdx.x * STEPS * 1024 + tid;
    const u32x4 *wn = c.w + (size_t)((k + 1) % c.nreg) * c.region + (size_t)blockIdx.x * STEPS * 1024 + tid;
    u32x4 r[PFN];
#pragma unroll
    for (int i = 0; i < PFN; i++) r[i] = __builtin_nontemporal_load(w + i * 1024);
    // WHERE == 2: translation warm-up only — the first workgroup of every XCD touches one word per 2 MB of the NEXT link's
    // weights right after its own first requests (no data is prefetched; the loads hide under the stream)
    unsigned tlb = 0;
    if (WHERE == 2 && blockIdx.x < 8) {
        const size_t region_bytes = c.region * 16, page = (size_t)2 << 20;
        const char *nb = (const char *)(c.w + (size_t)((k + 1) % c.nreg) * c.region);
        for (size_t off = (size_t)tid * page; off < region_bytes; off += (size_t)1024 * page) tlb += *(const unsigned *)(nb + off);
    }
    const f32x4 *xp = (const f32x4 *)(c.act + (size_t)k * XN) + tid;
    f32x4 xv = *xp;
    ((f32x4 *)s_x)[tid] = xv;
    __syncthreads();
    unsigned acc = 0;
    u32x4 p[PF > 0 ? PF : 1];
#pragma unroll
    for (int i = 0; i < STEPS; i++) {
        u32x4 q = r[i % PFN];
        if (i == 0 && tid == 0 && blockIdx.x == 0) { asm volatile("" ::"v"(q[0])); c.ts[k * 4 + 1] = wall_clock64(); }
        if (i + PFN < STEPS) r[i % PFN] = __builtin_nontemporal_load(w + (i + PFN) * 1024);
        if (PF > 0 && WHERE == 0 && i + PFN == STEPS) {
#pragma unroll
            for (int j = 0; j < PF; j++) p[j] = wn[j * 1024];
        }
        acc += (q[0] ^ q[1]) + (q[2] ^ q[3]);
    }
    if (PF > 0 && WHERE == 1) {
#pragma unroll
        for (int j = 0; j < PF; j++) p[j] = wn[j * 1024];
    }
    if (PF > 0 && WHERE != 2) {
#pragma unroll
        for (int j = 0; j < PF; j++) acc += p[j][0] & 1u;
    }
    acc += tlb & 1u;
    c.sink[blockIdx.x * 1024 + tid] = acc;
    float v = s_x[(tid * 5) & (XN - 1)] + 1.0f + (acc == 0x12345u ? 1.0f : 0.0f);
    if (tid < XN / W) c.act[(size_t)(k + 1) * XN + blockIdx.x * (XN / W) + tid] = v;
    if (tid == 0 && blockIdx.x == 0) c.ts[k * 4 + 2] = wall_clock64();
}

template <int PF, int WHERE>
static void run(const char *name, Chain c, int W, int L, hipStream_t s) {
    hipGraph_t gr; hipGraphExec_t ge; hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < L; k++) hipLaunchKernelGGL((k_link<PF, WHERE>), dim3(W), dim3(1024), 0, s, c, k);
    CK(hipStreamEndCapture(s, &gr));
    CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    float best = 1e9f; double first = 0, body = 0;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipMemsetAsync(c.act, 0, (size_t)(L + 1) * XN * 4, s));
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<long long> ts(L * 4); std::vector<float> act(XN);
    CK(hipMemcpy(ts.data(), c.ts, L * 32, hipMemcpyDeviceToHost));
    CK(hipMemcpy(act.data(), c.act + (size_t)L * XN, XN * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < XN; i++) bad += act[i] != (float)L;
    for (int k = 1; k < L; k++) { first += (ts[k * 4 + 1] - ts[k * 4]) / 100.0; body += (ts[k * 4 + 2] - ts[k * 4]) / 100.0; }
    printf("%-44s per link %6.2f us | WG0: entry -> first weights %5.2f us, entry -> exit %5.2f us | wrong %d\n", name,
           best * 1e3 / L, first / (L - 1), body / (L - 1), bad);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int W = pr.multiProcessorCount, L = 160;
    const int NREG = getenv("NREG") ? atoi(getenv("NREG")) : 64;
    Chain c; c.nreg = NREG; c.region = (size_t)W * STEPS * 1024;
    u32x4 *wbuf; CK(hipMalloc(&wbuf, c.region * NREG * 16)); CK(hipMemset(wbuf, 1, c.region * NREG * 16)); c.w = wbuf;
    CK(hipMalloc(&c.act, (size_t)(L + 1) * XN * 4)); CK(hipMalloc(&c.sink, (size_t)W * 1024 * 4)); CK(hipMalloc(&c.ts, L * 32));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    printf("device %s CUs %d; link = %.1f MB, %d links, weights cycled %.1f MB\n", pr.gcnArchName, W,
           (double)W * STEPS * 16384 / 1e6, L, (double)c.region * NREG * 16 / 1e6);
    for (int round = 0; round < 2; round++) {
        run<0, 0>("A no prefetch", c, W, L, s);
        run<1, 0>("P1 next link's step 0, under the tail", c, W, L, s);
        run<3, 0>("P3 next link's steps 0-2, under the tail", c, W, L, s);
        run<3, 1>("P3 next link's steps 0-2, after the last dot", c, W, L, s);
        run<6, 0>("P6 next link's steps 0-5, under the tail", c, W, L, s);
        run<1, 2>("T  translations of the next link only", c, W, L, s);
    }
    return 0;
}
