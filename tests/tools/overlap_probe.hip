// Probe: can consecutive decode kernels overlap on gfx950, and what does a flag hand-off cost next to a kernel boundary?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/overlap_probe tests/tools/overlap_probe.hip && /tmp/overlap_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Chain {
    unsigned long long *progress;  // finished workgroups, cumulative
    unsigned *err;                 // set when a wait timed out
    float *buf;                    // payload written by k, read by k+1 (visibility check)
    long long *ts;                 // [kernel][2] entry / pass-wait device clock of WG 0
};

// wait until *progress >= need; bounded (~2 ms) so a protocol bug cannot hang the box
__device__ __forceinline__ bool wait_progress(const unsigned long long *p, unsigned long long need) {
    long long t0 = wall_clock64();
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 200000) return false;
    }
    return true;
}

// work_iters: a little dependent ALU work per kernel so the chain resembles 5 us kernels when asked
__global__ __launch_bounds__(1024) void k_link(Chain c, unsigned long long need, int k, int use_flags, int work_iters) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0) c.ts[k * 2] = wall_clock64();
        ok = 1;
        if (use_flags && need) ok = wait_progress(c.progress, need) ? 1 : 0;
        if (blockIdx.x == 0) c.ts[k * 2 + 1] = wall_clock64();
    }
    __syncthreads();
    if (!ok) { if (threadIdx.x == 0) atomicExch(c.err, 1u); }
    // visibility check: every thread reads what the previous link wrote and writes k+1
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float v = c.buf[i];
    if (k > 0 && v != (float)k) atomicExch(c.err + 1, (unsigned)k);
    float acc = v;
    for (int it = 0; it < work_iters; it++) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);
    c.buf[i] = (float)(k + 1) + (acc > 1e30f ? 1.f : 0.f);
    __syncthreads();
    if (threadIdx.x == 0 && use_flags) {
        __threadfence();
        __hip_atomic_fetch_add(c.progress, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int W = pr.multiProcessorCount, L = 160;
    printf("device %s CUs %d\n", pr.gcnArchName, W);
    Chain c;
    CK(hipMalloc(&c.progress, 8)); CK(hipMalloc(&c.err, 8)); CK(hipMalloc(&c.buf, (size_t)W * 1024 * 4));
    CK(hipMalloc(&c.ts, L * 16));
    hipStream_t s[2]; CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));

    auto reset = [&](hipStream_t st) {
        CK(hipMemsetAsync(c.progress, 0, 8, st)); CK(hipMemsetAsync(c.err, 0, 8, st));
        CK(hipMemsetAsync(c.buf, 0, (size_t)W * 1024 * 4, st));
    };
    auto report = [&](const char *name, double host_us, float dev_ms) {
        unsigned err[2]; unsigned long long prog; std::vector<long long> ts(L * 2);
        CK(hipMemcpy(err, c.err, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&prog, c.progress, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ts.data(), c.ts, L * 16, hipMemcpyDeviceToHost));
        double waits = 0; int early = 0;
        for (int k = 1; k < L; k++) { waits += (ts[k * 2 + 1] - ts[k * 2]) / 100.0; if (ts[k * 2] < ts[(k - 1) * 2 + 1]) early++; }
        printf("%-34s host %8.1f us  device %8.1f us  per link %6.2f us  timeout %u stale %u progress %llu  "
               "avg wait %5.2f us  links entered before predecessor passed its wait: %d\n",
               name, host_us, dev_ms * 1e3, dev_ms * 1e3 / L, err[0], err[1], prog, waits / (L - 1), early);
    };

    for (int work : {0, 4000}) {
        printf("--- work_iters %d\n", work);
        // 1. plain in-order stream, no flags: the kernel boundary
        for (int rep = 0; rep < 2; rep++) {
            reset(s[0]); CK(hipEventRecord(e0, s[0]));
            double h0 = now_us();
            for (int k = 0; k < L; k++) hipLaunchKernelGGL(k_link, dim3(W), dim3(1024), 0, s[0], c, 0ull, k, 0, work);
            double h1 = now_us();
            CK(hipEventRecord(e1, s[0])); CK(hipStreamSynchronize(s[0]));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report("in-order stream (boundary)", h1 - h0, ms);
        }
        // 2. same stream, hipExtAnyOrderLaunch + flags
        for (int rep = 0; rep < 2; rep++) {
            reset(s[0]); CK(hipEventRecord(e0, s[0]));
            double h0 = now_us();
            for (int k = 0; k < L; k++)
                hipExtLaunchKernelGGL(k_link, dim3(W), dim3(1024), 0, s[0], nullptr, nullptr, k ? hipExtAnyOrderLaunch : 0, c,
                                      (unsigned long long)k * W, k, 1, work);
            double h1 = now_us();
            CK(hipEventRecord(e1, s[0])); CK(hipStreamSynchronize(s[0]));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report("any-order launch + flags", h1 - h0, ms);
        }
        // 3. two streams alternating + flags (direct launches)
        for (int rep = 0; rep < 2; rep++) {
            reset(s[0]); CK(hipEventRecord(e0, s[0])); CK(hipEventRecord(ef, s[0])); CK(hipStreamWaitEvent(s[1], ef, 0));
            double h0 = now_us();
            for (int k = 0; k < L; k++)
                hipLaunchKernelGGL(k_link, dim3(W), dim3(1024), 0, s[k & 1], c, (unsigned long long)k * W, k, 1, work);
            double h1 = now_us();
            CK(hipEventRecord(ej, s[1])); CK(hipStreamWaitEvent(s[0], ej, 0));
            CK(hipEventRecord(e1, s[0])); CK(hipStreamSynchronize(s[0]));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report("two streams + flags", h1 - h0, ms);
        }
        // 4. the same captured in a two-branch hipGraph
        {
            hipGraph_t gr; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
            CK(hipMemsetAsync(c.progress, 0, 8, s[0]));
            CK(hipEventRecord(ef, s[0])); CK(hipStreamWaitEvent(s[1], ef, 0));
            for (int k = 0; k < L; k++)
                hipLaunchKernelGGL(k_link, dim3(W), dim3(1024), 0, s[k & 1], c, (unsigned long long)k * W, k, 1, work);
            CK(hipEventRecord(ej, s[1])); CK(hipStreamWaitEvent(s[0], ej, 0));
            CK(hipStreamEndCapture(s[0], &gr));
            CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemsetAsync(c.err, 0, 8, s[0])); CK(hipMemsetAsync(c.buf, 0, (size_t)W * 1024 * 4, s[0]));
                CK(hipEventRecord(e0, s[0]));
                double h0 = now_us();
                CK(hipGraphLaunch(ge, s[0]));
                double h1 = now_us();
                CK(hipEventRecord(e1, s[0])); CK(hipStreamSynchronize(s[0]));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep == 2) report("two-branch hipGraph + flags", h1 - h0, ms);
            }
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
        }
        // 5. linear hipGraph, no flags (what the plan does today)
        {
            hipGraph_t gr; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < L; k++) hipLaunchKernelGGL(k_link, dim3(W), dim3(1024), 0, s[0], c, 0ull, k, 0, work);
            CK(hipStreamEndCapture(s[0], &gr));
            CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
            for (int rep = 0; rep < 3; rep++) {
                reset(s[0]); CK(hipEventRecord(e0, s[0]));
                double h0 = now_us();
                CK(hipGraphLaunch(ge, s[0]));
                double h1 = now_us();
                CK(hipEventRecord(e1, s[0])); CK(hipStreamSynchronize(s[0]));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep == 2) report("linear hipGraph (today)", h1 - h0, ms);
            }
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
        }
    }
    return 0;
}
