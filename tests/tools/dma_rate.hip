// dma_rate.hip — how fast can ONE (or two) waves per CU stream HBM into LDS by LDS-DMA, as a function of the number of
// scalar instructions spent per 1 KiB request?  No consumers, no flow control (the ring is simply overwritten).
// Also answers: does the instruction's immediate offset move the LDS destination too?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_rate tests/tools/dma_rate.hip && /tmp/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// variant 0: 16 requests per asm block: one M0 write + immediate offsets on 4 requests, base advanced by 4 KiB (if the
//            immediate moves the LDS address too this fills 16 consecutive KiB; else each 4 requests overwrite one KiB)
// variant 1: 16 requests per asm block, M0 advanced before every request (s_add_u32 m0 + s_nop), base by s_add/s_addc
// variant 2: like 1 plus 8 extra scalar instructions per request (what a C loop costs)
template <int VAR, int NW>
__global__ void __launch_bounds__(NW * 64) k_rate(const char *w, size_t per_cu, int iters, unsigned *out, int depth) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned v_lane16 = lane * 16u;
    const unsigned long long b0 = (unsigned long long)(uintptr_t)(w + (size_t)blockIdx.x * per_cu + (size_t)wave * (per_cu / NW));
    unsigned long long base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b0 >> 32)) << 32) |
                              (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b0);
    const unsigned ring = 49152u * wave;  // 48 KiB of LDS per loader wave
    unsigned slot = 0;
    for (int it = 0; it < iters; it++) {  // 16 KiB per iteration
        unsigned dst = __builtin_amdgcn_readfirstlane(ring + slot);
        if (VAR == 0) {
            asm volatile(
                "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %0, %1 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024 nt\n\t"
                "global_load_lds_dwordx4 %0, %1 offset:2048 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072 nt\n\t"
                :: "v"(v_lane16), "s"(base), "s"(dst) : "memory");
            asm volatile(
                "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %0, %1 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024 nt\n\t"
                "global_load_lds_dwordx4 %0, %1 offset:2048 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072 nt\n\t"
                :: "v"(v_lane16), "s"(base + 4096), "s"(dst + 4096) : "memory");
            asm volatile(
                "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %0, %1 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024 nt\n\t"
                "global_load_lds_dwordx4 %0, %1 offset:2048 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072 nt\n\t"
                :: "v"(v_lane16), "s"(base + 8192), "s"(dst + 8192) : "memory");
            asm volatile(
                "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %0, %1 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024 nt\n\t"
                "global_load_lds_dwordx4 %0, %1 offset:2048 nt\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072 nt\n\t"
                :: "v"(v_lane16), "s"(base + 12288), "s"(dst + 12288) : "memory");
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const unsigned long long b = base + (unsigned)i * 1024u;
                const unsigned d = dst + (unsigned)i * 1024u;
                if (VAR == 2) {
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt\n\t"
                                 "s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0"
                                 :: "v"(v_lane16), "s"(b), "s"(d) : "memory");
                } else {
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(v_lane16), "s"(b), "s"(d) : "memory");
                }
            }
        }
        base += 16384;
        slot = slot + 16384 >= 49152 ? 0 : slot + 16384;
        if (depth == 48) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (depth == 32) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(47)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (out && blockIdx.x == 0 && threadIdx.x < 64) {  // offset test: what is in the first 4 KiB of the ring?
        for (int i = 0; i < 4; i++) out[i * 64 + lane] = *(const unsigned *)(smem + i * 1024 + lane * 16);
    }
}

// What does s_getreg_b32 IB_STS show while LDS-DMA requests are outstanding?  out[2i] = raw IB_STS right after issuing
// n[i] requests, out[2i+1] = raw IB_STS after s_waitcnt vmcnt(0).
__global__ void __launch_bounds__(64) k_ibsts(const char *w, unsigned *out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned v_lane16 = (threadIdx.x & 63) * 16u;
    const unsigned long long b0 = (unsigned long long)(uintptr_t)w;
    const unsigned long long base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b0 >> 32)) << 32) |
                                    (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b0);
    const int counts[6] = {1, 4, 12, 20, 36, 52};
#pragma unroll
    for (int t = 0; t < 6; t++) {
        for (int i = 0; i < counts[t]; i++) {
            const unsigned long long b = base + ((unsigned long long)(t * 64 + i) << 20);  // 1 MiB apart: cold lines
            const unsigned d = (unsigned)(i & 31) * 1024u;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(v_lane16), "s"(b), "s"(d) : "memory");
        }
        unsigned r0, r1;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_IB_STS)" : "=s"(r0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_IB_STS)" : "=s"(r1));
        if (threadIdx.x == 0) { out[2 * t] = r0; out[2 * t + 1] = r1; }
    }
}

__global__ void k_fill(unsigned *p, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (unsigned)(i >> 8);  // dword i holds its KiB index (1 KiB = 256 dwords)
}

template <int VAR, int NW>
static void run(const char *w, size_t per_cu, int G, unsigned *out, int depth, const char *name) {
    static bool attr = false;
    if (!attr) { attr = true; CK(hipFuncSetAttribute((const void *)k_rate<VAR, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304)); }
    const int iters = (int)(per_cu / NW / 16384);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_rate<VAR, NW>), dim3(G), dim3(NW * 64), 98304, 0, w, per_cu, iters, out, depth);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    printf("%-44s %d loader wave(s), <=%d in flight each: %7.1f GB/s (%.1f GB/s per CU)\n", name, NW, depth, (double)per_cu * G / 1e6 / best,
           (double)per_cu / 1e6 / best);
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int G = pr.multiProcessorCount;
    const size_t per_cu = (size_t)6 << 20;  // 6 MiB per CU = 1.5 GiB total
    char *w; CK(hipMalloc(&w, per_cu * G));
    k_fill<<<(unsigned)(per_cu * G / 4 / 256), 256>>>((unsigned *)w, per_cu * G / 4);
    unsigned *out; CK(hipMalloc(&out, 1024)); CK(hipMemset(out, 0xff, 1024));
    CK(hipDeviceSynchronize());
    // offset semantics: one iteration of variant 0 on CU 0
    {
        CK(hipFuncSetAttribute((const void *)k_rate<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
        hipLaunchKernelGGL((k_rate<0, 1>), dim3(1), dim3(64), 98304, 0, w, per_cu, 1, out, 48);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> h(256); CK(hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost));
        printf("after 16 requests (4 per M0 write, immediates 0/1024/2048/3072): LDS KiB 0..3 hold source KiB %u %u %u %u  ->  %s\n",
               h[0], h[64], h[128], h[192], (h[0] == 0 && h[64] == 1 && h[128] == 2 && h[192] == 3) ? "the immediate offset moves the LDS address too"
                                                                                                  : "the immediate offset applies to the global address only");
    }
    {
        CK(hipFuncSetAttribute((const void *)k_ibsts, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        hipLaunchKernelGGL(k_ibsts, dim3(1), dim3(64), 65536, 0, w, out);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> h(12); CK(hipMemcpy(h.data(), out, 48, hipMemcpyDeviceToHost));
        const int counts[6] = {1, 4, 12, 20, 36, 52};
        for (int t = 0; t < 6; t++) {
            const unsigned r = h[2 * t], z = h[2 * t + 1];
            printf("IB_STS after issuing %2d requests: 0x%08x (bits 3:0 = %u, bits 23:22 = %u -> %u) | after vmcnt(0): 0x%08x\n", counts[t], r,
                   r & 15, (r >> 22) & 3, (r & 15) | (((r >> 22) & 3) << 4), z);
        }
    }
    for (int depth : {48}) {
        run<0, 1>(w, per_cu, G, nullptr, depth, "2.5 scalar instr per request (imm offsets)");
        run<1, 1>(w, per_cu, G, nullptr, depth, "~5 scalar instr per request");
        run<2, 1>(w, per_cu, G, nullptr, depth, "~13 scalar instr per request");
        run<0, 2>(w, per_cu, G, nullptr, depth, "2.5 scalar instr per request (imm offsets)");
        run<1, 2>(w, per_cu, G, nullptr, depth, "~5 scalar instr per request");
        run<2, 2>(w, per_cu, G, nullptr, depth, "~13 scalar instr per request");
    }
    run<1, 1>(w, per_cu, G, nullptr, 63, "~5 scalar instr per request");
    run<1, 2>(w, per_cu, G, nullptr, 63, "~5 scalar instr per request");
    return 0;
}
