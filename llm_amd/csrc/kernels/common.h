// common.h — shared device-side types and helpers for the gfx950 kernels of libggml_hip.
// Written for CDNA4 only: 64-lane wavefronts are hard-coded (no warpSize abstraction, no CUDA path).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WAVE 64

// Strided 4-D tensor view handed to the generic kernels (byte strides, ggml convention).
struct TView {
    char *p;
    int64_t ne[4];
    int64_t nb[4];
};

// kinds of quantized weight blocks (device-side ids, dense so they can be template ints)
enum { QT_Q4_0 = 0, QT_Q4_1 = 1, QT_Q5_0 = 2, QT_Q5_1 = 3, QT_Q8_0 = 4 };

// Device layout of a quantized 2-D weight after ggml_hip_transform_tensor: structure-of-arrays,
// same bytes per block as the GGML file (18/20/22/24/34), but every array 16-byte aligned so a lane
// fetches its 16 quant bytes with one global_load_dwordx4 and a wave reads 1 KiB contiguous.
//   qs : [M][nb][16]   low plane  (Q4/Q5: the block's 16 nibble bytes; Q8_0: elements 0..15)
//   qs2: [M][nb][16]   Q8_0 only: elements 16..31
//   qh : [M][nb] u32   Q5 only: the 32 fifth bits
//   d  : [M][nb] f16   scale;  m: [M][nb] f16 min (Q4_1/Q5_1)
struct QWeight {
    const uint8_t *qs;
    const uint8_t *qs2;
    const uint32_t *qh;
    const __half *d;
    const __half *m;
    int64_t M;   // rows (ne1)
    int64_t nb;  // blocks per row (ne0/32)
    int qt;
    const void *w16;  // optional resident f16 copy [M][nb * 32] for the prompt GEMM (kernels/mmq_w16.h); nullptr = none
};

// Activations re-quantized to the weight type's vec_dot_type (Q8_0 / Q8_1 semantics of ggml), in a
// planar layout private to this backend:
//   lo : [N][nb][16] int8  elements 0..15 of each block      hi: [N][nb][16] elements 16..31
//   d  : [N][nb] f32 (Q8_0 kind: the value after an f16 round trip, as ggml stores d as fp16)
//   sum: [N][nb] i32  sum of the 32 quants (Q8_1's s = sum*d; also folds the -8/-16 zero points)
struct QAct {
    const i32x4 *lo;
    const i32x4 *hi;
    const float *d;
    const int *sum;
};

// ---- cross-lane reductions on the VALU (DPP) instead of the LDS crossbar -----------------------------
// `__shfl_xor` lowers to ds_bpermute_b32: an LDS-pipe round trip (~100+ cycles) per step, and the steps of a
// reduction are dependent.  The short kernels of the decode chain are latency chains, so their reductions use
// DPP row operations (a few cycles each): quad_perm ×2, row_half_mirror, row_mirror reduce a 16-lane row with
// every lane holding the row's result; rows are combined with v_readlane (wave-uniform results) or one bpermute.
#define DPP_QUAD_XOR1 0xB1        /* quad_perm [1,0,3,2] */
#define DPP_QUAD_XOR2 0x4E        /* quad_perm [2,3,0,1] */
#define DPP_ROW_HALF_MIRROR 0x141 /* lane i <-> 7-i inside each 8 lanes */
#define DPP_ROW_MIRROR 0x140      /* lane i <-> 15-i inside each 16 lanes */

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// reductions over aligned groups of 8 / 16 lanes: every lane of the group ends with the group's result
__device__ __forceinline__ float g8_max_f32(float v) {
    v = fmaxf(v, dpp_f32<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_f32<DPP_QUAD_XOR2>(v));
    return fmaxf(v, dpp_f32<DPP_ROW_HALF_MIRROR>(v));
}
__device__ __forceinline__ float g8_sum_f32(float v) {
    v += dpp_f32<DPP_QUAD_XOR1>(v);
    v += dpp_f32<DPP_QUAD_XOR2>(v);
    return v + dpp_f32<DPP_ROW_HALF_MIRROR>(v);
}
__device__ __forceinline__ int g8_sum_i32(int v) {
    v += dpp_i32<DPP_QUAD_XOR1>(v);
    v += dpp_i32<DPP_QUAD_XOR2>(v);
    return v + dpp_i32<DPP_ROW_HALF_MIRROR>(v);
}
__device__ __forceinline__ float g16_sum_f32(float v) {
    v += dpp_f32<DPP_QUAD_XOR1>(v);
    v += dpp_f32<DPP_QUAD_XOR2>(v);
    v += dpp_f32<DPP_ROW_HALF_MIRROR>(v);
    return v + dpp_f32<DPP_ROW_MIRROR>(v);
}
__device__ __forceinline__ float g16_max_f32(float v) {
    v = fmaxf(v, dpp_f32<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_f32<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_f32<DPP_ROW_HALF_MIRROR>(v));
    return fmaxf(v, dpp_f32<DPP_ROW_MIRROR>(v));
}
__device__ __forceinline__ int g16_sum_i32(int v) {
    v += dpp_i32<DPP_QUAD_XOR1>(v);
    v += dpp_i32<DPP_QUAD_XOR2>(v);
    v += dpp_i32<DPP_ROW_HALF_MIRROR>(v);
    return v + dpp_i32<DPP_ROW_MIRROR>(v);
}
// groups of 32 lanes: row reduction + one exchange with the neighbouring row
__device__ __forceinline__ float g32_max_f32(float v) {
    v = g16_max_f32(v);
    return fmaxf(v, __shfl_xor(v, 16, 64));
}
__device__ __forceinline__ int g32_sum_i32(int v) {
    v = g16_sum_i32(v);
    return v + __shfl_xor(v, 16, 64);
}
// whole wave (64 lanes): row reduction + the four row results through v_readlane; result is wave-uniform
__device__ __forceinline__ float wave_sum_f32(float v) {
    v = g16_sum_f32(v);
    const int i = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max_f32(float v) {
    v = g16_max_f32(v);
    const int i = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_f64<DPP_QUAD_XOR1>(v);
    v += dpp_f64<DPP_QUAD_XOR2>(v);
    v += dpp_f64<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_f64<DPP_ROW_MIRROR>(v);
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = (int)(unsigned)u, hi = (int)(unsigned)(u >> 32);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned l = (unsigned)__builtin_amdgcn_readlane(lo, 16 * k), h = (unsigned)__builtin_amdgcn_readlane(hi, 16 * k);
        r[k] = __builtin_bit_cast(double, ((unsigned long long)h << 32) | l);
    }
    return (r[0] + r[1]) + (r[2] + r[3]);
}

// f32 -> f16 -> f32 round trip (RNE), the rounding ggml applies wherever it stores fp16.
__device__ __forceinline__ float round_f16(float x) { return __half2float(__float2half_rn(x)); }

// ---- 8-byte granules: a 32-bit value handed to other workgroups of the SAME launch, tagged with the launch's epoch ---------
// One naturally aligned 8-byte agent-scope store {tag, value} (global_store_dwordx2 sc1: write-through) on the producer, 8-byte
// agent-scope loads on the consumer until the tag matches: the data is the flag, no fence, no counter, nothing to drain
// (cdna_hip_programming.md Guideline 16, form R2).  Global address space on both sides, never flat, never a plain store.
typedef __attribute__((address_space(1))) unsigned long long gu64;
__device__ __forceinline__ void gran_store(unsigned long long *g_, unsigned epoch, unsigned value) {
    __hip_atomic_store((gu64 *)g_, ((unsigned long long)epoch << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long gran_load(const unsigned long long *g_) {
    return __hip_atomic_load((gu64 *)g_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The same granule between two workgroups that share an XCD: a plain store lands in that XCD's L2, a non-temporal load is served by it
// (a sc0 load would be answered by the reader's L1 for ever, tests/tools/handoff_probe.hip): 0.3 us per hand-off instead of 0.6 on an
// idle chip, and — what matters inside a launch whose other workgroups keep the HBM queues full — no trip through the memory side
// at all.  ONLY for a producer / consumer pair on one XCD: another XCD's L2 never sees the store before the kernel ends.
__device__ __forceinline__ void gran_store_l2(unsigned long long *g_, unsigned epoch, unsigned value) {
    const unsigned long long v = ((unsigned long long)epoch << 32) | value;
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(g_), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned long long gran_load_l2(const unsigned long long *g_) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(g_) : "memory");
    return v;
}
// Every in-launch wait is bounded, and bounded in POLLS, not in wall-clock time: a poll is an L2 round trip plus an s_sleep
// (>= ~0.7 us), so 2^18 of them are >= ~0.2 s of the wave actually running.  A wave that is context-switched out (another
// process's queue on the same GPU) does not poll, so time-slicing cannot trip the bound the way a wall-clock limit could; a peer
// that is never scheduled (the launch is not fully resident) does.  A wait that gives up raises the plan's error word, which the
// host reads back with every token's results (llama_plan.inc token_finish): the token is re-run on the kernels that do not wait
// inside a launch, or the process aborts with a message.
#define GRAN_SPIN_MAX (1 << 18)

// ---- which branch of ggml's activation quantizer (quantize_row_q8_0 / quantize_row_q8_1) the kernels restate ---------
// The reference builds ggml with -mavx2 -mfma -mf16c on every AVX2 host (crates/ggml/sys/build.rs:46-62), so what its CPU
// mul_mat runs is upstream's `#elif defined(__AVX2__)` branch:   d = amax / 127 ;  id = amax != 0 ? 127 / amax : 0 ;
// q = cvtps_epi32(round_ps(x * id, NEAREST))  (round half to EVEN = v_rndne_f32) ; Q8_1's s = d * (float)sum(q).
// The scalar branch (`*_reference`, non-SIMD hosts) multiplies by id = d != 0 ? 1 / d : 0 and rounds with roundf (half away
// from zero).  The two differ where x * id sits within an ulp of a rounding edge and on exact ties.  Default = the AVX2
// branch (0); ggml_hip_set_option("act_quant", 1) / GGML_HIP_ACT_QUANT=scalar selects the scalar one.  Read through the
// scalar cache once per kernel; a plan captured in a hipGraph follows the option without being re-captured.
__constant__ int c_act_quant_scalar = 0;
__device__ __forceinline__ bool aq_scalar() { return c_act_quant_scalar != 0; }
template <bool SC>
__device__ __forceinline__ float act_id(float amax, float d) {
    if constexpr (SC) return d != 0.0f ? 1.0f / d : 0.0f;
    return amax != 0.0f ? 127.0f / amax : 0.0f;
}
template <bool SC>
__device__ __forceinline__ int act_q(float x) {
    if constexpr (SC) return (int)roundf(x);
    return (int)__builtin_rintf(x);  // v_rndne_f32
}
// the same with the branch taken at run time (kernels where the quantizer is a few instructions of a long launch)
__device__ __forceinline__ float act_id(float amax, float d, bool sc) { return sc ? act_id<true>(amax, d) : act_id<false>(amax, d); }
__device__ __forceinline__ int act_q(float x, bool sc) { return sc ? act_q<true>(x) : act_q<false>(x); }
