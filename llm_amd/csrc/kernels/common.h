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

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// f32 -> f16 -> f32 round trip (RNE), the rounding ggml applies wherever it stores fp16.
__device__ __forceinline__ float round_f16(float x) { return __half2float(__float2half_rn(x)); }
