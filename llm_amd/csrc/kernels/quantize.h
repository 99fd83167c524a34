// ggml_quantize_q4_0 / q4_1 / q5_0 / q5_1 / q8_0 on the device: f32 rows -> raw GGML blocks, byte for
// byte what the host functions of the ABI produce (crates/ggml/src/lib.rs:419-483, called from
// crates/llm-base/src/quantize.rs:363-379; host restatement ggml_core.cpp quant_block, oracle quantize_row_q*).
// One lane per 32-weight block, the reference's scalar loop as written: the same f32 operations in the same order
// (the library is built with -ffp-contract=off and IEEE division), the same first-wins scan for the extreme value, the
// same float -> int8 truncations; f32 -> f16 by v_cvt_f16_f32 (round to nearest even, as ggml_fp32_to_fp16).
// The 16-bin histogram the quantizer reports is accumulated per workgroup in LDS and added to `hist` (u64 x 16).
#pragma once
#include "common.h"

__device__ __forceinline__ void q_put_f16(uint8_t *p, float f) {
    // the value is pinned in a register first: otherwise the compiler folds "x * c -> f16" into v_fma_mixlo_f16 x, c, +0,
    // and (-0) + (+0) = +0 loses the sign of d for an all-zero block (ggml stores 0x8000 there)
    asm volatile("" : "+v"(f));
    const __half h = __float2half_rn(f);
    const unsigned short u = __half_as_ushort(h);
    p[0] = (uint8_t)(u & 0xFF);
    p[1] = (uint8_t)(u >> 8);
}

// type: the ggml_type value (2, 3, 6, 7, 8).  x: nblocks * 32 floats; F16_SRC: the source is f16 (widened exactly)
template <bool F16_SRC>
__global__ void __launch_bounds__(256) k_quantize_blocks(const void *__restrict__ src, int type, int64_t nblocks, uint8_t *out,
                                                         unsigned long long *hist) {
    __shared__ unsigned int s_hist[16];
    if (threadIdx.x < 16) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblocks) {
        float x[32];
        if constexpr (F16_SRC) {
            const __half *p = (const __half *)src + b * 32;
#pragma unroll
            for (int j = 0; j < 32; j++) x[j] = __half2float(p[j]);
        } else {
            const f32x4 *p = (const f32x4 *)((const float *)src + b * 32);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const f32x4 v = p[j];
                x[4 * j] = v[0]; x[4 * j + 1] = v[1]; x[4 * j + 2] = v[2]; x[4 * j + 3] = v[3];
            }
        }
        float vmin = 3.402823466e+38f, vmax = -3.402823466e+38f, amax = 0.0f, amax_signed = 0.0f;
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const float v = x[j];
            if (v < vmin) vmin = v;
            if (v > vmax) vmax = v;
            if (amax < fabsf(v)) {
                amax = fabsf(v);
                amax_signed = v;
            }
        }
        const size_t bs = type == 2 ? 18 : type == 3 ? 20 : type == 6 ? 22 : type == 7 ? 24 : 34;
        uint8_t *o = out + (size_t)b * bs;
        if (type == 2) {  // Q4_0
            const float d = amax_signed / -8;
            const float id = d ? 1.0f / d : 0.0f;
            q_put_f16(o, d);
            for (int j = 0; j < 16; j++) {
                const int a = min(15, (int)(int8_t)(x[j] * id + 8.5f)), c = min(15, (int)(int8_t)(x[j + 16] * id + 8.5f));
                o[2 + j] = (uint8_t)(a | (c << 4));
                atomicAdd(&s_hist[a & 15], 1u);
                atomicAdd(&s_hist[c & 15], 1u);
            }
        } else if (type == 3) {  // Q4_1
            const float d = (vmax - vmin) / 15;
            const float id = d ? 1.0f / d : 0.0f;
            q_put_f16(o, d);
            q_put_f16(o + 2, vmin);
            for (int j = 0; j < 16; j++) {
                const int a = min(15, (int)(int8_t)((x[j] - vmin) * id + 0.5f)), c = min(15, (int)(int8_t)((x[j + 16] - vmin) * id + 0.5f));
                o[4 + j] = (uint8_t)(a | (c << 4));
                atomicAdd(&s_hist[a & 15], 1u);
                atomicAdd(&s_hist[c & 15], 1u);
            }
        } else if (type == 6 || type == 7) {  // Q5_0 / Q5_1
            const bool one = type == 7;
            const float d = one ? (vmax - vmin) / 31 : amax_signed / -16;
            const float id = d ? 1.0f / d : 0.0f;
            q_put_f16(o, d);
            if (one) q_put_f16(o + 2, vmin);
            uint8_t *qs = o + (one ? 8 : 6);
            uint32_t qh = 0;
            for (int j = 0; j < 16; j++) {
                uint8_t a, c;
                if (one) {
                    a = (uint8_t)((x[j] - vmin) * id + 0.5f);
                    c = (uint8_t)((x[j + 16] - vmin) * id + 0.5f);
                } else {
                    a = (uint8_t)min(31, (int)(int8_t)(x[j] * id + 16.5f));
                    c = (uint8_t)min(31, (int)(int8_t)(x[j + 16] * id + 16.5f));
                }
                qs[j] = (uint8_t)((a & 0x0F) | ((c & 0x0F) << 4));
                qh |= (uint32_t)((a & 0x10u) >> 4) << j;
                qh |= (uint32_t)((c & 0x10u) >> 4) << (j + 16);
            }
            uint8_t *ph = o + (one ? 4 : 2);
            ph[0] = (uint8_t)qh; ph[1] = (uint8_t)(qh >> 8); ph[2] = (uint8_t)(qh >> 16); ph[3] = (uint8_t)(qh >> 24);
            // upstream's histogram of the 5-bit types: j = 0, 2, .., 30 pairs qs[j/2] with the high bits at j and j + 16
            // (shift counts wrap as on x86), halved into 16 bins
            for (int j = 0; j < 32; j += 2) {
                const uint8_t vh0 = (uint8_t)(((qh & (1u << j)) >> j) << 4);
                const uint8_t vh1 = (uint8_t)((qh & (1u << ((j + 16) & 31))) >> ((j + 12) & 31));
                atomicAdd(&s_hist[(((qs[j / 2] & 0x0F) | vh0) / 2) & 15], 1u);
                atomicAdd(&s_hist[(((qs[j / 2] >> 4) | vh1) / 2) & 15], 1u);
            }
        } else {  // Q8_0
            const float d = fabsf(amax_signed) / 127;
            const float id = d ? 1.0f / d : 0.0f;
            q_put_f16(o, d);
            for (int j = 0; j < 32; j++) {
                const int q = (int)(int8_t)roundf(x[j] * id);
                o[2 + j] = (uint8_t)(int8_t)q;
                atomicAdd(&s_hist[(q / 16 + 8) & 15], 1u);
            }
        }
    }
    __syncthreads();
    if (hist && threadIdx.x < 16 && s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
}
