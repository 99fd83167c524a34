// Top-k prefilter of a logits row on the device: the k largest entries as (value, id) pairs, value
// descending, lower id first among equal values — what a stable descending sort of the row would put first — so that the
// host sampler chain (crates/llm-base/src/samplers.rs:289-306 sample_token -> top-k / top-p / temperature) touches k
// entries instead of n_vocab and only k pairs cross PCIe.
//
// One 1024-thread workgroup.  Keys are unique 64-bit integers: (order-preserving image of the f32) << 32 | ~id.  Eight
// radix passes of 8 bits find the k-th largest key exactly (256-bin LDS histogram of the keys that match the prefix so
// far, suffix sums by 256 lanes); the k keys >= it are gathered and sorted by a bitonic network in LDS.  The row
// (<= a few hundred KB) is re-read from L2 in every pass.  NaNs order by their bit pattern (positive NaN above +inf).
#pragma once
#include "common.h"

#define TOPK_MAX 1024

__device__ __forceinline__ unsigned long long topk_key(float v, int id) {
    uint32_t u = __builtin_bit_cast(uint32_t, v);
    if (u == 0x80000000u) u = 0;  // -0.0 compares equal to 0.0 in the host sort: same key, the id decides
    const uint32_t o = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
    return ((unsigned long long)o << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)id);
}

__global__ void __launch_bounds__(1024) k_topk(const float *__restrict__ x, int n, int k, const int *__restrict__ extra_ids,
                                               int n_extra, float *out_vals, int *out_ids) {
    __shared__ unsigned int s_hist[256];
    __shared__ unsigned long long s_prefix, s_keys[TOPK_MAX];
    __shared__ int s_remaining, s_cnt;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_prefix = 0;
        s_remaining = k;
        s_cnt = 0;
    }
    unsigned long long mask = 0;
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 56 - 8 * pass;
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        for (int i = tid; i < n; i += 1024) {
            const unsigned long long key = topk_key(x[i], i);
            if ((key & mask) == prefix) atomicAdd(&s_hist[(unsigned)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        unsigned int above = 0, mine = 0;
        if (tid < 256) {
            for (int d = tid + 1; d < 256; d++) above += s_hist[d];
            mine = s_hist[tid];
        }
        const unsigned int rem = (unsigned int)s_remaining;
        __syncthreads();  // everyone has read s_remaining and the histogram
        if (tid < 256 && above < rem && rem <= above + mine) {  // exactly one digit
            s_prefix = prefix | ((unsigned long long)tid << shift);
            s_remaining = (int)(rem - above);
        }
        mask |= 0xFFull << shift;
        __syncthreads();
    }
    const unsigned long long kth = s_prefix;  // the k-th largest key
    int P = 1;
    while (P < k) P <<= 1;
    for (int i = tid; i < P; i += 1024) s_keys[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const unsigned long long key = topk_key(x[i], i);
        if (key >= kth) s_keys[atomicAdd(&s_cnt, 1)] = key;
    }
    __syncthreads();
    // bitonic sort, descending, P <= 1024 elements (one compare-exchange per thread and step)
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int i = tid;
            if (i < P) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool desc = (i & size) == 0;
                    const unsigned long long a = s_keys[i], b = s_keys[j];
                    if (desc ? a < b : a > b) {
                        s_keys[i] = b;
                        s_keys[j] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += 1024) {
        const int id = (int)(0xFFFFFFFFu - (uint32_t)(s_keys[i] & 0xFFFFFFFFull));
        out_ids[i] = id;
        out_vals[i] = x[id];
    }
    for (int i = tid; i < n_extra; i += 1024) {
        const int id = extra_ids[i];
        out_ids[k + i] = id;
        out_vals[k + i] = (id >= 0 && id < n) ? x[id] : 0.0f;
    }
}
