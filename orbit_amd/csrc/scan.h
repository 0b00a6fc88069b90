// scan.h — wave64 / block scan primitives used for deterministic compaction.
//
// Output order of every append on the path is the canonical sequential order
// (SURVEY.md §8a quirk 2), so appends are prefix sums, never atomics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbit {

// Number of set bits of `mask` below this lane (wave64).
__device__ __forceinline__ uint32_t lane_prefix(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Inclusive wave64 scan (sum) via cross-lane shuffles.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__device__ __forceinline__ uint32_t wave_reduce_max(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t t = __shfl_xor(v, d, 64);
        v = v > t ? v : t;
    }
    return v;
}

// Exclusive scan over a block of NW waves; returns the exclusive prefix of
// `v` and stores the block total in *total.  `smem` holds NW+1 words.
template <int NW>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *smem, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const uint32_t s = smem[w];
        if (w < wave) base += s;
        sum += s;
    }
    __syncthreads();
    *total = sum;
    return base + inc - v;
}

} // namespace orbit
