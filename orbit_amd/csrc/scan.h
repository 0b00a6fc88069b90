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

// Cross-lane steps through DPP (data-parallel primitives of the VALU: no LDS round trip, where a __shfl is a
// ds_bpermute with ~100 cycles of latency per step).  The six steps below leave in lane i the combination of lanes
// 0..i: row_shr 1/2/4/8 scan inside each row of 16 lanes, row_bcast15 adds row 0 to row 1 and row 2 to row 3,
// row_bcast31 adds rows 0-1 to rows 2-3.  Lanes without a source (and rows masked off) take `identity`.
// All 64 lanes must be active at the call (wave-uniform control flow).
#define ORBIT_DPP_STEP(OP, ctrl, row_mask) \
    v = OP(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, ctrl, row_mask, 0xF, false))
#define ORBIT_DPP_SCAN(OP)          \
    ORBIT_DPP_STEP(OP, 0x111, 0xF); \
    ORBIT_DPP_STEP(OP, 0x112, 0xF); \
    ORBIT_DPP_STEP(OP, 0x114, 0xF); \
    ORBIT_DPP_STEP(OP, 0x118, 0xF); \
    ORBIT_DPP_STEP(OP, 0x142, 0xA); \
    ORBIT_DPP_STEP(OP, 0x143, 0xC)

__device__ __forceinline__ uint32_t dpp_op_add(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t dpp_op_max(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t dpp_op_or(uint32_t a, uint32_t b) { return a | b; }

// Inclusive wave64 scan (sum).
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const uint32_t identity = 0u;
    ORBIT_DPP_SCAN(dpp_op_add);
    return v;
}

// Inclusive scans (sum) of the two 32-lane halves of the wave, side by side: the first five steps of the above.
__device__ __forceinline__ uint32_t half_inclusive_scan(uint32_t v) {
    const uint32_t identity = 0u;
    ORBIT_DPP_STEP(dpp_op_add, 0x111, 0xF);
    ORBIT_DPP_STEP(dpp_op_add, 0x112, 0xF);
    ORBIT_DPP_STEP(dpp_op_add, 0x114, 0xF);
    ORBIT_DPP_STEP(dpp_op_add, 0x118, 0xF);
    ORBIT_DPP_STEP(dpp_op_add, 0x142, 0xA);
    return v;
}

// Lane i <- lane i - 1 of the same row of 16 lanes (row_shr:1); the row's first lane gets `first`.
__device__ __forceinline__ uint32_t row_shr1(uint32_t v, uint32_t first) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x111, 0xF, 0xF, false);
}

// Wave-wide reductions: the scan's last lane holds the total.
__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v) {
    const uint32_t identity = 0u;
    ORBIT_DPP_SCAN(dpp_op_add);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t wave_reduce_max(uint32_t v) {
    const uint32_t identity = 0u;
    ORBIT_DPP_SCAN(dpp_op_max);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t wave_reduce_or(uint32_t v) {
    const uint32_t identity = 0u;
    ORBIT_DPP_SCAN(dpp_op_or);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
#undef ORBIT_DPP_SCAN
#undef ORBIT_DPP_STEP

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
