// handoff.h — what workgroups of ONE launch use to hand work to each other (cull_fused.hip, entity_cull.hip): tickets,
// write-through stores and sc1 loads, the drain, bounded polls.  The per-XCD L2s of gfx950 are not coherent with each
// other and a CU's L1 is never refreshed by another CU's stores (cdna_hip_programming.md §6 Guideline 16): everything
// one workgroup reads of another's is stored write-through (sc1), drained (s_waitcnt vmcnt(0)) and signalled by a store
// or an atomic that follows the drain, and loaded with sc1 loads behind the poll that saw the signal — no cache
// write-back or invalidate anywhere.  Work is handed out by tickets: whatever a wave waits for is held by a wave that is
// already running and never waits for a later ticket, so nothing depends on dispatch order, residency or placement.
#pragma once
#include "kernels.h"
#include "scan.h"

namespace orbit {

namespace {

constexpr uint32_t kHandoffBufFlags = 0x00020000u; // raw buffer, 32-bit data format
constexpr uint32_t kHandoffNoOffset = 0xFFFFFF00u; // an offset outside every resource: loads read zero, stores are dropped
#ifndef ORBIT_FUSED_SPIN_LOG2
#define ORBIT_FUSED_SPIN_LOG2 18
#endif
constexpr uint32_t kSpinLimit = 1u << ORBIT_FUSED_SPIN_LOG2; // polls before a wait gives up (a good fraction of a second; a hop takes microseconds)

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
constexpr int kSc1 = 16; // aux bit 4 of the buffer intrinsics on gfx950: sc1 — loads bypass the L1s, stores write through

// flag words (entity chunks and tiles): value << 2 | state
constexpr uint32_t kAggregate = 1u, kPrefix = 2u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, kHandoffBufFlags);
}
__device__ __forceinline__ uint32_t ld_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, kSc1);
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, uint32_t v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, r, byte_off, 0, kSc1);
}
// every store this wave has issued has been performed (write-through stores: has reached memory)
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// between two polls (the clobber also keeps the poll's load inside the loop)
__device__ __forceinline__ void poll_pause() { asm volatile("s_sleep 1" ::: "memory"); }
__device__ __forceinline__ uint32_t ticket(uint32_t *counter) {
    return __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a value every lane of the wave holds, as a scalar: loop conditions and branches on it are scalar branches (the
// compiler cannot see that a word read from LDS behind a barrier is uniform, and wraps what follows — barriers
// included — in divergent-loop bookkeeping)
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// the entity phase's arrival word: chunks that have their records in memory << 40 | records so far — one atomic per
// chunk, one 8-B poll tells a waiting workgroup both "all chunks are in" and the total.  A chunk publishes at most
// capacity + 1 records (entity_chunks), so 40 bits hold the sum of any number of chunks a launch can have: the record
// field never carries into the chunk count, whatever mesh_infos holds.
constexpr unsigned long long kArriveOne = 1ull << 40, kArriveRecords = kArriveOne - 1ull;
__device__ __forceinline__ unsigned long long arrive(uint32_t *word, uint32_t records) {
    return __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(word), kArriveOne | records, __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
}
// the arrival word as loaded with one 8-byte load: {chunks in, records so far (saturated to 32 bits)}
template <class V2>
__device__ __forceinline__ uint32_t arrived_chunks(const V2 &w) { return w[1] >> 8; }
template <class V2>
__device__ __forceinline__ uint32_t arrived_records(const V2 &w) { return (w[1] & 0xFFu) != 0u ? 0xFFFFFFFFu : w[0]; }
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

} // namespace

} // namespace orbit
