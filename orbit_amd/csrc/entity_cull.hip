// entity_cull.hip — per-entity frustum / HiZ cull, LOD pick and deterministic
// emission of MeshletDispatch records for gfx950, as ONE launch.
//
// Restates shaders/entity_cull.comp:104-245 (reference repo) under the launch
// protocol of src/passes/draw_gen.rs:327-380.  Differences in *how*, not *what*:
//   - the record append is a prefix sum, so the record sequence is the canonical
//     ascending-gID order instead of atomics order: a workgroup evaluates a chunk of
//     256 entity-draws, publishes the chunk's record count and adds up the counts
//     of the chunks before it inside the launch (entity_common.h entity_chunks;
//     rounds 1-3 made two launches of it: eval, then an emit that summed the block
//     counts — 13 + 11 us at BASELINE config 5, each mostly its own ramp);
//   - visibility words are written from wave64 ballots split into two 32-bit
//     words (the reference's `gl_SubgroupSize == 32` addressing made explicit);
//   - the {0,1,1} header clear (draw_gen.rs:356-363) is fused into the launch.
#include "entity_common.h"

namespace orbit {

namespace {

__device__ __forceinline__ void entity_cull_body(const EntityCullParams &p) {
    __shared__ EntityChunksLds lds;
    EntityChunksArgs a;
    a.dispatch_buffer = p.dispatch_buffer;
    a.dispatch_capacity = p.dispatch_capacity;
    a.draw_first = p.draw_first, a.draw_limit = p.draw_limit;
    a.ne_chunks = p.ne_chunks;
    a.sync = p.sync, a.ent_flags = p.ent_flags, a.total = p.total, a.status = p.status;
    entity_chunks(p, a, lds, [](uint32_t) {});
    // the last workgroup out zeroes the flags: the next launch (or a graph replay) finds them as this one did
    __syncthreads();
    if (threadIdx.x == 0) lds.word = ticket(p.sync + kSyBlocksDone) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (uniform(lds.word) != 0u) { // every other workgroup has made its last access to them
        for (uint32_t i = threadIdx.x; i < p.ne_chunks; i += kEntityBlock) p.ent_flags[i] = 0u;
        if (threadIdx.x < kFusedSyncWords) p.sync[threadIdx.x] = 0u;
    }
}

// The parameter block is the kernel's own argument (one view) or entry blockIdx.y of an array of them passed by value
// (several views in one launch: orbit_cull_views).
__global__ __launch_bounds__(kEntityBlock) void entity_cull_kernel(const EntityCullParams p) { entity_cull_body(p); }
__global__ __launch_bounds__(kEntityBlock) void entity_cull_views_kernel(const EntityCullViews all) {
    const EntityCullParams &p = all.v[blockIdx.y];
    if (p.ne_chunks == 0u) { // a view without entity-draws: the pre-clear {0,1,1} is the whole effect
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            uint32_t *hdr = reinterpret_cast<uint32_t *>(p.dispatch_buffer);
            hdr[0] = 0u, hdr[1] = 1u, hdr[2] = 1u;
            *p.total = 0u;
        }
        return;
    }
    entity_cull_body(p);
}

__global__ void write_dispatch_header(uint32_t *hdr, uint32_t x) {
    hdr[0] = x;
    hdr[1] = 1u;
    hdr[2] = 1u;
}

// chunks are drawn by ticket: any grid is right; one workgroup per chunk up to four per CU keeps every chunk in flight
uint32_t entity_grid(uint32_t chunks, uint32_t num_cus) { return max(min(chunks, num_cus * 4u), 1u); }

} // namespace

hipError_t launch_entity_cull(const EntityCullParams &p, uint32_t entity_draw_count, uint32_t num_cus, hipStream_t s) {
    if (entity_draw_count == 0) {
        // nothing to test: the pre-clear {0,1,1} is the whole effect (draw_gen.rs:361-362)
        hipLaunchKernelGGL(write_dispatch_header, dim3(1), dim3(1), 0, s,
                           reinterpret_cast<uint32_t *>(p.dispatch_buffer), 0u);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(entity_cull_kernel, dim3(entity_grid(p.ne_chunks, num_cus)), dim3(kEntityBlock), 0, s, p);
    return hipGetLastError();
}

// Several views in one launch (orbit_cull_views): the grid covers the view with the most entity-draws.
hipError_t launch_entity_cull_views(const EntityCullViews &all, uint32_t n, uint32_t max_entity_draw_count,
                                    uint32_t num_cus, hipStream_t s) {
    const uint32_t chunks = (max_entity_draw_count + kEntityBlock - 1) / kEntityBlock;
    hipLaunchKernelGGL(entity_cull_views_kernel, dim3(entity_grid(chunks, max(num_cus / max(n, 1u), 1u)), n),
                       dim3(kEntityBlock), 0, s, all);
    return hipGetLastError();
}

} // namespace orbit
