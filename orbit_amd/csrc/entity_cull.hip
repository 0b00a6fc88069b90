// entity_cull.hip — per-entity frustum / HiZ cull, LOD pick and deterministic
// emission of MeshletDispatch records for gfx950: ONE launch up to 32 768 entity-draws, two above.
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
    a.rec_shift = p.rec_shift;
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

// ---------------------------------------------------------------------------------------------------------------
// The two-launch form, for culls of more than kEntityOneLaunchChunks chunks.  Every workgroup of the one-launch form
// makes three returning atomics on shared words (its ticket, its arrival, its leaving), and one word serves ~88 of
// them per microsecond on this chip: at BASELINE config 5 (763 chunks) the tickets alone queue for 9 us and the launch
// takes 44 us where eval + emit take 13 + 11 (profiles/r04_notes.md).  Below a hundred chunks that queue is a microsecond
// and one launch wins.
constexpr uint32_t kEntityOneLaunchChunks = 128;

// One thread per entity-draw: evaluates the shader body up to the record
// emission, stores a proto-record (meshlet_count == 0 when nothing is drawn).
// The parameter block is the kernel's own argument (one view) or entry blockIdx.y of an array of them passed by value
// (several views in one launch: orbit_cull_views).

__device__ __forceinline__ void entity_cull_eval_body(const EntityCullParams &p) {
    __shared__ uint32_t smem[kEntityBlock / 64 + 1];
    const uint32_t local = blockIdx.x * kEntityBlock + threadIdx.x;
    const uint32_t g = p.draw_first + local;
    const uint32_t count = *reinterpret_cast<const uint32_t *>(p.entity_draw_buffer);
    const uint32_t end = min(count, p.draw_limit);
    const bool active = g < end; // entity_cull.comp:106
    bool visible;
    const OrbitMeshletDispatch pr = entity_eval_one(p, g, active, visible);
    entity_write_visibility(p, g, end, active && visible);

    p.proto[local] = pr;
    const uint32_t n = (pr.meshlet_count + (1u << p.rec_shift) - 1u) >> p.rec_shift; // :210
    uint32_t total;
    (void)block_exclusive_scan<kEntityBlock / 64>(n, smem, &total);
    if (threadIdx.x == 0) p.block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kEntityBlock) void entity_cull_eval(const EntityCullParams p) { entity_cull_eval_body(p); }
__global__ __launch_bounds__(kEntityBlock) void entity_cull_eval_views(const EntityCullViews all) {
    entity_cull_eval_body(all.v[blockIdx.y]);
}

// Sum of v over the block (every thread gets it).  `smem` holds NW words.
template <int NW>
__device__ __forceinline__ uint32_t block_reduce_add(uint32_t v, uint32_t *smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t w = wave_reduce_add(v);
    if (lane == 0) smem[wave] = w;
    __syncthreads();
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) sum += smem[i];
    __syncthreads();
    return sum;
}

// Expands proto-records into ceil(m/S) MeshletDispatch records each, at the
// scanned offsets (:211-223).  The block's base is the sum of the preceding
// blocks' record counts, reduced here (a few KB from L2) instead of by a scan
// launch in between; block 0 reduces all of them for the {x,1,1} header.
// Thread = output record (owner found by bisection over the block's offsets in
// LDS): consecutive lanes store consecutive 16-B records, where a thread-per-
// entity loop stores 16 B every n x 16 B.
__device__ __forceinline__ void entity_cull_emit_body(const EntityCullParams &p) {
    __shared__ uint32_t smem[kEntityBlock / 64 + 1];
    __shared__ uint32_t s_off[kEntityBlock];
    __shared__ OrbitMeshletDispatch s_proto[kEntityBlock];
    const uint32_t local = blockIdx.x * kEntityBlock + threadIdx.x;
    const OrbitMeshletDispatch pr = p.proto[local];
    const uint32_t n = (pr.meshlet_count + (1u << p.rec_shift) - 1u) >> p.rec_shift;
    const uint32_t preceding = blockIdx.x == 0 ? gridDim.x : blockIdx.x; // block 0: grand total
    uint32_t part = 0;
    for (uint32_t i = threadIdx.x; i < preceding; i += kEntityBlock) part += p.block_sums[i];
    const uint32_t reduced = block_reduce_add<kEntityBlock / 64>(part, smem);
    const uint32_t block_base = blockIdx.x == 0 ? 0u : reduced;
    uint32_t block_total;
    s_off[threadIdx.x] = block_exclusive_scan<kEntityBlock / 64>(n, smem, &block_total);
    s_proto[threadIdx.x] = pr;
    __syncthreads();
    OrbitMeshletDispatch *out = reinterpret_cast<OrbitMeshletDispatch *>(p.dispatch_buffer + ORBIT_DISPATCH_HEADER);
    entity_expand_records(s_off, s_proto, block_total, p.rec_shift, [&](uint32_t r, const OrbitMeshletDispatch &rec) {
        if (block_base + r < p.dispatch_capacity) out[block_base + r] = rec;
    });
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t total = reduced;
        *p.total = total;
        uint32_t *hdr = reinterpret_cast<uint32_t *>(p.dispatch_buffer);
        hdr[0] = min(total, p.dispatch_capacity);
        hdr[1] = 1u;
        hdr[2] = 1u;
        if (total > p.dispatch_capacity) latch_status(p.status, ORBIT_E_CAPACITY);
    }
}

__global__ __launch_bounds__(kEntityBlock) void entity_cull_emit(const EntityCullParams p) { entity_cull_emit_body(p); }
__global__ __launch_bounds__(kEntityBlock) void entity_cull_emit_views(const EntityCullViews all) {
    entity_cull_emit_body(all.v[blockIdx.y]);
}

__global__ void write_dispatch_header(uint32_t *hdr, uint32_t x) {
    hdr[0] = x;
    hdr[1] = 1u;
    hdr[2] = 1u;
}

// OrbitCaps.dispatch_size 64 / 128: the caller's dispatch buffer holds S-sized records (what the reference's entity_cull
// writes with that spec constant); the meshlet stage's kernels are built on 32-meshlet records — one wave row holds two.
// Every S-record becomes S / 32 records of 32 in a buffer of the context (a record's tail parts may be empty: count 0), in
// order: the canonical (record, lane) command order is the ascending meshlet order per entity however the records are cut
// (tests/golden/spirv_cull_dispatch_sizes.npz: the reference's binaries produce the same command list at 32, 64 and 128).
__device__ __forceinline__ void split_records_body(const SplitRecordsParams &p) {
    const uint32_t nrec = min(*reinterpret_cast<const uint32_t *>(p.src), p.src_capacity);
    const uint32_t parts = 1u << (p.rec_shift - 5u);
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v == 0u) {
        uint32_t *hdr = reinterpret_cast<uint32_t *>(p.dst);
        hdr[0] = nrec * parts, hdr[1] = 1u, hdr[2] = 1u;
    }
    if (v >= nrec * parts) return;
    const uint4 rec = reinterpret_cast<const uint4 *>(p.src + ORBIT_DISPATCH_HEADER)[v >> (p.rec_shift - 5u)];
    const uint32_t first = (v & (parts - 1u)) * 32u;
    const uint32_t count = rec.z > first ? min(rec.z - first, 32u) : 0u;
    // (an empty part keeps the record's own offset: no lane of it is ever active, no address past the entity is formed)
    reinterpret_cast<uint4 *>(p.dst + ORBIT_DISPATCH_HEADER)[v] =
        make_uint4(rec.x, count ? rec.y + first : rec.y, count, rec.w + (v & (parts - 1u)));
}
__global__ __launch_bounds__(256) void split_records_kernel(const SplitRecordsParams p) { split_records_body(p); }
__global__ __launch_bounds__(256) void split_records_views_kernel(const SplitRecordsViews all) { split_records_body(all.v[blockIdx.y]); }

// chunks are drawn by ticket: any grid is right; one workgroup per chunk up to four per CU keeps every chunk in flight
uint32_t entity_grid(uint32_t chunks, uint32_t num_cus) { return max(min(chunks, num_cus * 4u), 1u); }

} // namespace

hipError_t launch_entity_cull(const EntityCullParams &p, uint32_t entity_draw_count, uint32_t num_cus, hipStream_t s) {
#if !ORBIT_CONTRACT
    if (p.arith != 0u) return launch_entity_cull_contracted(p, entity_draw_count, num_cus, s); // OrbitCaps.arith_profile
#endif
    if (entity_draw_count == 0) {
        // nothing to test: the pre-clear {0,1,1} is the whole effect (draw_gen.rs:361-362)
        hipLaunchKernelGGL(write_dispatch_header, dim3(1), dim3(1), 0, s,
                           reinterpret_cast<uint32_t *>(p.dispatch_buffer), 0u);
        return hipGetLastError();
    }
    if (p.ne_chunks <= kEntityOneLaunchChunks) {
        hipLaunchKernelGGL(entity_cull_kernel, dim3(entity_grid(p.ne_chunks, num_cus)), dim3(kEntityBlock), 0, s, p);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(entity_cull_eval, dim3(p.ne_chunks), dim3(kEntityBlock), 0, s, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(entity_cull_emit, dim3(p.ne_chunks), dim3(kEntityBlock), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_split_records(const SplitRecordsParams &p, hipStream_t s) {
    const uint64_t threads = ((uint64_t)p.src_capacity << (p.rec_shift - 5u)) + 1u;
    hipLaunchKernelGGL(split_records_kernel, dim3((uint32_t)((threads + 255u) / 256u)), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_split_records_views(const SplitRecordsViews &all, uint32_t n, hipStream_t s) {
    uint64_t most = 1;
    for (uint32_t i = 0; i < n; i++) most = max(most, ((uint64_t)all.v[i].src_capacity << (all.v[i].rec_shift - 5u)) + 1u);
    hipLaunchKernelGGL(split_records_views_kernel, dim3((uint32_t)((most + 255u) / 256u), n), dim3(256), 0, s, all);
    return hipGetLastError();
}

// Several views in one launch (orbit_cull_views): the grid covers the view with the most entity-draws.
hipError_t launch_entity_cull_views(const EntityCullViews &all, uint32_t n, uint32_t max_entity_draw_count,
                                    uint32_t num_cus, hipStream_t s) {
#if !ORBIT_CONTRACT
    if (all.v[0].arith != 0u) return launch_entity_cull_views_contracted(all, n, max_entity_draw_count, num_cus, s);
#endif
    const uint32_t chunks = (max_entity_draw_count + kEntityBlock - 1) / kEntityBlock;
    if (chunks > kEntityOneLaunchChunks) { // (surplus blocks of the smaller views find nothing active and write zero sums)
        hipLaunchKernelGGL(entity_cull_eval_views, dim3(chunks, n), dim3(kEntityBlock), 0, s, all);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(entity_cull_emit_views, dim3(chunks, n), dim3(kEntityBlock), 0, s, all);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(entity_cull_views_kernel, dim3(entity_grid(chunks, max(num_cus / max(n, 1u), 1u)), n),
                       dim3(kEntityBlock), 0, s, all);
    return hipGetLastError();
}

} // namespace orbit
