// entity_cull.hip — per-entity frustum / HiZ cull, LOD pick and deterministic
// emission of MeshletDispatch records for gfx950.
//
// Restates shaders/entity_cull.comp:104-245 (reference repo) under the launch
// protocol of src/passes/draw_gen.rs:327-380.  Differences in *how*, not *what*:
//   - the record append is a prefix sum (eval -> emit, the emit launch sums the
//     preceding blocks' counts itself), so the record sequence is the canonical
//     ascending-gID order instead of atomics order;
//   - visibility words are written from wave64 ballots split into two 32-bit
//     words (the reference's `gl_SubgroupSize == 32` addressing made explicit);
//   - the {0,1,1} header clear (draw_gen.rs:356-363) is fused into the emit.
#include "kernels.h"
#include "scan.h"

namespace orbit {

namespace {

constexpr int kBlock = 256;

// One thread per entity-draw: evaluates the shader body up to the record
// emission, stores a proto-record (meshlet_count == 0 when nothing is drawn).
// The parameter block is the kernel's own argument (one view) or entry blockIdx.y of an array of them passed by value
// (several views in one launch: orbit_cull_views).

__device__ __forceinline__ void entity_cull_eval_body(const EntityCullParams &p) {
    __shared__ uint32_t smem[kBlock / 64 + 1];
    const OrbitGpuCullInfo &ci = p.ci;
    const uint32_t local = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t g = p.draw_first + local;
    const uint32_t count = *reinterpret_cast<const uint32_t *>(p.entity_draw_buffer);
    const uint32_t end = min(count, p.draw_limit);
    const bool active = g < end; // entity_cull.comp:106
    const bool meshlet_occ = ci.meshlet_visibility_buffer != ORBIT_NONE;

    bool visible = false;
    OrbitMeshletDispatch pr = {0u, 0u, 0u, 0u};
    if (active) {
        const OrbitEntityDraw *draws =
            reinterpret_cast<const OrbitEntityDraw *>(p.entity_draw_buffer + ORBIT_ENTITY_DRAW_HEADER);
        const OrbitEntityDraw ed = draws[g];
        const OrbitMeshInfo *mi = p.mesh_infos + ed.mesh_index;
        const float4 bs = *reinterpret_cast<const float4 *>(mi->bounding_sphere);
        const float4 *mm = reinterpret_cast<const float4 *>(p.entities[ed.entity_index].model_matrix);
        const float4 c0 = mm[0], c1 = mm[1], c2 = mm[2], c3 = mm[3];

        visible = true;
        bool visible_in_buffer = true;
        if (ci.occlusion_pass == 1 || ci.occlusion_pass == 2)
            visible_in_buffer = (p.visibility[g / 32] & (1u << (g % 32))) != 0; // :117-120
        if (ci.occlusion_pass == 1) visible = visible_in_buffer;                // :123

        // view * model, transform_sphere (:37-43,:131-134)
        float m0[4], m1[4], m2[4], m3[4];
        mat4_mul_col(ci.view_matrix, c0.x, c0.y, c0.z, c0.w, m0);
        mat4_mul_col(ci.view_matrix, c1.x, c1.y, c1.z, c1.w, m1);
        mat4_mul_col(ci.view_matrix, c2.x, c2.y, c2.z, c2.w, m2);
        mat4_mul_col(ci.view_matrix, c3.x, c3.y, c3.z, c3.w, m3);
        float pv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) pv[r] = ((m0[r] * bs.x + m1[r] * bs.y) + m2[r] * bs.z) + m3[r] * 1.0f;
        Sphere s;
        s.x = pv[0] / pv[3];
        s.y = pv[1] / pv[3];
        s.z = pv[2] / pv[3];
        const float dx = dot3(m0[0], m0[1], m0[2], m0[0], m0[1], m0[2]);
        const float dy = dot3(m1[0], m1[1], m1[2], m1[0], m1[1], m1[2]);
        const float dz = dot3(m2[0], m2[1], m2[2], m2[0], m2[1], m2[2]);
        const float scale = sqrtf(gmax(dx, gmax(dy, dz)));
        s.r = bs.w * scale;

        if (visible) visible = plane_test(ci, s);                                        // :137-144
        if (ci.occlusion_pass == 2 && visible) visible = occlusion_test(ci, s, bs.w, scale, p.pyr);  // :147-191

        bool should_draw = visible;
        if (ci.occlusion_pass == 2) should_draw = visible && (!visible_in_buffer || meshlet_occ); // :198-200

        if (should_draw) { // :203-209
            const float ex = ci.lod_target_pos_view_space[0] - s.x;
            const float ey = ci.lod_target_pos_view_space[1] - s.y;
            const float ez = ci.lod_target_pos_view_space[2] - s.z;
            const float lod_distance = sqrtf(dot3(ex, ey, ez, ex, ey, ez)) - s.r;
            const float lod_f = log2c(gmax(lod_distance, 0.0f) / ci.lod_base) / log2c(ci.lod_step);
            uint32_t lod = f2u_sat(gmax(lod_f + 1.0f, 0.0f));
            lod = min(max(lod, ci.min_mesh_lod), ci.max_mesh_lod);
            lod = min(lod, mi->lod_count - 1u);
            lod = min(lod, (uint32_t)ORBIT_MAX_MESH_LODS - 1u);
            const OrbitMeshLod ml = mi->mesh_lods[lod];
            pr.entity_index = ed.entity_index;
            pr.meshlet_offset = ml.meshlet_offset;
            pr.meshlet_count = ml.meshlet_count;
            pr.visibility_offset = ed.visibility_offset;
        }
    }

    // entity visibility bitset, pass 2 only (:226-243): word = g/32, bit = g%32.
    if (ci.occlusion_pass == 2) {
        const uint64_t b = __ballot(active && visible);
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t wave_g = g - lane;
        if (lane == 0 && wave_g < end) p.visibility[wave_g / 32] = (uint32_t)b;
        if (lane == 32 && wave_g + 32 < end) p.visibility[wave_g / 32 + 1] = (uint32_t)(b >> 32);
    }

    p.proto[local] = pr;
    const uint32_t n = (pr.meshlet_count + ORBIT_MESHLET_DISPATCH_SIZE - 1u) / ORBIT_MESHLET_DISPATCH_SIZE; // :210
    uint32_t total;
    (void)block_exclusive_scan<kBlock / 64>(n, smem, &total);
    if (threadIdx.x == 0) p.block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void entity_cull_eval(const EntityCullParams p) { entity_cull_eval_body(p); }
__global__ __launch_bounds__(kBlock) void entity_cull_eval_views(const EntityCullViews all) {
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
    __shared__ uint32_t smem[kBlock / 64 + 1];
    __shared__ uint32_t s_off[kBlock];
    __shared__ OrbitMeshletDispatch s_proto[kBlock];
    const uint32_t local = blockIdx.x * kBlock + threadIdx.x;
    const OrbitMeshletDispatch pr = p.proto[local];
    const uint32_t S = ORBIT_MESHLET_DISPATCH_SIZE;
    const uint32_t n = (pr.meshlet_count + S - 1u) / S;
    const uint32_t preceding = blockIdx.x == 0 ? gridDim.x : blockIdx.x; // block 0: grand total
    uint32_t part = 0;
    for (uint32_t i = threadIdx.x; i < preceding; i += kBlock) part += p.block_sums[i];
    const uint32_t reduced = block_reduce_add<kBlock / 64>(part, smem);
    const uint32_t block_base = blockIdx.x == 0 ? 0u : reduced;
    uint32_t block_total;
    s_off[threadIdx.x] = block_exclusive_scan<kBlock / 64>(n, smem, &block_total);
    s_proto[threadIdx.x] = pr;
    __syncthreads();
    OrbitMeshletDispatch *out = reinterpret_cast<OrbitMeshletDispatch *>(p.dispatch_buffer + ORBIT_DISPATCH_HEADER);
    for (uint32_t r = threadIdx.x; r < block_total; r += kBlock) {
        // owner = last entity whose offset is <= r (entities without records share the next one's offset)
        uint32_t lo = 0, hi = kBlock;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_off[mid] <= r) lo = mid; else hi = mid;
        }
        const OrbitMeshletDispatch o = s_proto[lo];
        const uint32_t j = r - s_off[lo];
        OrbitMeshletDispatch rec;
        rec.entity_index = o.entity_index;
        rec.meshlet_offset = o.meshlet_offset + S * j;
        rec.meshlet_count = min(o.meshlet_count - S * j, S);
        rec.visibility_offset = o.visibility_offset + j; // :222: every record before j is full (count / S == 1)
        if (block_base + r < p.dispatch_capacity) out[block_base + r] = rec;
    }
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

__global__ __launch_bounds__(kBlock) void entity_cull_emit(const EntityCullParams p) { entity_cull_emit_body(p); }
__global__ __launch_bounds__(kBlock) void entity_cull_emit_views(const EntityCullViews all) {
    entity_cull_emit_body(all.v[blockIdx.y]);
}

__global__ void write_dispatch_header(uint32_t *hdr, uint32_t x) {
    hdr[0] = x;
    hdr[1] = 1u;
    hdr[2] = 1u;
}

} // namespace

hipError_t launch_entity_cull(const EntityCullParams &p, uint32_t entity_draw_count, hipStream_t s) {
    const uint32_t blocks = (entity_draw_count + kBlock - 1) / kBlock;
    if (blocks == 0) {
        // nothing to test: the pre-clear {0,1,1} is the whole effect (draw_gen.rs:361-362)
        hipLaunchKernelGGL(write_dispatch_header, dim3(1), dim3(1), 0, s,
                           reinterpret_cast<uint32_t *>(p.dispatch_buffer), 0u);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(entity_cull_eval, dim3(blocks), dim3(kBlock), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(entity_cull_emit, dim3(blocks), dim3(kBlock), 0, s, p);
    return hipGetLastError();
}

// Several views in one launch pair (orbit_cull_views): the grid covers the view with the most entity-draws, the other
// views' surplus blocks find nothing active and write zero sums.
hipError_t launch_entity_cull_views(const EntityCullViews &all, uint32_t n, uint32_t max_entity_draw_count,
                                    hipStream_t s) {
    uint32_t blocks = (max_entity_draw_count + kBlock - 1) / kBlock;
    if (blocks == 0) blocks = 1; // emit block 0 still writes every view's {0,1,1} header
    hipLaunchKernelGGL(entity_cull_eval_views, dim3(blocks, n), dim3(kBlock), 0, s, all);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(entity_cull_emit_views, dim3(blocks, n), dim3(kBlock), 0, s, all);
    return hipGetLastError();
}

} // namespace orbit
