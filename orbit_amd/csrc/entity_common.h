// entity_common.h — one entity-draw of entity_cull (shaders/entity_cull.comp:104-245 of the reference repo) and
// the expansion of its proto-record into MeshletDispatch records, shared by the launches of entity_cull.hip and the
// one-launch cull of cull_fused.hip.  Header-only.
#pragma once
#include "kernels.h"
#include "scan.h"

namespace orbit {

namespace {

constexpr int kEntityBlock = 256; // entity-draws per block of every entity launch (one thread each)

// The shader body up to the record emission for entity-draw `g` (:106-209): the proto-record (meshlet_count == 0 when
// nothing is drawn) and `visible` for the pass-2 bitset.
// P: EntityCullParams, or any block with its ci / entity_draw_buffer / mesh_infos / entities / visibility / pyr.
template <class P>
__device__ __forceinline__ OrbitMeshletDispatch entity_eval_one(const P &p, uint32_t g, bool active, bool &visible) {
    const OrbitGpuCullInfo &ci = p.ci;
    const bool meshlet_occ = ci.meshlet_visibility_buffer != ORBIT_NONE;
    visible = false;
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

    return pr;
}

// entity visibility bitset, pass 2 only (:226-243): word = g/32, bit = g%32, from the wave64 ballot's two halves.
template <class P>
__device__ __forceinline__ void entity_write_visibility(const P &p, uint32_t g, uint32_t end, bool active_and_visible) {
    if (p.ci.occlusion_pass == 2) {
        const uint64_t b = __ballot(active_and_visible);
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t wave_g = g - lane;
        if (lane == 0 && wave_g < end) p.visibility[wave_g / 32] = (uint32_t)b;
        if (lane == 32 && wave_g + 32 < end) p.visibility[wave_g / 32 + 1] = (uint32_t)(b >> 32);
    }
}

// Record j of an entity whose proto-record is `o` (:211-223): every record before j is full (count / S == 1, :222).
__device__ __forceinline__ OrbitMeshletDispatch entity_record(const OrbitMeshletDispatch &o, uint32_t j) {
    const uint32_t S = ORBIT_MESHLET_DISPATCH_SIZE;
    OrbitMeshletDispatch rec;
    rec.entity_index = o.entity_index;
    rec.meshlet_offset = o.meshlet_offset + S * j;
    rec.meshlet_count = min(o.meshlet_count - S * j, S);
    rec.visibility_offset = o.visibility_offset + j;
    return rec;
}

// owner of output record r of a block = last entity whose exclusive record offset is <= r (entities without records
// share the next one's offset): bisection over the block's offsets in LDS
__device__ __forceinline__ uint32_t entity_record_owner(const uint32_t *s_off, uint32_t r) {
    uint32_t lo = 0, hi = kEntityBlock;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_off[mid] <= r) lo = mid; else hi = mid;
    }
    return lo;
}

} // namespace

} // namespace orbit
