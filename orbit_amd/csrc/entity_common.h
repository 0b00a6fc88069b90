// entity_common.h — one entity-draw of entity_cull (shaders/entity_cull.comp:104-245 of the reference repo) and
// the expansion of its proto-record into MeshletDispatch records, shared by the launches of entity_cull.hip and the
// one-launch cull of cull_fused.hip.  Header-only.
#pragma once
#include "handoff.h"

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
        // the mesh's 32-B side entry where the caller keeps one (kernels.h MeshSide): the same bits as the MeshInfo's
        uint4 side_a = make_uint4(0u, 0u, 0u, 0u), side_b = make_uint4(0u, 0u, 0u, 0u);
        if (p.mesh_side.table != nullptr && ed.mesh_index < p.mesh_side.count) {
            const uint4 *se = reinterpret_cast<const uint4 *>(p.mesh_side.table + ed.mesh_index);
            side_a = se[0], side_b = se[1];
        }
        const bool side = side_b.w != 0u;
        float4 bs = make_float4(__uint_as_float(side_a.x), __uint_as_float(side_a.y), __uint_as_float(side_a.z), __uint_as_float(side_a.w));
        if (!side) bs = *reinterpret_cast<const float4 *>(mi->bounding_sphere);
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
        for (int r = 0; r < 4; r++) pv[r] = madd(m3[r], 1.0f, madd(m2[r], bs.z, madd(m1[r], bs.y, m0[r] * bs.x)));
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
            lod = min(lod, (side ? side_b.x : mi->lod_count) - 1u);
            lod = min(lod, (uint32_t)ORBIT_MAX_MESH_LODS - 1u);
            OrbitMeshLod ml = {side_b.y, side_b.z};
            if (!side || lod != 0u) ml = mi->mesh_lods[lod];
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
// S = MESHLET_DISPATCH_SIZE = 1 << rec_shift (OrbitCaps.dispatch_size: 32, or 64 / 128 in pass 0).
__device__ __forceinline__ OrbitMeshletDispatch entity_record(const OrbitMeshletDispatch &o, uint32_t j, uint32_t rec_shift) {
    const uint32_t S = 1u << rec_shift;
    OrbitMeshletDispatch rec;
    rec.entity_index = o.entity_index;
    rec.meshlet_offset = o.meshlet_offset + S * j;
    rec.meshlet_count = min(o.meshlet_count - S * j, S);
    rec.visibility_offset = o.visibility_offset + j;
    return rec;
}

// The block's `total` records, thread = output record, handed to store(r, record).  The owner of output record r of a
// block = the last entity whose exclusive record offset is <= r (entities without records share the next one's offset):
// bisection over the block's offsets in LDS.  FOUR records per thread and trip:
// a bisection is eight dependent LDS reads and a workgroup of this stage runs one wave per SIMD with little else
// resident, so nothing hides that latency but the thread's own other records (the four chains are independent; the loop
// with one record per trip was 2/3 of a chunk's time at 8 records per entity).  kEntityBlock = 2^8: the bisection's trip
// count is fixed, the compiler interleaves the chains.
template <class Store>
__device__ __forceinline__ void entity_expand_records(const uint32_t *s_off, const OrbitMeshletDispatch *s_proto, uint32_t total,
                                                      uint32_t rec_shift, Store store) {
    constexpr uint32_t U = 4;
    for (uint32_t r0 = threadIdx.x; r0 < total; r0 += U * kEntityBlock) {
        uint32_t lo[U], hi[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) lo[u] = 0u, hi[u] = kEntityBlock;
#pragma unroll
        for (int step = 0; step < 8; step++) {
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t mid = (lo[u] + hi[u]) >> 1;
                if (s_off[mid] <= r0 + u * kEntityBlock) lo[u] = mid; else hi[u] = mid;
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t r = r0 + u * kEntityBlock;
            if (r < total) store(r, entity_record(s_proto[lo[u]], r - s_off[lo[u]], rec_shift));
        }
    }
}
static_assert(kEntityBlock == 256, "entity_expand_records: eight bisection steps");

// what entity_eval_one reads, over a fused parameter block (FusedCullParams: ci and entities are the meshlet side's)
struct FusedEntity {
    const OrbitGpuCullInfo &ci;
    const uint8_t *entity_draw_buffer;
    const OrbitMeshInfo *mesh_infos;
    MeshSideView mesh_side;
    const OrbitEntityData *entities;
    uint32_t *visibility;
    PyramidView pyr;
};

// ---------------------------------------------------------------------------------------------------------------
// The whole entity stage as ONE launch (entity_cull.hip's single launch, and the first phase of cull_fused.hip).
// A workgroup draws a chunk of kEntityBlock entity-draws (ticket), evaluates them, publishes the chunk's record
// count, adds up the counts published for the chunks before it (one poll: a cull has a few hundred chunks at most),
// stores its MeshletDispatch records write-through at their canonical positions and adds itself to the arrival word
// (chunks in << 32 | records so far); the last chunk in writes the {x,1,1} header (draw_gen.rs:356-363).
struct EntityChunksArgs {
    uint8_t *dispatch_buffer;
    uint32_t dispatch_capacity;
    uint32_t draw_first, draw_limit; // as EntityCullParams
    uint32_t ne_chunks;              // ceil(entity_draw_count / kEntityBlock) >= 1
    uint32_t rec_shift;              // log2 of MESHLET_DISPATCH_SIZE (5; 6 / 7: OrbitCaps.dispatch_size 64 / 128)
    uint32_t *sync;                  // kFusedSyncWords counters (kernels.h), zero between launches
    uint32_t *ent_flags;             // per chunk: records << 2 | state
    uint32_t *total;
    int32_t *status;
};
struct EntityChunksLds {
    uint32_t smem[kEntityBlock / 64 + 1];
    uint32_t off[kEntityBlock];
    OrbitMeshletDispatch proto[kEntityBlock];
    uint32_t word;
};

// Σ of the counts published for entity chunks [0, eb): every thread polls its share (bounded) — the whole look-back is
// one step, because a cull has at most a few hundred chunks.
__device__ __forceinline__ uint32_t chunks_before(int32_t *status, __amdgpu_buffer_rsrc_t EF, uint32_t eb, uint32_t *smem) {
    uint32_t part = 0;
    for (uint32_t j = threadIdx.x; j < eb; j += kEntityBlock) {
        uint32_t v = ld_sc1(EF, j * 4u);
        for (uint32_t spins = 0; v == 0u; spins++) {
            if (spins == kSpinLimit) {
                latch_status(status, ORBIT_E_TIMEOUT);
                break;
            }
            poll_pause();
            v = ld_sc1(EF, j * 4u);
        }
        part += v >> 2;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t w = wave_reduce_add(part);
    __syncthreads(); // smem is the block scan's
    if (lane == 0) smem[wave] = w;
    __syncthreads();
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t i = 0; i < kEntityBlock / 64; i++) sum += smem[i];
    __syncthreads();
    return uniform(sum);
}



// E: what entity_eval_one reads (+ entity_draw_buffer for the count); on_total(total): thread 0 of the last chunk in.
template <class E, class OnTotal>
__device__ __forceinline__ void entity_chunks(const E &ent, const EntityChunksArgs &a, EntityChunksLds &S, OnTotal on_total) {
    const __amdgpu_buffer_rsrc_t EF = rsrc_of(a.ent_flags, a.ne_chunks * 4u);
    // the dispatch records as this phase stores them: a slot past the capacity falls outside and is dropped
    const __amdgpu_buffer_rsrc_t REC_W = rsrc_of(a.dispatch_buffer + ORBIT_DISPATCH_HEADER, a.dispatch_capacity * 16u);
    const uint32_t draw_count = *reinterpret_cast<const uint32_t *>(ent.entity_draw_buffer);
    const uint32_t draw_end = min(draw_count, a.draw_limit);
    // ONE thread-0 region per iteration (signal the chunk just finished, draw the next ticket), its result broadcast
    // behind a barrier and made a scalar: with a thread-0 region at the end of the body and another at the head of the
    // next iteration the compiler threaded the two together and let the other lanes of wave 0 run ahead through the
    // loop head's barrier — wave 0 then passed that barrier twice per iteration and the workgroup fell out of step
    // (found on hardware, tools/fused_probe.py; look for s_barrier inside a depth-2 loop in the ISA).
    if (threadIdx.x == 0) S.word = ticket(a.sync + kSyEntTicket);
    __syncthreads();
    uint32_t eb = uniform(S.word);
    __syncthreads();
    while (eb < a.ne_chunks) {
        const uint32_t g = a.draw_first + eb * kEntityBlock + threadIdx.x;
        const bool active = g < draw_end; // :106
        bool visible;
        const OrbitMeshletDispatch pr = entity_eval_one(ent, g, active, visible);
        entity_write_visibility(ent, g, draw_end, active && visible);
        const uint32_t n = (pr.meshlet_count + (1u << a.rec_shift) - 1u) >> a.rec_shift; // :210
        uint32_t chunk_total;
        S.off[threadIdx.x] = block_exclusive_scan<kEntityBlock / 64>(n, S.smem, &chunk_total);
        // What this chunk PUBLISHES is cut to capacity + 1: the flag holds the count << 2 and the arrival word adds
        // chunks << 32 | records, so a count of 2^30 (a corrupt mesh_infos meshlet_count) would wrap the one and carry
        // into the chunk counter of the other — the header would never be written, or tiles would start early.  Every
        // record past the capacity is dropped anyway, and capacity + 1 is what latches ORBIT_E_CAPACITY.
        chunk_total = min(uniform(chunk_total), a.dispatch_capacity + 1u);
        S.proto[threadIdx.x] = pr;
        // the chunk's count first (whoever waits for it must never wait for this chunk's own look-back) ...
        st_sc1(EF, threadIdx.x == 0 ? eb * 4u : kHandoffNoOffset, chunk_total << 2 | kAggregate);
        // ... then the records of the chunks before it
        const uint32_t base = chunks_before(a.status, EF, eb, S.smem);
        entity_expand_records(S.off, S.proto, chunk_total, a.rec_shift, [&](uint32_t r, const OrbitMeshletDispatch &rec) {
            const v4u w = {rec.entity_index, rec.meshlet_offset, rec.meshlet_count, rec.visibility_offset};
            const uint64_t slot = (uint64_t)base + r; // (the sum can pass 2^32 / 16 only far beyond any capacity)
            __builtin_amdgcn_raw_buffer_store_b128(w, REC_W, slot < a.dispatch_capacity ? (uint32_t)slot * 16u : kHandoffNoOffset, 0, kSc1);
        });
        drain_stores();
        __syncthreads(); // every wave's records are in memory; off / proto may be rewritten
        if (threadIdx.x == 0) {
            // behind the drain above: "this chunk's records are in memory"
            const unsigned long long before = arrive(a.sync + kSyEntDone, chunk_total);
            if ((uint32_t)(before >> 40) == a.ne_chunks - 1u) { // the last chunk in: the {x,1,1} header (draw_gen.rs:356-363)
                const unsigned long long total64 = (before & kArriveRecords) + chunk_total;
                const uint32_t total = total64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)total64;
                uint32_t *hdr = reinterpret_cast<uint32_t *>(a.dispatch_buffer);
                hdr[0] = min(total, a.dispatch_capacity);
                hdr[1] = 1u;
                hdr[2] = 1u;
                *a.total = total;
                if (total > a.dispatch_capacity) latch_status(a.status, ORBIT_E_CAPACITY);
                on_total(total);
            }
            S.word = ticket(a.sync + kSyEntTicket);
        }
        __syncthreads();
        eb = uniform(S.word);
        __syncthreads();
    }
}

} // namespace

} // namespace orbit
