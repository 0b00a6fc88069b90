// cull_fused.hip — one cull of a scene of the reference's own size as ONE launch.
//
// The reference records a cull as a clear and two dispatches (src/passes/draw_gen.rs:283-322: entity_cull.comp, then
// meshlet_cull.comp over the records it appended); this library's large-scene path makes five launches of it (entity
// eval -> emit, meshlet eval -> scan -> emit), each of which is, on a scene of a few thousand entities, little more than
// its own chain of dependent loads: 36-40 us per cull at BASELINE configs 2-4 (profiles/r03_final_configs.json).  Here
// the whole cull is one persistent launch whose workgroups hand work to each other through memory:
//
//   entity chunks   a workgroup draws a chunk of 256 entity-draws (ticket), evaluates them (entity_common.h), publishes
//                   the chunk's record count, adds up the counts of the chunks before it (one poll: a few hundred
//                   words at most), stores its MeshletDispatch records write-through and signals;
//   tiles           once every chunk has signalled, waves draw wave tiles of 16 records (ticket), load all eight rows
//                   of the tile at once (one memory round trip), evaluate them with the code of meshlet_common.h —
//                   the survivors' command words stay in LDS —, publish the tile's survivor count, find their base
//                   with a decoupled look-back over the tiles before (aggregate / inclusive-prefix flags, 64 tiles per
//                   step) and write the 28-B commands: the canonical (record, lane) order, like the launches'.
//
// Work is handed out by tickets, so whatever a wave waits for is held by a wave that is already running and never waits
// for a later ticket: no assumption about dispatch order, residency or placement (HIP promises none), any grid size
// works.  Everything one workgroup reads of another's (record counts, records, tile flags) is stored write-through
// (sc1), drained (s_waitcnt vmcnt(0)) and signalled by a store / atomic that follows the drain, and loaded with sc1
// loads behind the poll that saw the signal — no cache write-back or invalidate anywhere (the per-XCD L2s are not
// coherent; cdna_hip_programming.md §6 Guideline 16).  The last workgroup to leave zeroes the flags: a replayed graph
// finds them as the first launch did.  Every spin is bounded (ORBIT_E_TIMEOUT is latched, nothing hangs).
//
// The same per-tile look-back lost against eval -> scan -> emit on the 50 M-meshlet scene (0.66 vs 0.42 ms,
// profiles/r01_notes.md: with ~100 k tiles the early finishers wait and the retries queue behind the bulk loads);
// with a few hundred tiles, all in flight at once, the wait is one hop.  The library takes this path for culls of at
// most kFusedMaxEntityDraws entity-draws (abi.hip) and always evaluates the 32-B Meshlet buffer here: a scene this
// small is bound by latency, not by bytes, and the survivors' command words are then already in the row registers.
// Outputs are bit-identical to the launches' (tests/test_fused_gpu.py).
#include "entity_common.h"
#include "meshlet_common.h"

namespace orbit {

namespace {

constexpr uint32_t kFusedWaves = 4;                // = kEntityBlock / 64
constexpr uint32_t kFusedCap = kTileRecords * 32u; // survivors a tile can have: its payload slab holds them all
#ifndef ORBIT_FUSED_SPIN_LOG2
#define ORBIT_FUSED_SPIN_LOG2 18
#endif
constexpr uint32_t kSpinLimit = 1u << ORBIT_FUSED_SPIN_LOG2; // polls before a wait gives up (a good fraction of a second; a hop takes microseconds)
static_assert(kFusedWaves * 64 == kEntityBlock, "one block shape for both phases");

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
constexpr int kSc1 = 16; // aux bit 4 of the buffer intrinsics on gfx950: sc1 — loads bypass the L1s, stores write through

// flag words (entity chunks and tiles): value << 2 | state
constexpr uint32_t kAggregate = 1u, kPrefix = 2u;

// what entity_eval_one reads, over the fused parameter block (ci and entities are the meshlet side's)
struct FusedEntity {
    const OrbitGpuCullInfo &ci;
    const uint8_t *entity_draw_buffer;
    const OrbitMeshInfo *mesh_infos;
    const OrbitEntityData *entities;
    uint32_t *visibility;
    PyramidView pyr;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, kBufFlags);
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
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Σ of the counts published for entity chunks [0, eb): every thread polls its share (bounded) — the whole look-back is
// one step, because a cull has at most a few hundred chunks.
__device__ __forceinline__ uint32_t chunks_before(const FusedCullParams &p, __amdgpu_buffer_rsrc_t EF, uint32_t eb,
                                                  uint32_t *smem) {
    uint32_t part = 0;
    for (uint32_t j = threadIdx.x; j < eb; j += kEntityBlock) {
        uint32_t v = ld_sc1(EF, j * 4u);
        for (uint32_t spins = 0; v == 0u; spins++) {
            if (spins == kSpinLimit) {
                latch_status(p.m.status, ORBIT_E_TIMEOUT);
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
    for (uint32_t i = 0; i < kFusedWaves; i++) sum += smem[i];
    __syncthreads();
    return uniform(sum);
}

// Exclusive prefix of the survivor counts of the tiles before `t` (wave-uniform), by decoupled look-back: lane i reads
// the flag of tile pos - 1 - i; the nearest inclusive prefix ends the walk, aggregates in front of it add up.  Tiles
// below 0 read as "prefix 0".
__device__ __forceinline__ uint32_t tiles_before(const FusedCullParams &p, __amdgpu_buffer_rsrc_t TF, uint32_t t, int lane) {
    uint32_t base = 0, spins = 0;
    for (uint32_t pos = t;;) {
        const bool exists = (uint32_t)lane < pos;
        uint32_t v = kPrefix;
        if (exists) v = ld_sc1(TF, (pos - 1u - (uint32_t)lane) * 4u);
        const uint64_t prefix = __ballot((v & 3u) == kPrefix), ready = __ballot((v & 3u) != 0u);
        const uint32_t k = prefix != 0ull ? (uint32_t)__builtin_ctzll(prefix) : 64u; // nearest tile with a prefix
        const uint64_t need = k < 63u ? ((2ull << k) - 1ull) : ~0ull;                 // it and every tile behind it
        if ((ready & need) != need) {
            if (++spins == kSpinLimit) {
                if (lane == 0) latch_status(p.m.status, ORBIT_E_TIMEOUT);
                return base;
            }
            poll_pause();
            continue;
        }
        base += wave_reduce_add((uint32_t)lane <= k ? v >> 2 : 0u);
        if (k < 64u) return base;
        pos -= 64u; // 64 aggregates and no prefix among them: the next 64 (pos >= 64 here: tiles below 0 are prefixes)
    }
}

// -DORBIT_FUSED_DEBUG builds only (tools/fused_probe.py): progress marks, written through so that a copy engine can
// read them while the launch is still running
#ifdef ORBIT_FUSED_DEBUG
#define FUSED_MARK(stage, value)                                                                                        \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 256u)                                                                    \
    __hip_atomic_store(p.m.debug_cycles + 64 + (blockIdx.x * 4u + (threadIdx.x >> 6)) * 8u + (stage),                   \
                       (unsigned long long)(value) | 1ull << 63, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define FUSED_MARK(stage, value)
#endif

template <int PASS, int PROJ>
__device__ __forceinline__ void cull_fused_body(const FusedCullParams &p) {
    __shared__ PlaneLds planes;
    __shared__ AlphaLds alpha_tab;
    __shared__ uint32_t smem[kFusedWaves + 1];
    __shared__ uint32_t s_off[kEntityBlock];
    __shared__ OrbitMeshletDispatch s_proto[kEntityBlock];
    __shared__ WaveTileLds slab[kFusedWaves];
    __shared__ CandLds<(PASS == 2 ? kCandRing : 1u)> ring[kFusedWaves]; // pass 2 only
    __shared__ Payload s_pay[kFusedWaves][kFusedCap];
    __shared__ uint32_t s_cmd[kFusedWaves][64 * 7];
    __shared__ uint32_t s_word[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t max_tiles = (p.m.dispatch_capacity + kTileRecords - 1u) / kTileRecords;
    const __amdgpu_buffer_rsrc_t SY = rsrc_of(p.sync, kFusedSyncWords * 4u);
    const __amdgpu_buffer_rsrc_t EF = rsrc_of(p.ent_flags, p.ne_chunks * 4u);
    const __amdgpu_buffer_rsrc_t TF = rsrc_of(p.tile_flags, max_tiles * 4u);
    // the dispatch records as the entity phase stores them: a slot past the capacity falls outside and is dropped
    uint8_t *dispatch_buffer = const_cast<uint8_t *>(p.m.dispatch_buffer);
    const __amdgpu_buffer_rsrc_t REC_W = rsrc_of(dispatch_buffer + ORBIT_DISPATCH_HEADER, p.e_dispatch_capacity * 16u);
    const FusedEntity E{p.m.ci, p.entity_draw_buffer, p.mesh_infos, p.m.entities, p.visibility, p.e_pyr};

    planes_to_lds(p.m, planes);
    const AlphaLds *alpha = alpha_table_fill(p.m, alpha_tab) ? &alpha_tab : nullptr;
    const uint32_t draw_count = *reinterpret_cast<const uint32_t *>(p.entity_draw_buffer);
    const uint32_t draw_end = min(draw_count, p.draw_limit);

    // ---------------------------------------------------------------- entity chunks (entity_cull.comp:104-245)
    // ONE thread-0 region per iteration (signal the chunk just finished, draw the next ticket), its result broadcast
    // behind a barrier and made a scalar: with a thread-0 region at the end of the body and another at the head of the
    // next iteration the compiler threaded the two together and let the other lanes of wave 0 run ahead through the
    // loop head's barrier — wave 0 then passed that barrier twice per iteration and the workgroup fell out of step
    // (found on hardware with the progress marks below; look for s_barrier inside a depth-2 loop in the ISA).
    if (threadIdx.x == 0) s_word[0] = ticket(p.sync + kSyEntTicket);
    __syncthreads();
    uint32_t eb = uniform(s_word[0]);
    __syncthreads();
    while (eb < p.ne_chunks) {
        FUSED_MARK(0, eb);
        const uint32_t g = p.draw_first + eb * kEntityBlock + threadIdx.x;
        const bool active = g < draw_end; // :106
        bool visible;
        const OrbitMeshletDispatch pr = entity_eval_one(E, g, active, visible);
        entity_write_visibility(E, g, draw_end, active && visible);
        const uint32_t n = (pr.meshlet_count + ORBIT_MESHLET_DISPATCH_SIZE - 1u) / ORBIT_MESHLET_DISPATCH_SIZE; // :210
        uint32_t chunk_total;
        s_off[threadIdx.x] = block_exclusive_scan<kFusedWaves>(n, smem, &chunk_total);
        chunk_total = uniform(chunk_total);
        s_proto[threadIdx.x] = pr;
        // the chunk's count first (whoever waits for it must never wait for this chunk's own look-back) ...
        st_sc1(EF, threadIdx.x == 0 ? eb * 4u : kNoOffset, chunk_total << 2 | kAggregate);
        FUSED_MARK(1, chunk_total);
        // ... then the records of the chunks before it
        const uint32_t base = chunks_before(p, EF, eb, smem);
        FUSED_MARK(2, base);
        for (uint32_t r = threadIdx.x; r < chunk_total; r += kEntityBlock) {
            const uint32_t owner = entity_record_owner(s_off, r);
            const OrbitMeshletDispatch rec = entity_record(s_proto[owner], r - s_off[owner]);
            const v4u w = {rec.entity_index, rec.meshlet_offset, rec.meshlet_count, rec.visibility_offset};
            const uint64_t slot = (uint64_t)base + r; // (the sum can pass 2^32 / 16 only far beyond any capacity)
            __builtin_amdgcn_raw_buffer_store_b128(w, REC_W, slot < p.e_dispatch_capacity ? (uint32_t)slot * 16u : kNoOffset, 0, kSc1);
        }
        drain_stores();
        __syncthreads(); // every wave's records are in memory; s_off / s_proto may be rewritten
        FUSED_MARK(3, eb);
        if (threadIdx.x == 0) {
            if (eb == p.ne_chunks - 1u) { // the last chunk knows the total: the {x,1,1} header (draw_gen.rs:356-363)
                const uint32_t total = base + chunk_total;
                uint32_t *hdr = reinterpret_cast<uint32_t *>(dispatch_buffer);
                hdr[0] = min(total, p.e_dispatch_capacity);
                hdr[1] = 1u;
                hdr[2] = 1u;
                *p.e_total = total;
                if (total > p.e_dispatch_capacity) latch_status(p.m.status, ORBIT_E_CAPACITY);
                if (min(total, min(p.e_dispatch_capacity, p.m.dispatch_capacity)) == 0u) {
                    // no tile will run: the empty draw list is written here
                    *reinterpret_cast<uint32_t *>(p.m.draw_buffer) = 0u;
                    *p.m.total = 0u;
                }
                st_sc1(SY, kSyTotal * 4u, total + 1u);
                drain_stores();
            }
            (void)ticket(p.sync + kSyEntDone); // behind the drains above: "this chunk's records are in memory"
            s_word[0] = ticket(p.sync + kSyEntTicket);
        }
        __syncthreads();
        eb = uniform(s_word[0]);
        __syncthreads();
    }

    // ---------------------------------------------------------------- every record is in memory
    if (threadIdx.x == 0) {
        uint32_t done = ld_sc1(SY, kSyEntDone * 4u);
        for (uint32_t spins = 0; done < p.ne_chunks; spins++) {
            if (spins == kSpinLimit) {
                latch_status(p.m.status, ORBIT_E_TIMEOUT);
                break;
            }
            poll_pause();
            done = ld_sc1(SY, kSyEntDone * 4u);
        }
        // stored and drained before the last chunk signalled
        const uint32_t total1 = ld_sc1(SY, kSyTotal * 4u);
        s_word[1] = total1 != 0u ? total1 - 1u : 0u;
    }
    __syncthreads();
    const uint32_t nrec = min(uniform(s_word[1]), min(p.e_dispatch_capacity, p.m.dispatch_capacity));
    FUSED_MARK(4, nrec);
    const uint32_t ntiles = (nrec + kTileRecords - 1u) / kTileRecords;
    const __amdgpu_buffer_rsrc_t REC_R = rsrc_of(p.m.dispatch_buffer + ORBIT_DISPATCH_HEADER, nrec * 16u);

    // ---------------------------------------------------------------- tiles (meshlet_cull.comp:108-255)
    const bool occ2 = PASS == 2 && p.m.ci.meshlet_visibility_buffer != ORBIT_NONE;
    WaveTileLds &L = slab[wave];
    Payload *pay = s_pay[wave];
    uint32_t *cmd = s_cmd[wave];
    uint32_t *out = reinterpret_cast<uint32_t *>(p.m.draw_buffer + ORBIT_DRAW_HEADER);
    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = ticket(p.sync + kSyTileTicket);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        FUSED_MARK(5, t);
        if (t >= ntiles) break;
        // the tile's 16 records (lane = record * 4 + matrix column; past the last record: zeros), matrices, slab
        const uint32_t rec_index = t * kTileRecords + ((uint32_t)lane >> 2);
        const auto rw = __builtin_amdgcn_raw_buffer_load_b128(REC_R, rec_index < nrec ? rec_index * 16u : kNoOffset, 0, kSc1);
        const uint4 rec = make_uint4(rw[0], rw[1], rw[2], rw[3]);
        const float4 mat = setup_load_mat(p.m, rec, lane);
        setup_write(p.m, L, rec, mat, lane);
        // all eight rows at once: one memory round trip per tile
        RowRegs<(int)kTileRows> q;
        rows_load<PASS, 0, (int)kTileRows>(p.m, L, lane, q);
        CandState cs; // cs.total = the tile's survivor count
        if (occ2) {
            if constexpr (PASS == 2) {
#define ORBIT_FUSED_ROW(R)                                                                                             \
    {                                                                                                                  \
        RowRegs<1> q1;                                                                                                 \
        q1.a[0] = q.a[R], q1.b[0] = q.b[R], q1.prev[0] = q.prev[R];                                                    \
        row_eval_occ<PROJ, R, kCandRing, false, kFusedCap>(p.m, L, planes, alpha, ring[wave], lane, q1, cs, pay);      \
    }
                ORBIT_FUSED_ROW(0) ORBIT_FUSED_ROW(1) ORBIT_FUSED_ROW(2) ORBIT_FUSED_ROW(3)
                ORBIT_FUSED_ROW(4) ORBIT_FUSED_ROW(5) ORBIT_FUSED_ROW(6) ORBIT_FUSED_ROW(7)
#undef ORBIT_FUSED_ROW
                tile_finish_occ<kCandRing, kFusedCap>(p.m, L, ring[wave], lane, cs, pay);
            }
        } else {
            cs.total = rows_eval<PASS, PROJ, 0, (int)kTileRows, false, kFusedCap>(p.m, L, planes, alpha, lane, q, 0u, pay);
        }
        const uint32_t n = cs.total;
        FUSED_MARK(6, n);
        // count out, base in
        uint32_t base = 0;
        if (t == 0u) {
            if (lane == 0) st_sc1(TF, 0u, n << 2 | kPrefix);
        } else {
            if (lane == 0) st_sc1(TF, t * 4u, n << 2 | kAggregate);
            base = tiles_before(p, TF, t, lane);
            if (lane == 0) st_sc1(TF, t * 4u, (base + n) << 2 | kPrefix);
        }
        // the commands (:216-230), 64 at a time through LDS so that each of the 7 stores covers 256 contiguous bytes
        wave_lds_fence(); // the payload was written by other lanes
        for (uint32_t j0 = 0; j0 < n; j0 += 64u) {
            const uint32_t j = j0 + (uint32_t)lane;
            const Payload e = j < n ? pay[j] : Payload{0u, 0u, 0u};
            const uint32_t rid = e.w & 15u;
            const uint4 r = L.r[rid].rec;
            const uint32_t vertex_count = (e.w >> 16) & 0xFFu, triangle_count = e.w >> 24;
            uint32_t *c = cmd + lane * 7;
            c[0] = triangle_count * 3u;       // cmd_index_count
            c[1] = 1u;                        // cmd_instance_count
            c[2] = (e.z + vertex_count) * 4u; // cmd_first_index
            c[3] = e.z;                       // cmd_vertex_offset = int(data_offset)
            c[4] = r.x;                       // cmd_first_instance = entity_index
            c[5] = e.y;                       // meshlet_vertex_offset
            c[6] = r.y + ((e.w >> 4) & 31u);  // meshlet_index
            wave_lds_fence();
            const uint32_t first = base + j0; // slot of the trip's first command
            const uint32_t room = first < p.m.draw_capacity ? p.m.draw_capacity - first : 0u;
            const uint32_t bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(min(min(n - j0, 64u), room) * 28u));
            const __amdgpu_buffer_rsrc_t win = rsrc_of(out + (size_t)first * 7u, bytes);
#pragma unroll
            for (uint32_t k = 0; k < 7; k++) {
                const uint32_t w = k * 64u + (uint32_t)lane;
                __builtin_amdgcn_raw_buffer_store_b32(cmd[w], win, w * 4u, 0, 0);
            }
            __builtin_amdgcn_wave_barrier(); // cmd is rewritten by the next trip
        }
        if (t == ntiles - 1u && lane == 0) { // `count` header: subsumes the fill_buffer(count = 0) of draw_gen.rs:411-417
            const uint32_t total = base + n;
            *p.m.total = total;
            *reinterpret_cast<uint32_t *>(p.m.draw_buffer) = min(total, p.m.draw_capacity);
            if (total > p.m.draw_capacity) latch_status(p.m.status, ORBIT_E_CAPACITY);
        }
        __builtin_amdgcn_wave_barrier(); // slab and payload are rewritten by the next tile
    }

    // ---------------------------------------------------------------- the last workgroup out resets the flags
    FUSED_MARK(7, 1);
    __syncthreads();
    if (threadIdx.x == 0) s_word[0] = ticket(p.sync + kSyBlocksDone) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (uniform(s_word[0]) != 0u) { // every other workgroup has made its last access to them
        for (uint32_t i = threadIdx.x; i < p.ne_chunks; i += kEntityBlock) p.ent_flags[i] = 0u;
        for (uint32_t i = threadIdx.x; i < ntiles; i += kEntityBlock) p.tile_flags[i] = 0u;
        if (threadIdx.x < kFusedSyncWords / 32u) p.sync[threadIdx.x * 32u] = 0u;
    }
}

// blockIdx.y = view (orbit_cull_views: one or several), every view on flags of its own.  The parameter blocks travel by
// value in the kernel-argument segment and are read at a uniform offset (scalar loads, as needed): the single-view
// form with the whole block in scalar registers spilled 67-86 of them.
template <int PASS, int PROJ>
__global__ __launch_bounds__(kEntityBlock, 2) void cull_fused_views_kernel(const FusedCullViews all, const ViewGroup g) {
    cull_fused_body<PASS, PROJ>(all.v[g.idx[blockIdx.y]]);
}

// Workgroups of one view: a wave per expected tile (a record per entity-draw or so: a wave per 8 entity-draws is
// plenty), a workgroup per entity chunk at least, never more than two per CU: the tickets make any number right, this
// one makes the tiles one round.
uint32_t fused_grid(uint32_t entity_draw_count, uint32_t num_cus) {
    const uint32_t chunks = (entity_draw_count + kEntityBlock - 1u) / kEntityBlock;
    const uint32_t waves = (entity_draw_count + 7u) / 8u;
    return max(min(max(chunks, (waves + kFusedWaves - 1u) / kFusedWaves), num_cus * 2u), 1u);
}

} // namespace

// orbit_cull_views: the n views of `all`, one launch per (occlusion pass, projection) they use
hipError_t launch_cull_fused_views(const FusedCullViews &all, const uint32_t *entity_draw_counts, uint32_t n,
                                   uint32_t num_cus, hipStream_t s) {
    bool done[ORBIT_MAX_CULL_VIEWS] = {};
    for (uint32_t i = 0; i < n; i++) {
        if (done[i]) continue;
        const uint32_t pass = all.v[i].m.ci.occlusion_pass, proj = all.v[i].m.ci.projection_type;
        ViewGroup g{};
        uint32_t m = 0, most = 0;
        for (uint32_t j = i; j < n; j++)
            if (!done[j] && all.v[j].m.ci.occlusion_pass == pass && all.v[j].m.ci.projection_type == proj) {
                done[j] = true;
                g.idx[m++] = (uint8_t)j;
                most = max(most, entity_draw_counts[j]);
            }
        // the views share the device: each gets its part of the two-workgroups-per-CU budget
        const dim3 grid(fused_grid(most, max(num_cus / m, 1u)), m), block(kEntityBlock);
        const bool ortho = proj == 1;
#define ORBIT_FUSED(PASS)                                                                              \
    if (ortho) hipLaunchKernelGGL((cull_fused_views_kernel<PASS, 1>), grid, block, 0, s, all, g);      \
    else hipLaunchKernelGGL((cull_fused_views_kernel<PASS, 0>), grid, block, 0, s, all, g);
        switch (pass) {
        case 0: ORBIT_FUSED(0) break;
        case 1: ORBIT_FUSED(1) break;
        default: ORBIT_FUSED(2) break;
        }
#undef ORBIT_FUSED
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

} // namespace orbit
