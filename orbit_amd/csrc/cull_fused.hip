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
#include "handoff.h"
#include "meshlet_common.h"

namespace orbit {

namespace {

constexpr uint32_t kFusedWaves = 4;                // = kEntityBlock / 64
// A wave's unit of meshlet work here is the first ROWS rows of a wave tile (2 ROWS records): 8 rows (the launches' tile)
// for larger scenes, 2 rows for the small ones — a lone wave evaluates a row in ~0.75 us (nothing else on its SIMD hides
// its LDS reads and dependent arithmetic: 6.1 us for 8 rows, measured with the stamps below), so a scene of a few
// hundred records is better cut into four times as many tiles on four times as many waves.
constexpr uint32_t fused_tile_records(int rows) { return 2u * (uint32_t)rows; }
constexpr uint32_t kFusedSmallEntityDraws = 4096; // views of up to this many entity-draws take the 2-row tiles
static_assert(kFusedWaves * 64 == kEntityBlock, "one block shape for both phases");

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
// (-DORBIT_FUSED_DEBUG=2: the time of the mark, 100 MHz ticks, instead of its value)
#define FUSED_MARK(stage, value)                                                                                        \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 256u)                                                                    \
    __hip_atomic_store(p.m.debug_cycles + 64 + (blockIdx.x * 4u + (threadIdx.x >> 6)) * 8u + (stage),                   \
                       (ORBIT_FUSED_DEBUG >= 2 ? (unsigned long long)wall_clock64() : (unsigned long long)(value)) |     \
                           1ull << 63,                                                                                   \
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define FUSED_MARK(stage, value)
#endif

// The workgroup's LDS: one set for the kernel, whichever pass its view runs (the three passes are three instantiations of
// the body inside ONE kernel — a view's pass is uniform per workgroup — so that a frame's views of different passes and
// projections are one launch).
// Scenes of up to kFusedLocalChunks chunks (768 entity-draws: BASELINE config 2 has 400) in passes 0 and 1: EVERY workgroup
// evaluates every entity-draw itself and keeps the proto-records and their record offsets in LDS — the entity stage's
// four hand-overs between workgroups (chunk ticket, count look-back, "all chunks in", the records' way through memory
// and back) are then no part of the cull's chain of dependent round trips; a tile's records are rebuilt from the
// offsets by bisection.  Workgroup 0 writes the caller's dispatch buffer.  Pass 2 is excluded: it rewrites the entity
// bitset its own evaluation reads, which only the chunked form (each entity-draw evaluated once) orders.
constexpr uint32_t kFusedLocalChunks = 3;
constexpr uint32_t kFusedLocalEntries = kFusedLocalChunks * kEntityBlock;
struct LocalEntityLds {
    uint32_t smem[kEntityBlock / 64 + 1];
    uint32_t off[1024]; // exclusive record offsets of the entity-draws; entries past the last one hold 0xFFFFFFFF
    OrbitMeshletDispatch proto[kFusedLocalEntries];
};
static_assert(kFusedLocalEntries <= 1024, "ten bisection steps");
struct ChunkedEntityLds {
    EntityChunksLds ent;
    CandLds<kCandRing> ring[kFusedWaves]; // pass 2 only
};

template <int ROWS>
struct FusedLds {
    PlaneLds planes;
    AlphaLds alpha_tab;
    union { // (local: passes 0 / 1 of small scenes; never together with the chunked phase or pass 2's ring)
        ChunkedEntityLds chunked;
        LocalEntityLds local;
    } u;
    WaveTileLds slab[kFusedWaves];
    Payload pay[kFusedWaves][fused_tile_records(ROWS) * 32u];
    uint32_t cmd[kFusedWaves][64 * 7];
    uint32_t word[2];
};

// PASS is a template parameter (it shapes the evaluation); the projection type is read at run time (PROJ = -1 in the
// evaluation's templates: one uniform branch in the cone test).
// LOCAL: the entity stage in every workgroup (see LocalEntityLds) — a kernel of its own: with both forms of the entity
// stage inlined into one kernel the compiler copies the whole parameter block to scratch (6.8 KB per lane, every scalar
// of the evaluation a vector register).
template <int PASS, int ROWS, bool LOCAL>
__device__ __forceinline__ void cull_fused_body(const FusedCullParams &p, FusedLds<ROWS> &S) {
    static_assert(!LOCAL || (ROWS == 2 && PASS != 2), "the local entity stage: small scenes, passes 0 and 1");
    constexpr int PROJ = -1;
    constexpr uint32_t kRecs = fused_tile_records(ROWS); // records per tile
    constexpr uint32_t kFusedCap = kRecs * 32u;          // survivors a tile can have: its payload slab holds them all
    PlaneLds &planes = S.planes;
    AlphaLds &alpha_tab = S.alpha_tab;
    EntityChunksLds &ent_lds = S.u.chunked.ent;
    LocalEntityLds &loc = S.u.local;
    WaveTileLds(&slab)[kFusedWaves] = S.slab;
    CandLds<kCandRing>(&ring)[kFusedWaves] = S.u.chunked.ring;
    Payload(&s_pay)[kFusedWaves][kFusedCap] = S.pay;
    uint32_t(&s_cmd)[kFusedWaves][64 * 7] = S.cmd;
    uint32_t(&s_word)[2] = S.word;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t max_tiles = (p.m.dispatch_capacity + kRecs - 1u) / kRecs;
    const __amdgpu_buffer_rsrc_t SY = rsrc_of(p.sync, kFusedSyncWords * 4u);
    const __amdgpu_buffer_rsrc_t TF = rsrc_of(p.tile_flags, max_tiles * 4u);
    uint8_t *dispatch_buffer = const_cast<uint8_t *>(p.m.dispatch_buffer);
    const FusedEntity E{p.m.ci, p.entity_draw_buffer, p.mesh_infos, p.mesh_side, p.m.entities, p.visibility, p.e_pyr};

    // The parameter block lives in the kernel-argument segment and is read with scalar loads where it is used; a line
    // of it that no wave of the CU has touched yet is a full memory round trip, paid in the middle of whichever phase
    // touches it first.  One word of every 64-B line now, all in flight together under the first ticket's round trip.
    {
        const uint32_t *words = reinterpret_cast<const uint32_t *>(&p);
        uint32_t touch = 0;
#pragma unroll
        for (uint32_t i = 0; i < sizeof(FusedCullParams) / 4u; i += 16u) touch |= words[i];
        asm volatile("" ::"s"(touch));
    }
    planes_to_lds(p.m, planes);
    const AlphaLds *alpha = alpha_table_fill(p.m, alpha_tab) ? &alpha_tab : nullptr;

    constexpr bool local = LOCAL; // (the launcher sends a view here only with ne_chunks <= kFusedLocalChunks)
    uint32_t local_total = 0;
    if constexpr (LOCAL) {
        // ------------------------------------------------------------ every entity-draw, in every workgroup (:104-223)
        const uint32_t draw_count = *reinterpret_cast<const uint32_t *>(E.entity_draw_buffer);
        const uint32_t draw_end = min(draw_count, p.draw_limit);
        // (evaluated first, all chunks: the chunks' chains of dependent gathers — draw -> mesh entry + matrix — are
        // independent of each other and unrolled side by side; the scans with their barriers follow)
        OrbitMeshletDispatch prs[kFusedLocalChunks];
#pragma unroll
        for (uint32_t c = 0; c < kFusedLocalChunks; c++) {
            const uint32_t g = p.draw_first + c * kEntityBlock + threadIdx.x;
            const bool active = c < p.ne_chunks && g < draw_end; // :106
            bool visible;
            prs[c] = entity_eval_one(E, g, active, visible);
        }
#pragma unroll
        for (uint32_t c = 0; c < kFusedLocalChunks; c++) { // (fixed trip count: barriers inside)
            const OrbitMeshletDispatch pr = prs[c];
            const uint32_t n = (pr.meshlet_count + 31u) >> 5; // :210
            uint32_t chunk_total;
            const uint32_t ex = block_exclusive_scan<kEntityBlock / 64>(n, loc.smem, &chunk_total);
            loc.off[c * kEntityBlock + threadIdx.x] = local_total + ex;
            loc.proto[c * kEntityBlock + threadIdx.x] = pr;
            // (cut as the chunked form cuts what a chunk publishes: a corrupt meshlet_count must not wrap the sum)
            local_total = min(local_total + uniform(chunk_total), p.e_dispatch_capacity + 1u);
        }
        for (uint32_t i = kFusedLocalEntries + threadIdx.x; i < 1024u; i += kEntityBlock) loc.off[i] = 0xFFFFFFFFu;
        __syncthreads();
        if (blockIdx.x == 0) { // the caller's dispatch buffer: records, the {x,1,1} header (draw_gen.rs:356-363), the total
            OrbitMeshletDispatch *out_rec = reinterpret_cast<OrbitMeshletDispatch *>(dispatch_buffer + ORBIT_DISPATCH_HEADER);
            for (uint32_t r = threadIdx.x; r < min(local_total, p.e_dispatch_capacity); r += kEntityBlock) {
                uint32_t lo = 0;
#pragma unroll
                for (uint32_t step = 512u; step >= 1u; step >>= 1) lo += loc.off[lo + step] <= r ? step : 0u;
                out_rec[r] = entity_record(loc.proto[lo], r - loc.off[lo], 5u);
            }
            if (threadIdx.x == 0) {
                uint32_t *hdr = reinterpret_cast<uint32_t *>(dispatch_buffer);
                hdr[0] = min(local_total, p.e_dispatch_capacity), hdr[1] = 1u, hdr[2] = 1u;
                *p.e_total = local_total;
                if (local_total > p.e_dispatch_capacity) latch_status(p.m.status, ORBIT_E_CAPACITY);
                if (min(local_total, min(p.e_dispatch_capacity, p.m.dispatch_capacity)) == 0u) {
                    *reinterpret_cast<uint32_t *>(p.m.draw_buffer) = 0u; // no tile will run: the empty draw list
                    *p.m.total = 0u;
                }
            }
        }
    } else {
        // ------------------------------------------------------------ entity chunks (entity_cull.comp:104-245)
        EntityChunksArgs a;
        a.dispatch_buffer = dispatch_buffer;
        a.dispatch_capacity = p.e_dispatch_capacity;
        a.draw_first = p.draw_first, a.draw_limit = p.draw_limit;
        a.ne_chunks = p.ne_chunks;
        a.rec_shift = 5u; // (contexts of another dispatch size take the launch chain: abi.hip)
        a.sync = p.sync, a.ent_flags = p.ent_flags, a.total = p.e_total, a.status = p.m.status;
        entity_chunks(E, a, ent_lds, [&](uint32_t total) {
            if (min(total, min(p.e_dispatch_capacity, p.m.dispatch_capacity)) == 0u) {
                // no tile will run: the empty draw list is written here
                *reinterpret_cast<uint32_t *>(p.m.draw_buffer) = 0u;
                *p.m.total = 0u;
            }
        });
    }

    // ---------------------------------------------------------------- every record is in memory
    // (the first tile tickets are drawn before the wait: their round trip runs under the poll; workgroups that had
    // entity chunks arrive last and draw the highest tickets, so no low tile waits behind a busy workgroup)
    // ONE atomic per workgroup for its waves' first tickets (a returning atomic on one word serves ~88 per us on this
    // chip: a ticket per wave of a few hundred workgroups was microseconds of queueing in front of every tile)
    if (threadIdx.x == 0 && local) { // nothing to wait for: the records are this workgroup's own
        s_word[0] = __hip_atomic_fetch_add(p.sync + kSyTileTicket, kFusedWaves, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_word[1] = local_total;
    } else if (threadIdx.x == 0) {
        s_word[0] = __hip_atomic_fetch_add(p.sync + kSyTileTicket, kFusedWaves, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        auto w = __builtin_amdgcn_raw_buffer_load_b64(SY, kSyEntDone * 4u, 0, kSc1);
        for (uint32_t spins = 0; arrived_chunks(w) < p.ne_chunks; spins++) {
            if (spins == kSpinLimit) {
                latch_status(p.m.status, ORBIT_E_TIMEOUT);
                break;
            }
            poll_pause();
            w = __builtin_amdgcn_raw_buffer_load_b64(SY, kSyEntDone * 4u, 0, kSc1);
        }
        s_word[1] = arrived_records(w);
    }
    __syncthreads();
    uint32_t t_next = uniform(s_word[0]) + (uint32_t)wave;
    const uint32_t nrec = min(uniform(s_word[1]), min(p.e_dispatch_capacity, p.m.dispatch_capacity));
    FUSED_MARK(4, nrec);
    const uint32_t ntiles = (nrec + kRecs - 1u) / kRecs;
    const __amdgpu_buffer_rsrc_t REC_R = rsrc_of(p.m.dispatch_buffer + ORBIT_DISPATCH_HEADER, nrec * 16u);

    // ---------------------------------------------------------------- tiles (meshlet_cull.comp:108-255)
    const bool occ2 = PASS == 2 && p.m.ci.meshlet_visibility_buffer != ORBIT_NONE;
    WaveTileLds &L = slab[wave];
    Payload *pay = s_pay[wave];
    uint32_t *cmd = s_cmd[wave];
    uint32_t *out = reinterpret_cast<uint32_t *>(p.m.draw_buffer + ORBIT_DRAW_HEADER);
    for (;;) {
        const uint32_t t = uniform(t_next);
        FUSED_MARK(5, t);
        if (t >= ntiles) break;
        // the tile's 16 records (lane = record * 4 + matrix column; past the last record: zeros), matrices, slab
        const uint32_t rec_index = t * kRecs + ((uint32_t)lane >> 2);
        const bool has_rec = ((uint32_t)lane >> 2) < kRecs && rec_index < nrec;
        uint4 rec;
        if constexpr (LOCAL) { // from the offsets in LDS: the last entity-draw whose exclusive offset is <= the record's index
            uint32_t lo = 0;
#pragma unroll
            for (uint32_t step = 512u; step >= 1u; step >>= 1) lo += loc.off[lo + step] <= rec_index ? step : 0u;
            const OrbitMeshletDispatch r = entity_record(loc.proto[lo], rec_index - loc.off[lo], 5u);
            rec = has_rec ? make_uint4(r.entity_index, r.meshlet_offset, r.meshlet_count, r.visibility_offset)
                          : make_uint4(0u, 0u, 0u, 0u);
        } else {
            const auto rw = __builtin_amdgcn_raw_buffer_load_b128(REC_R, has_rec ? rec_index * 16u : kNoOffset, 0, kSc1);
            rec = make_uint4(rw[0], rw[1], rw[2], rw[3]);
        }
#if defined(ORBIT_FUSED_DEBUG) && ORBIT_FUSED_DEBUG == 3
        FUSED_MARK(0, t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FUSED_MARK(1, t);
#endif
        // The rows need the records only (meshlet offset and count), not the matrices: records into the slab first, then
        // the rows' loads and the matrix loads go out TOGETHER — one memory round trip instead of two (in the streaming
        // evaluation a software pipeline hides this dependence; a one-tile wave has nothing to hide it behind).
        if ((lane & 3) == 0) L.r[lane >> 2].rec = rec;
        wave_lds_fence();
        // all eight rows at once
        RowRegs<ROWS> q;
        rows_load<PASS, 0, ROWS>(p.m, L, lane, q);
        const float4 mat = setup_load_mat(p.m, rec, lane);
        setup_write(p.m, L, rec, mat, lane);
#if defined(ORBIT_FUSED_DEBUG) && ORBIT_FUSED_DEBUG == 3
        FUSED_MARK(2, t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FUSED_MARK(3, t);
#endif
        if (lane == 0) t_next = ticket(p.sync + kSyTileTicket); // the next one, under this tile's evaluation
        CandState cs; // cs.total = the tile's survivor count
        if (occ2) {
            if constexpr (PASS == 2) {
#define ORBIT_FUSED_ROW(R)                                                                                             \
    if constexpr (R < ROWS) {                                                                                          \
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
            cs.total = rows_eval<PASS, PROJ, 0, ROWS, false, kFusedCap, false>(p.m, L, planes, alpha, lane, q, 0u, pay);
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
        if (threadIdx.x < kFusedSyncWords) p.sync[threadIdx.x] = 0u;
    }
}

// blockIdx.y = view (orbit_cull_views: one or several), every view on flags of its own.  The parameter blocks travel by
// value in the kernel-argument segment and are read at a uniform offset (scalar loads, as needed): the single-view
// form with the whole block in scalar registers spilled 67-86 of them.
template <int ROWS>
__global__ __launch_bounds__(kEntityBlock, 2) void cull_fused_views_kernel(const FusedCullViews all, const ViewGroup g) {
    __shared__ FusedLds<ROWS> lds;
    const FusedCullParams &p = all.v[g.idx[blockIdx.y]];
    switch (p.m.ci.occlusion_pass) { // uniform per workgroup
    case 0: cull_fused_body<0, ROWS, false>(p, lds); break;
    case 1: cull_fused_body<1, ROWS, false>(p, lds); break;
    default: cull_fused_body<2, ROWS, false>(p, lds); break;
    }
}
// views of passes 0 / 1 with at most kFusedLocalChunks chunks of entity-draws (the launcher's grouping)
__global__ __launch_bounds__(kEntityBlock, 2) void cull_fused_local_views_kernel(const FusedCullViews all, const ViewGroup g) {
    __shared__ FusedLds<2> lds;
    const FusedCullParams &p = all.v[g.idx[blockIdx.y]];
    if (p.m.ci.occlusion_pass == 0) cull_fused_body<0, 2, true>(p, lds);
    else cull_fused_body<1, 2, true>(p, lds);
}

// Workgroups of one view: a wave per expected tile (a record per entity-draw or so), a workgroup per entity chunk at
// least, never more than a CU holds at once (3 per CU with the 2-row tiles' 43 KB of LDS, 2 with the 8-row tiles' 61 KB),
// shared among the `views` of the launch: the tickets make any number right, this one makes the tiles one round.
uint32_t fused_grid(uint32_t entity_draw_count, uint32_t rows, uint32_t num_cus, uint32_t views) {
    const uint32_t chunks = (entity_draw_count + kEntityBlock - 1u) / kEntityBlock;
    // a tile is 2 x rows records, and scenes have about a record per entity-draw, fewer after the entity test
    const uint32_t waves = (entity_draw_count + 2u * rows - 1u) / (2u * rows);
    const uint32_t cap = max(num_cus * (rows == 2u ? 3u : 2u) / max(views, 1u), 1u);
    return max(min(max(chunks, (waves + kFusedWaves - 1u) / kFusedWaves), cap), 1u);
}

} // namespace

// orbit_cull_views: the n views of `all` — ONE launch for all of them (two when views of up to and of more than
// kFusedSmallEntityDraws entity-draws meet in one call: they take different tile sizes)
hipError_t launch_cull_fused_views(const FusedCullViews &all, const uint32_t *entity_draw_counts, uint32_t n,
                                   uint32_t num_cus, hipStream_t s) {
#if !ORBIT_CONTRACT
    if (all.v[0].m.arith != 0u) return launch_cull_fused_views_contracted(all, entity_draw_counts, n, num_cus, s);
#endif
    // three shapes: 0 = the local entity stage (passes 0 / 1, at most kFusedLocalChunks chunks: 2-row tiles), 1 = 2-row
    // tiles, 2 = 8-row tiles
    for (uint32_t shape = 0; shape < 3u; shape++) {
        ViewGroup g{};
        uint32_t m = 0, most = 0;
        for (uint32_t j = 0; j < n; j++) {
#ifdef ORBIT_FUSED_FORCE_ROWS
            const uint32_t want = ORBIT_FUSED_FORCE_ROWS == 2 ? 1u : 2u;
#else
            const uint32_t chunks = (entity_draw_counts[j] + kEntityBlock - 1u) / kEntityBlock;
#ifdef ORBIT_FUSED_NO_LOCAL // (A/B builds: the chunked entity stage for every view)
            const uint32_t local_max = 0u;
#else
            const uint32_t local_max = kFusedLocalChunks;
#endif
            const uint32_t want = (chunks <= local_max && all.v[j].m.ci.occlusion_pass != 2u) ? 0u
                                  : entity_draw_counts[j] <= kFusedSmallEntityDraws                    ? 1u
                                                                                                       : 2u;
#endif
            if (want == shape) {
                g.idx[m++] = (uint8_t)j;
                most = max(most, entity_draw_counts[j]);
            }
        }
        if (m == 0) continue;
        const uint32_t rows = shape == 2u ? 8u : 2u;
        const dim3 grid(fused_grid(most, rows, num_cus, m), m), block(kEntityBlock);
        if (shape == 0u) hipLaunchKernelGGL(cull_fused_local_views_kernel, grid, block, 0, s, all, g);
        else if (shape == 1u) hipLaunchKernelGGL((cull_fused_views_kernel<2>), grid, block, 0, s, all, g);
        else hipLaunchKernelGGL((cull_fused_views_kernel<8>), grid, block, 0, s, all, g);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

} // namespace orbit
