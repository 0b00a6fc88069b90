// meshlet_eval.hip — the evaluation launches of meshlet_cull (shaders/meshlet_cull.comp:108-213,233-254 of the
// reference repo): which meshlets of the dispatched records are drawn.
//
//   pass 0 / 2 : meshlet_eval_kernel streams every meshlet of every record; in
//                pass 2 the HiZ test runs on candidates compacted in LDS.
//   pass 1     : meshlet_pass1_kernel is list-driven — only meshlets whose
//                last-frame bit is set are read at all.
//   streams    : with derived meshlet streams bound (orbit_meshlet_stream_*; kernels.h
//                MeshletStreamView) passes 0 / 2 read 20.25 - 22 B per meshlet instead of
//                the 32-B Meshlet and keep no survivor payload (template flags STREAM, CLS).
// Outputs per wave tile: the survivor count, the should-draw ballots and (Meshlet buffer) the survivors' payload —
// what scan + emit (meshlet_emit.hip) or the list launches (meshlet_lists.hip) turn into the caller's buffers.
#include "entity_common.h"
#include "meshlet_common.h"

namespace orbit {

namespace {

// ---------------------------------------------------------------------------
// eval launch -> scan -> emit launch.
//
//   meshlet_eval : streams records, matrices and meshlets through the software
//                  pipeline below and writes, per wave tile, the 8 should-draw
//                  ballots (64 B) and their popcount.  No inter-wave dependency
//                  of any kind: tiles are assigned statically (grid stride), so
//                  waves drift out of phase and one wave's evaluation overlaps
//                  another's loads.  Pass-2 visibility words are written here.
//   scan         : exclusive prefix over the tile counts (scan.h / entity_cull.hip).
//   meshlet_emit : per tile, re-reads only the survivors' command words and
//                  writes the 28-B commands at the scanned offsets.
//
// Software pipeline of one wave in meshlet_eval.  A tile's 8 rows go through two
// single-row register sets A/B, so that one row's meshlet loads are in flight
// during every evaluation (16 waves per CU x 2 KiB is more than the bytes in
// flight that HBM latency x bandwidth asks of a CU):
//     eval row 0 (A) ; load A <- row 2      eval row 1 (B) ; load B <- row 3
//     eval row 2 (A) ; load A <- row 4      eval row 3 (B) ; load B <- row 5
//     issue matrix loads (i+2), record load (i+3)      | small, one iteration ahead
//     eval row 4 (A) ; load A <- row 6      eval row 5 (B) ; load B <- row 7
//     eval row 6 (A) ; load A <- row 0 of i+1   eval row 7 (B) ; load B <- row 1 of i+1
//     write slab(i+2) ; store masks(i), count(i)
// The dependent chain record -> matrix -> meshlets of a tile is spread over three
// iterations; all loads are unconditional (inactive lanes read a zero page) so
// that the compiler's vmcnt accounting stays exact.
constexpr int kEvWaves = 4;
constexpr int kEvWavesPerSimd = 4; // resident waves per SIMD: compiled for (128 VGPRs) and launched at
// Pass 0 from the derived streams needs fewer registers than the passes with the HiZ test (84 against 107-119) and is
// compiled for / launched at five waves per SIMD: with the rows as nt loads the fifth wave is worth 3.5 % (0.1948 /
// 0.1964 -> 0.1879 / 0.1897 ms; six: 0.1897 / 0.1909, three: 0.2175 / 0.2184) — with the default cache policy it was
// worth nothing, and from the Meshlet buffer it costs 1.5 % (0.3470 -> 0.3525; r03_notes.md).
#ifndef ORBIT_EV_WPS0
#define ORBIT_EV_WPS0 5
#endif
constexpr int ev_waves_per_simd(int pass, bool stream) { return (pass == 0 && stream) ? ORBIT_EV_WPS0 : kEvWavesPerSimd; }


// The emit launch behind this evaluation scans the tiles' counts itself and publishes each chunk's sum as a non-zero
// word of chunk_sums (meshlet_emit.hip emit_scan_duty): the first workgroup of every evaluation zeroes them.
__device__ __forceinline__ void clear_chunk_sums(const MeshletCullParams &p) {
    if (blockIdx.x != 0) return;
    const uint32_t max_tiles = (p.dispatch_capacity + kTileRecords - 1) / kTileRecords;
    const uint32_t max_chunks = (max_tiles + kScanChunk - 1) / kScanChunk;
    for (uint32_t c = threadIdx.x; c < max_chunks; c += kEvWaves * 64) p.chunk_sums[c] = 0u;
    if (threadIdx.x < kEmitTicketPools) p.tickets[(kTicketPools + threadIdx.x) * kTicketStride] = 0u; // its group tickets
}

// STREAM: meshlets from the derived streams; CLS (with STREAM): predicate bits from the streams' alpha classes
// LIST: the instantiation can write the record list (visible_list == 2).  Only the several-views kernel has it: it
// reads its parameter block from the argument segment where it needs it, while the single-view kernels hold theirs
// in scalar registers, and the list's handful of extra words there cost the pass-0 kernels 13 v_readlane per tile
// (18 spilled SGPRs instead of 2) whether a cull writes a list or not.  A cull with a list is launched as one view.
//
// FUSED (with LIST; the shard launch of the sharded engine, shard_cull_kernel below): the ENTITY STAGE runs inside this
// launch, in front of the tiles — the one-launch form of entity_cull (entity_common.h entity_chunks: chunks drawn by
// ticket, counts published, look-back, records stored write-through, arrival) by the first workgroups, while every
// workgroup sets up and then waits for "all chunks in" (one 8-byte word: chunks << 32 | records).  A shard's step is
// then ONE launch where it was entity launch + evaluation + two list launches: at a 1/8 shard of config 5 the launches'
// fixed costs (each ~4 us of dispatch and drain around its waves' life, a boundary between them) were a third of the
// step.  What a workgroup waits for here is drawn by whoever runs (tickets), and a workgroup whose patience runs out
// draws entity tickets itself: nothing depends on dispatch order or residency (handoff.h).  Records are read with sc1
// loads (another workgroup of this launch stored them), never before the arrival word says they are there.
constexpr uint32_t kShardPatience = 256; // polls of the arrival word before a waiting workgroup helps with the chunks

template <bool FUSED>
__device__ __forceinline__ uint4 eval_load_rec(const MeshletCullParams &p, uint32_t tile, bool tile_valid, uint32_t nrec, int lane) {
    if constexpr (FUSED) {
        const __amdgpu_buffer_rsrc_t R = rsrc_of(p.dispatch_buffer + ORBIT_DISPATCH_HEADER, nrec * 16u);
        const uint32_t rec_index = tile * kTileRecords + (uint32_t)(lane >> 2);
        const auto w = __builtin_amdgcn_raw_buffer_load_b128(R, (tile_valid && rec_index < nrec) ? rec_index * 16u : kNoOffset, 0, kSc1);
        return make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        return setup_load_rec(p, tile, tile_valid, nrec, lane);
    }
}

template <int PASS, int PROJ, bool PROF, bool STREAM = false, bool CLS = false, bool LIST = false, bool FUSED = false>
__device__ __forceinline__ void meshlet_eval_body(const MeshletCullParams &p, const FusedCullParams *fp = nullptr) {
    static_assert(!FUSED || LIST, "the shard launch writes the record list");
    __shared__ std::conditional_t<FUSED, EntityChunksLds, uint32_t> ent_lds;
    __shared__ uint32_t s_word[2];
    __shared__ WaveTileLds lds[kEvWaves][3];
    __shared__ PlaneLds planes;
    __shared__ AlphaLds alpha_tab;
    __shared__ CandLds<(PASS == 2 ? kCandRing : 1u)> cand_ring[kEvWaves]; // pass 2 only
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef ORBIT_TRIAGE
    const unsigned long long wall_enter = wall_clock64(); // (ORBIT_SP_DEBUG & 2048: the wave's begin stamp is this one)
    unsigned long long wall_chunks = 0;
#endif
    // pass 2 with meshlet occlusion: HiZ tests run on compacted candidates (cand_flush), not per row
    const bool occ2 = PASS == 2 && p.ci.meshlet_visibility_buffer != ORBIT_NONE;
    __shared__ uint32_t cls_sel[CLS ? 8 : 1]; // [2 c + k] = all ones if alpha class c has predicate bit k (alpha_bits)
    __shared__ uint32_t s_red[kEvWaves];      // records_finish
    // The launch's ramp is a chain of dependent round trips — record count -> records -> matrices -> rows — and on a
    // shard of the sharded engine (three tiles per wave) it is a fifth of a wave's life.  Two of the hops go: the
    // records of a wave's first three tiles (always static, below) are requested TOGETHER with the count, bounded by
    // the buffer's capacity instead of the count and cut to the count when it is there; and the first rows go out with
    // the first matrices (the rows need a record's meshlet offset and count, not its matrix).
    uint32_t nrec_raw = 0;
    if constexpr (!FUSED) nrec_raw = *reinterpret_cast<const uint32_t *>(p.dispatch_buffer);
    const uint32_t stride = gridDim.x * kEvWaves, wave_g = blockIdx.x * kEvWaves + wave;
    uint4 r0, r1, rec2;
    if constexpr (!FUSED) {
        const uint32_t cap_tiles = (p.dispatch_capacity + kTileRecords - 1) / kTileRecords;
        r0 = setup_load_rec(p, wave_g, wave_g < cap_tiles, p.dispatch_capacity, lane);
        r1 = setup_load_rec(p, wave_g + stride, wave_g + stride < cap_tiles, p.dispatch_capacity, lane);
        rec2 = setup_load_rec(p, wave_g + 2u * stride, wave_g + 2u * stride < cap_tiles, p.dispatch_capacity, lane);
    }
    planes_to_lds(p, planes);
    clear_chunk_sums(p);
    if (threadIdx.x < kEvWaves) s_red[threadIdx.x] = 0u;
    if (CLS && threadIdx.x < 6) cls_sel[threadIdx.x] = ((alpha_bits(p.ci, threadIdx.x >> 1) >> (threadIdx.x & 1u)) & 1u) ? ~0u : 0u;
    const AlphaLds *alpha = alpha_table_fill(p, alpha_tab) ? &alpha_tab : nullptr;
    __syncthreads();
    if constexpr (FUSED) {
        // ------------------------------------------------------------ the entity stage (entity_cull.comp:104-245)
        const FusedCullParams &f = *fp;
        const FusedEntity E{p.ci, f.entity_draw_buffer, f.mesh_infos, f.mesh_side, p.entities, f.visibility, f.e_pyr};
        EntityChunksArgs a;
        a.dispatch_buffer = const_cast<uint8_t *>(p.dispatch_buffer);
        a.dispatch_capacity = f.e_dispatch_capacity;
        a.draw_first = f.draw_first, a.draw_limit = f.draw_limit;
        a.ne_chunks = f.ne_chunks;
        a.rec_shift = 5u; // (orbit_cull_shard refuses other dispatch sizes)
        a.sync = f.sync, a.ent_flags = f.ent_flags, a.total = f.e_total, a.status = p.status;
        // whoever completes the LAST chunk tells everybody: a copy of {1 << 40 | records} on each of kShardDoneWords
        // lines (thread 0 of that workgroup; write-through).  The waiting workgroups poll a copy each, not the arrival
        // word the chunks' atomics need.
        const __amdgpu_buffer_rsrc_t DN = rsrc_of(f.done_flags, kShardDoneWords * kShardDoneStride * 4u);
        auto tell_all = [&](uint32_t total) {
            typedef unsigned int v2u __attribute__((ext_vector_type(2)));
            const v2u v = {total, 1u << 8};
            for (uint32_t k = 0; k < kShardDoneWords; k++)
                __builtin_amdgcn_raw_buffer_store_b64(v, DN, k * kShardDoneStride * 4u, 0, kSc1);
        };
        // the first workgroups take the chunks (tickets: whichever of them run); everybody else goes to wait
        if (blockIdx.x < f.ne_chunks) entity_chunks(E, a, ent_lds, tell_all);
#ifdef ORBIT_TRIAGE
        wall_chunks = wall_clock64(); // (ORBIT_SP_DEBUG & 4096: the wave's begin stamp is "its workgroup's chunks are done")
#endif
        const uint32_t my_copy = (blockIdx.x % kShardDoneWords) * kShardDoneStride * 4u;
        for (uint32_t round = 0;; round++) {
            if (threadIdx.x == 0) {
                if (round == 0u && blockIdx.x >= f.ne_chunks) asm volatile("s_sleep 64" ::: "memory"); // a chunk takes microseconds
                auto w = __builtin_amdgcn_raw_buffer_load_b64(DN, my_copy, 0, kSc1);
                for (uint32_t spins = 0; arrived_chunks(w) == 0u && spins < kShardPatience; spins++) {
                    asm volatile("s_sleep 8" ::: "memory");
                    w = __builtin_amdgcn_raw_buffer_load_b64(DN, my_copy, 0, kSc1);
                }
                s_word[0] = arrived_chunks(w) != 0u ? 1u : 0u;
                s_word[1] = arrived_records(w);
            }
            __syncthreads();
            const uint32_t done = uniform(s_word[0]);
            nrec_raw = uniform(s_word[1]);
            __syncthreads();
            if (done != 0u) break;
            if (round == kSpinLimit / kShardPatience) {
                if (threadIdx.x == 0) latch_status(p.status, ORBIT_E_TIMEOUT);
                nrec_raw = 0u; // nothing is evaluated from records that may not be there
                break;
            }
            // patience ran out: whoever holds the missing chunks may not be running — draw them here
            entity_chunks(E, a, ent_lds, tell_all);
        }
        nrec_raw = min(nrec_raw, f.e_dispatch_capacity);
    }
    const uint32_t nrec = min(nrec_raw, p.dispatch_capacity);
    const uint32_t ntiles = (nrec + kTileRecords - 1) / kTileRecords;
    // Tile assignment: a static grid-stride share first, tickets for the rest.  With 23 or 24 tiles per wave handed out
    // statically the waves finish over a span of ~45 us (the SIMD arbiter favours older waves, CUs differ) and the last
    // 6 % of the kernel run at falling occupancy (tools/wave_tail.py); handing every tile out through an atomic ticket
    // removes the tail but costs more than it saves (+5 %: the returning atomic sits in the in-order vector-memory
    // queue in front of the row loads, once per tile).  So every wave takes its first S tiles statically (tile
    // k * stride + g, as before) and only the last ~3 tiles' worth per wave go through tickets — 3-4 atomics per wave:
    // fast waves draw more of them.  One counter cannot serve the ~290 claims/us at the very end (a returning atomic
    // on one address saturates near 88/us on MI355X), so there are kTicketPools: wave g draws from pool g % npools,
    // ticket n of pool q is tile S * stride + n * npools + q; every pool is drained by 1/npools of the waves of all
    // XCDs, so the pools run dry within a tile's time of each other and nobody steals.  A claim is made three tiles
    // before its rows are read (records -> matrices -> slab -> rows); a wave whose ticket is past the end drains its
    // pipeline and leaves.
    const bool static_share = ORBIT_DBG(p, 64u); // triage: everything static, for A/B runs
    const uint32_t full_rounds = ntiles / stride;
    // ticketed rounds: three on a long launch, fewer on a short one (a shard of the sharded engine has 2-3 rounds in all)
    const uint32_t dyn_rounds = ORBIT_DBG(p, 256u) ? 2u : ORBIT_DBG(p, 512u) ? 4u : ORBIT_DBG(p, 1024u) ? 1u : min(max(full_rounds / 4u, 1u), 3u);
    // a scene of at most one tile per wave has nothing to balance: static, no atomic in front of the first loads
    // ... and the three tiles a wave claims before its first row never go through a ticket either: their records,
    // matrices and rows are the launch's ramp, and an atomic in front of them is one more round trip before anything
    // streams (a 1/8 shard of config 5 is three tiles per wave)
    const uint32_t n_static = (static_share || ntiles <= stride) ? 0xFFFFFFFFu : (ORBIT_DBG(p, 128u) ? 3u : max(full_rounds >= dyn_rounds ? full_rounds - dyn_rounds : 0u, 3u));
    const uint32_t npools = min((uint32_t)kTicketPools, stride);
    const uint32_t pool = wave_g % npools;
    uint32_t *ticket_ctr = p.tickets + pool * kTicketStride;
    uint32_t claims = 0; // claims made so far by this wave
    auto claim = [&]() -> uint32_t { // lane 0's VGPR carries the claim; not read before ticket_tile()
        uint32_t raw = claims;
        if (claims >= n_static) {
            raw = 0;
            if (lane == 0) raw = atomicAdd(ticket_ctr, 1u);
        }
        claims++;
        return raw;
    };
    auto ticket_tile = [&](uint32_t raw, uint32_t k) -> uint32_t { // k = index of the claim that returned `raw`
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)raw);
        if (k < n_static) return t * stride + wave_g;
        return t >= 0x08000000u ? 0xFFFFFFFFu : n_static * stride + t * npools + pool;
    };
    StreamRsrc SR = {};
    if constexpr (STREAM) {
        SR.first = p.ms.first;
        SR.count = p.ms.count;
        SR.sphere = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(p.ms.sphere + p.ms.first), 0, p.ms.count * 16u, kBufFlags);
        SR.cone = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.ms.cone + p.ms.first), 0, p.ms.count * 4u, kBufFlags);
        if constexpr (!CLS)
            SR.mat = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.ms.mat + p.ms.first), 0, p.ms.count * 2u, kBufFlags);
    }
    uint32_t w0, w1, w2, tk_raw;
    {
        const uint32_t t0 = claim(), t1 = claim(), t2 = claim();
        tk_raw = claim();
        w0 = ticket_tile(t0, 0u);
        w1 = ticket_tile(t1, 1u);
        w2 = ticket_tile(t2, 2u);
    }
    // two single-row register sets: while one row is evaluated the next row's loads are in flight
    // (STREAM: a row is 1.25 KiB instead of 2 KiB; sets of two rows each were measured in rounds 2 and 3 — neutral or
    // slower, as are 5 and 6 resident waves per SIMD: profiles/r02_notes.md, r03_notes.md)
    RowRegs<1> qa, qb;
    {
        // the records requested before the count was known: a lane past the count holds "no record" (zeros), as if it
        // had read the zero page (w0 .. w2 are the static tiles wave_g + k * stride: n_static >= 3)
        if constexpr (FUSED) { // the records exist now: loaded here, past the caches
            r0 = eval_load_rec<true>(p, w0, w0 < ntiles, nrec, lane);
            r1 = eval_load_rec<true>(p, w1, w1 < ntiles, nrec, lane);
            rec2 = eval_load_rec<true>(p, w2, w2 < ntiles, nrec, lane);
        } else {
            const uint4 none = make_uint4(0u, 0u, 0u, 0u);
            const uint32_t rl = (uint32_t)lane >> 2;
            if (!(w0 < ntiles && w0 * kTileRecords + rl < nrec)) r0 = none;
            if (!(w1 < ntiles && w1 * kTileRecords + rl < nrec)) r1 = none;
            if (!(w2 < ntiles && w2 * kTileRecords + rl < nrec)) rec2 = none;
        }
        // tile 0's records into its slab, so that its first rows can go out with the matrices (setup_write stores
        // them again, with everything else)
        if ((lane & 3) == 0) lds[wave][0].r[lane >> 2].rec = r0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (the matrices first: the slab writes below wait for them, with the rows still in flight behind)
        const float4 m0 = setup_load_mat(p, r0, lane);
        const float4 m1 = setup_load_mat(p, r1, lane);
        uint32_t c0 = 0u, c1 = 0u;
        if constexpr (CLS) c0 = setup_load_cls(p, r0, lane), c1 = setup_load_cls(p, r1, lane);
        rows_load<PASS, 0, 1, STREAM, CLS>(p, lds[wave][0], lane, qa, &SR);
        rows_load<PASS, 1, 1, STREAM, CLS>(p, lds[wave][0], lane, qb, &SR);
        if constexpr (CLS) {
            setup_write_cls(lds[wave][0], cls_sel, r0, c0, lane);
            setup_write_cls(lds[wave][1], cls_sel, r1, c1, lane);
        }
        setup_write(p, lds[wave][0], r0, m0, lane);
        setup_write(p, lds[wave][1], r1, m1, lane);
    }
    uint32_t it = 0;
#ifdef ORBIT_TRIAGE
    unsigned long long tc[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, t1;
#define EV_STAMP(k)                            \
    if (PROF && !(p.debug_flags & 16u)) {      \
        t1 = __builtin_amdgcn_s_memtime();     \
        __builtin_amdgcn_s_waitcnt(0xC07F);    \
        tc[k] += t1 - t0;                      \
        t0 = t1;                               \
    }
    unsigned long long wall_begin = 0;
    if (p.debug_flags & 16u) wall_begin = (p.debug_flags & 2048u) ? wall_enter : (p.debug_flags & 4096u) ? wall_chunks : wall_clock64(); // per-wave life stamps (tools/wave_tail.py)
    if (PROF) {
        t0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
#else
#define EV_STAMP(k)
    static_assert(!PROF, "phase stamps: -DORBIT_TRIAGE builds only");
#endif
    // The SIMD's arbiter serves the oldest wave first, and the four waves of a SIMD belong to the four blocks a CU
    // holds, dispatched one after the other: left alone, the first block's waves finish at half the kernel's
    // duration and the rest of the kernel runs at 3, 2, 1 waves per SIMD (tools/wave_tail.py).  Every wave
    // therefore rotates its priority once per tile, offset by its block's dispatch rank, so that over any four
    // tiles each wave of a SIMD has held each priority once.
    const uint32_t prio_rank = blockIdx.x / max(gridDim.x / (uint32_t)ev_waves_per_simd(PASS, STREAM), 1u);
    while (w0 < ntiles) {
        if (!ORBIT_DBG(p, 32u)) {
            switch ((it + prio_rank) & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
            }
        }
        const uint32_t w3 = ticket_tile(tk_raw, claims - 1u); // claimed one tile ago
        tk_raw = claim();                                     // the tile after it
        WaveTileLds &L = lds[wave][it % 3];
        WaveTileLds &Ln = lds[wave][(it + 1) % 3]; // slab of an absent tile holds zero records
        // no payload with the derived streams: the emit launch derives the survivors' command words itself
        Payload *payload = (STREAM || p.task_records || (p.visible_list == 2u && p.also_commands == nullptr))
                               ? nullptr : p.tile_payload + (size_t)w0 * kPayloadCap;
        CandState cs; // cs.total = the tile's running survivor count
#define EV_ROW(R, Q)                                                                                                  \
    if constexpr (PASS == 2) {                                                                                        \
        if (occ2) row_eval_occ<PROJ, R, kCandRing, CLS>(p, L, planes, alpha, cand_ring[wave], lane, Q, cs, payload);  \
        else cs.total = rows_eval<PASS, PROJ, R, 1, CLS>(p, L, planes, alpha, lane, Q, cs.total, payload);      \
    } else {                                                                                                          \
        cs.total = rows_eval<PASS, PROJ, R, 1, CLS>(p, L, planes, alpha, lane, Q, cs.total, payload);           \
    }
        EV_ROW(0, qa)
        rows_load<PASS, 2, 1, STREAM, CLS>(p, L, lane, qa, &SR);
        EV_ROW(1, qb)
        rows_load<PASS, 3, 1, STREAM, CLS>(p, L, lane, qb, &SR);
        EV_STAMP(0)
        EV_ROW(2, qa)
        rows_load<PASS, 4, 1, STREAM, CLS>(p, L, lane, qa, &SR);
        EV_ROW(3, qb)
        rows_load<PASS, 5, 1, STREAM, CLS>(p, L, lane, qb, &SR);
        const float4 mat2 = setup_load_mat(p, rec2, lane);
        uint32_t cls2 = 0u;
        if constexpr (CLS) cls2 = setup_load_cls(p, rec2, lane);
        const uint4 rec3 = eval_load_rec<FUSED>(p, w3, w3 < ntiles, nrec, lane);
        EV_STAMP(1)
        EV_ROW(4, qa)
        rows_load<PASS, 6, 1, STREAM, CLS>(p, L, lane, qa, &SR);
        EV_ROW(5, qb)
        rows_load<PASS, 7, 1, STREAM, CLS>(p, L, lane, qb, &SR);
        EV_STAMP(2)
        EV_ROW(6, qa)
        rows_load<PASS, 0, 1, STREAM, CLS>(p, Ln, lane, qa, &SR);
        EV_ROW(7, qb)
        rows_load<PASS, 1, 1, STREAM, CLS>(p, Ln, lane, qb, &SR);
        EV_STAMP(3)
        if constexpr (PASS == 2) {
            if (occ2) tile_finish_occ(p, L, cand_ring[wave], lane, cs, payload);
        }
#undef EV_ROW
        if constexpr (CLS) setup_write_cls(lds[wave][(it + 2) % 3], cls_sel, rec2, cls2, lane);
        setup_write(p, lds[wave][(it + 2) % 3], rec2, mat2, lane); // also fences the mask writes of this tile
        // the ballots are read back only by the task-record launch and, in the emit launch, for tiles with more
        // survivors than the payload holds: the common tile writes its count and nothing else
        // (nor does anything read them behind a record list without commands)
        const bool rec_list = LIST && p.visible_list == 2u;
        if (((STREAM && !(rec_list && p.also_commands == nullptr)) || p.task_records != nullptr || cs.total > kPayloadCap) && lane < 2 * (int)kTileRows)
            p.tile_masks[(size_t)w0 * (2 * kTileRows) + lane] = reinterpret_cast<const uint32_t *>(L.draw_mask)[lane];
        if (lane == 0) p.tile_counts[w0] = cs.total;
        // the record list (visible_list == 2) is written here, an entry per dispatch record at the record's index; the
        // wave's survivors add up in LDS (nothing of the list lives in a register across tiles)
        if (rec_list) tile_records_out(p, L, nrec, w0, cs.total, s_red + wave, lane);
        EV_STAMP(4)
        w0 = w1;
        w1 = w2;
        w2 = w3;
        rec2 = rec3;
        it++;
    }
#undef EV_STAMP
    if constexpr (FUSED) records_finish<kEvWaves>(p, nrec, s_red, fp->ent_flags, fp->ne_chunks, fp->sync, fp->done_flags);
    else if (LIST && p.visible_list == 2u) records_finish<kEvWaves>(p, nrec, s_red);
#ifdef ORBIT_TRIAGE
    if (PROF && lane == 0) {
        for (int k = 0; k < 5; k++) atomicAdd(p.debug_cycles + k, tc[k]);
        atomicAdd(p.debug_cycles + 7, 1ull);
    }
    if ((p.debug_flags & 16u) && lane == 0) {
        const uint32_t wg = blockIdx.x * kEvWaves + wave;
        if (wg < 8192u) {
            p.debug_cycles[32 + 2 * wg] = wall_begin;
            p.debug_cycles[32 + 2 * wg + 1] = wall_clock64();
        }
    }
#endif
}

template <int PASS, int PROJ, bool PROF, bool STREAM = false, bool CLS = false>
__global__ __launch_bounds__(kEvWaves * 64, ev_waves_per_simd(PASS, STREAM)) void meshlet_eval_kernel(const MeshletCullParams p) {
    meshlet_eval_body<PASS, PROJ, PROF, STREAM, CLS>(p);
}
// several views in one launch (orbit_cull_views): the views' parameter blocks by value, blockIdx.y picks one
template <int PASS, int PROJ, bool STREAM, bool CLS = false>
__global__ __launch_bounds__(kEvWaves * 64, ev_waves_per_simd(PASS, STREAM)) void meshlet_eval_views_kernel(const MeshletCullViews all,
                                                                                           const ViewGroup g) {
    meshlet_eval_body<PASS, PROJ, false, STREAM, CLS, true>(all.v[g.idx[blockIdx.y]]);
}

// The shard launch: entity stage + evaluation + record list of ONE view (pass 0).  The parameter block is read from the
// argument segment at a uniform offset like the several-views kernels' (a block held in scalar registers spills).
struct ShardCullArgs {
    FusedCullParams v[2]; // [1] unused: the index is what keeps the block out of the scalar registers
};
template <int PROJ, bool STREAM, bool CLS>
__global__ __launch_bounds__(kEvWaves * 64, kEvWavesPerSimd) void shard_cull_kernel(const ShardCullArgs all, const ViewGroup g) {
    const FusedCullParams &f = all.v[g.idx[blockIdx.y] & 1u];
    meshlet_eval_body<0, PROJ, false, STREAM, CLS, true, true>(f.m, &f);
}

// ---------------------------------------------------------------------------
// Pass 1 (occlusion_pass == 1, :137): only meshlets that were visible last frame
// are processed at all, and which those are is known from 4 bytes per record
// before any meshlet is read.  So pass 1 is list-driven instead of streaming:
// per wave tile the set bits of the 16 visibility words become a candidate list
// in canonical (row, lane) order (LDS), and the tile is evaluated in chunks of
// 64 candidates — gathered 32-B meshlet reads for the candidates only (dense
// runs coalesce exactly like the streamed rows), full lanes in the evaluation.
// At 10 % visibility that is ~40 % of the streamed bytes and one evaluation per
// tile instead of eight.  Outputs (ballots, count, payload) and the scan + emit
// that follow are those of the other passes.
//
// Pipeline of one wave: records 3 tiles ahead, matrices + visibility words 2
// ahead, slab 1 ahead (as in meshlet_eval); the work items are (tile, chunk)
// pairs through two register sets — while item k is evaluated the gathers of
// item k+1 (next chunk, or chunk 0 of the next tile, whose list is built first)
// are in flight.
struct Pass1Item {
    uint4 a, b;    // the candidate's Meshlet
    uint32_t code; // row*64 + lane of the candidate, ~0u = none
};

__device__ __forceinline__ uint32_t pass1_load_prev(const MeshletCullParams &p, const uint4 &rec, int lane) {
    const bool rw_vis = p.ci.meshlet_visibility_buffer != ORBIT_NONE;
    const uint32_t *src = (rw_vis && rec.z != 0u && (lane & 3) == 0) ? p.meshlet_visibility + rec.w
                                                                    : reinterpret_cast<const uint32_t *>(p.zero_page);
    const uint32_t w = *src;
    return rw_vis ? w : 0xFFFFFFFFu; // no meshlet visibility buffer: visible_in_buffer stays true (:128)
}

// Candidate list of a tile: lanes whose last-frame bit is set (:129-137), canonical order.
__device__ __forceinline__ uint32_t pass1_build_list(const WaveTileLds &L, uint16_t *list, int lane) {
    const uint32_t half = lane >> 5, ml = lane & 31;
    uint32_t n = 0;
#pragma unroll
    for (int r = 0; r < (int)kTileRows; r++) {
        const uint32_t rid = 2 * r + half;
        const bool c = ml < L.r[rid].rec.z && ((L.r[rid].prev >> ml) & 1u) != 0;
        const uint64_t mask = __ballot(c);
        if (c) list[n + lane_prefix(mask)] = (uint16_t)(r * 64 + lane);
        n += (uint32_t)__popcll(mask);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return n;
}

__device__ __forceinline__ Pass1Item pass1_load(const MeshletCullParams &p, const WaveTileLds &L,
                                                const uint16_t *list, uint32_t n, uint32_t chunk, int lane) {
    Pass1Item it;
    const uint32_t j = chunk * 64u + (uint32_t)lane;
    const bool valid = j < n;
    it.code = valid ? (uint32_t)list[j] : ~0u;
    const uint32_t rid = (it.code >> 5) & 15u, ml = it.code & 31u;
    // unconditional loads (exact vmcnt accounting): lanes without a candidate read the zero page
    const uint4 *src = valid ? reinterpret_cast<const uint4 *>(p.meshlets + (L.r[rid].rec.y + ml))
                             : reinterpret_cast<const uint4 *>(p.zero_page);
    it.a = src[0];
    it.b = src[1];
    return it;
}

template <int PROJ>
__device__ __forceinline__ uint32_t pass1_eval(const MeshletCullParams &p, WaveTileLds &L, const PlaneLds &P,
                                               const AlphaLds *A, int lane, const Pass1Item &it, uint32_t total,
                                               Payload *payload) {
    const bool valid = it.code != ~0u;
    const uint32_t rid = (it.code >> 5) & 15u, ml = it.code & 31u;
    const uint32_t mi = it.b.w & 0xFFFFu; // zero page for lanes without a candidate: material 0, unused
    const uint32_t abits = A ? (uint32_t)A->bits[mi & (kAlphaTable - 1u)] : alpha_bits(p.ci, p.materials[mi].alpha_mode);
    bool should_draw = false;
    if (valid) {
        Sphere s;
        const bool visible = eval_geometry<1, PROJ>(p, L, P, rid, it.a, it.b, true, s); // candidates have the bit set
        should_draw = should_draw_of(visible, abits, true, false);
    }
    if (should_draw) atomicOr(reinterpret_cast<uint32_t *>(L.draw_mask) + rid, 1u << ml);
    const uint64_t dm = __ballot(should_draw);
    if (payload && should_draw) {
        const uint32_t rank = total + lane_prefix(dm);
        if (rank < kPayloadCap) payload[rank] = make_payload(ml, rid, it.b);
    }
    return total + (uint32_t)__popcll(dm);
}

template <int PROJ, bool LIST = false> // LIST: as in meshlet_eval_body
__device__ __forceinline__ void meshlet_pass1_body(const MeshletCullParams &p) {
    __shared__ WaveTileLds lds[kEvWaves][3];
    __shared__ PlaneLds planes;
    __shared__ AlphaLds alpha_tab;
    __shared__ uint16_t s_list[kEvWaves][2][kTileRows * 64];
    __shared__ uint32_t s_red[kEvWaves]; // records_finish
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    planes_to_lds(p, planes);
    clear_chunk_sums(p);
    if (threadIdx.x < kEvWaves) s_red[threadIdx.x] = 0u;
    const AlphaLds *alpha = alpha_table_fill(p, alpha_tab) ? &alpha_tab : nullptr;
    __syncthreads();
    const uint32_t nrec = min(*reinterpret_cast<const uint32_t *>(p.dispatch_buffer), p.dispatch_capacity);
    const uint32_t ntiles = (nrec + kTileRecords - 1) / kTileRecords;
    const uint32_t stride = gridDim.x * kEvWaves;
    uint32_t w0 = blockIdx.x * kEvWaves + wave, w1 = w0 + stride, w2 = w1 + stride;
    uint16_t *list_cur = s_list[wave][0], *list_next = s_list[wave][1];
    uint4 rec2;
    uint32_t n_cur;
    Pass1Item cur;
    {
        const uint4 r0 = setup_load_rec(p, w0, w0 < ntiles, nrec, lane);
        const uint4 r1 = setup_load_rec(p, w1, w1 < ntiles, nrec, lane);
        rec2 = setup_load_rec(p, w2, w2 < ntiles, nrec, lane);
        const float4 m0 = setup_load_mat(p, r0, lane);
        const uint32_t p0 = pass1_load_prev(p, r0, lane);
        const float4 m1 = setup_load_mat(p, r1, lane);
        const uint32_t p1 = pass1_load_prev(p, r1, lane);
        setup_write(p, lds[wave][0], r0, m0, lane, p0);
        setup_write(p, lds[wave][1], r1, m1, lane, p1);
        n_cur = pass1_build_list(lds[wave][0], list_cur, lane); // absent tile: zero records, empty list
        cur = pass1_load(p, lds[wave][0], list_cur, n_cur, 0u, lane);
    }
    uint32_t it = 0;
    while (w0 < ntiles) {
        const uint32_t w3 = w2 + stride;
        WaveTileLds &L = lds[wave][it % 3];
        WaveTileLds &Ln = lds[wave][(it + 1) % 3]; // slab of an absent tile holds zero records
        Payload *payload = p.task_records ? nullptr : p.tile_payload + (size_t)w0 * kPayloadCap;
        const float4 mat2 = setup_load_mat(p, rec2, lane);
        const uint32_t prev2 = pass1_load_prev(p, rec2, lane);
        const uint4 rec3 = setup_load_rec(p, w3, w3 < ntiles, nrec, lane);
        uint32_t total = 0, n_next = 0;
        for (uint32_t chunk = 0;; chunk++) { // at least one trip: an empty tile still hands over to the next one
            const bool last = (chunk + 1u) * 64u >= n_cur;
            Pass1Item next;
            if (!last) {
                next = pass1_load(p, L, list_cur, n_cur, chunk + 1u, lane);
            } else {
                n_next = pass1_build_list(Ln, list_next, lane);
                next = pass1_load(p, Ln, list_next, n_next, 0u, lane);
            }
            total = pass1_eval<PROJ>(p, L, planes, alpha, lane, cur, total, payload);
            cur = next;
            if (last) break;
        }
        setup_write(p, lds[wave][(it + 2) % 3], rec2, mat2, lane, prev2); // also fences the mask atomics of this tile
        if ((p.task_records != nullptr || total > kPayloadCap) && lane < 2 * (int)kTileRows) // see meshlet_eval
            p.tile_masks[(size_t)w0 * (2 * kTileRows) + lane] = reinterpret_cast<const uint32_t *>(L.draw_mask)[lane];
        if (lane == 0) p.tile_counts[w0] = total;
        if (LIST && p.visible_list == 2u) tile_records_out(p, L, nrec, w0, total, s_red + wave, lane);
        uint16_t *t = list_cur;
        list_cur = list_next;
        list_next = t;
        n_cur = n_next;
        w0 = w1;
        w1 = w2;
        w2 = w3;
        rec2 = rec3;
        it++;
    }
    if (LIST && p.visible_list == 2u) records_finish<kEvWaves>(p, nrec, s_red);
}

template <int PROJ>
__global__ __launch_bounds__(kEvWaves * 64, kEvWavesPerSimd) void meshlet_pass1_kernel(const MeshletCullParams p) {
    meshlet_pass1_body<PROJ>(p);
}
template <int PROJ>
__global__ __launch_bounds__(kEvWaves * 64, kEvWavesPerSimd) void meshlet_pass1_views_kernel(const MeshletCullViews all,
                                                                                            const ViewGroup g) {
    meshlet_pass1_body<PROJ, true>(all.v[g.idx[blockIdx.y]]);
}

// 4-wave blocks, one wave per SIMD each.  The fifth wave per SIMD pays once a wave has enough tiles to stream (8 and
// more); a small launch is mostly its pipeline's ramp, which more, shorter waves only lengthen (a 1/4 and a 1/8
// shard of config 5: 0.095 / 0.085 ms at four waves, 0.1025 / 0.090 at five)
uint32_t eval_grid(uint32_t pass, bool stream, uint32_t max_tiles, uint32_t num_cus) {
    uint32_t wps = (uint32_t)ev_waves_per_simd((int)pass, stream);
    if (wps > (uint32_t)kEvWavesPerSimd && max_tiles < 8u * num_cus * wps * (uint32_t)kEvWaves) wps = (uint32_t)kEvWavesPerSimd;
    uint32_t blocks = num_cus * wps;
    const uint32_t need = (max_tiles + kEvWaves - 1) / kEvWaves;
    if (blocks > need) blocks = need;
    return blocks == 0 ? 1u : blocks;
}

template <int PASS>
void launch_eval_pass(const MeshletCullParams &p, dim3 grid, hipStream_t s) {
    const dim3 block(kEvWaves * 64);
    const bool ortho = p.ci.projection_type == 1;
    if constexpr (PASS == 1) { // list-driven: only last frame's visible meshlets are read
        if (ortho)
            hipLaunchKernelGGL((meshlet_pass1_kernel<1>), grid, block, 0, s, p);
        else
            hipLaunchKernelGGL((meshlet_pass1_kernel<0>), grid, block, 0, s, p);
    } else if (p.ms.sphere != nullptr && p.ms.cls0 != nullptr) { // derived streams + alpha classes: 20.25 B per meshlet
        if (ortho)
            hipLaunchKernelGGL((meshlet_eval_kernel<PASS, 1, false, true, true>), grid, block, 0, s, p);
        else
            hipLaunchKernelGGL((meshlet_eval_kernel<PASS, 0, false, true, true>), grid, block, 0, s, p);
    } else if (p.ms.sphere != nullptr) { // derived streams: 22 B per meshlet, no payload
        if (ortho)
            hipLaunchKernelGGL((meshlet_eval_kernel<PASS, 1, false, true>), grid, block, 0, s, p);
        else
            hipLaunchKernelGGL((meshlet_eval_kernel<PASS, 0, false, true>), grid, block, 0, s, p);
    } else if constexpr (PASS == 0) {
#ifdef ORBIT_TRIAGE
        if (p.debug_flags & 8u) { // phase-stamped kernel (tools/phase.py); | 16: only the per-wave wall-clock stamps
            if (ortho)
                hipLaunchKernelGGL((meshlet_eval_kernel<0, 1, true>), grid, block, 0, s, p);
            else
                hipLaunchKernelGGL((meshlet_eval_kernel<0, 0, true>), grid, block, 0, s, p);
        } else
#endif
        if (ortho) {
            hipLaunchKernelGGL((meshlet_eval_kernel<0, 1, false>), grid, block, 0, s, p);
        } else {
            hipLaunchKernelGGL((meshlet_eval_kernel<0, 0, false>), grid, block, 0, s, p);
        }
    } else {
        if (ortho)
            hipLaunchKernelGGL((meshlet_eval_kernel<PASS, 1, false>), grid, block, 0, s, p);
        else
            hipLaunchKernelGGL((meshlet_eval_kernel<PASS, 0, false>), grid, block, 0, s, p);
    }
}

} // namespace

hipError_t launch_meshlet_eval(const MeshletCullParams &p, uint32_t num_cus, hipStream_t s) {
#if !ORBIT_CONTRACT
    if (p.arith != 0u) return launch_meshlet_eval_contracted(p, num_cus, s); // OrbitCaps.arith_profile
#endif
    const uint32_t max_tiles = (p.dispatch_capacity + kTileRecords - 1) / kTileRecords;
    const uint32_t pass = p.ci.occlusion_pass;
    const dim3 grid(eval_grid(pass, p.ms.sphere != nullptr, max_tiles, num_cus));
#ifndef ORBIT_PASS2_BY_VALUE
    // (a cull that writes the record list, whatever its pass: only the several-views kernel can — meshlet_eval_body LIST)
    if (p.visible_list == 2u) {
        MeshletCullViews one{};
        one.v[0] = p;
        const ViewGroup g{};
        const int src = p.ms.sphere == nullptr ? 0 : (p.ms.cls0 != nullptr ? 2 : 1);
        return launch_meshlet_eval_views(one, g, 1u, pass, p.ci.projection_type == 1, src, max_tiles, num_cus, s);
    }
    // Pass 2 goes through the several-views kernel with one view.  With the block as the kernel's own argument the
    // compiler keeps all of it in scalar registers and, next to the HiZ test's own, runs out of them: 25-34 spilled
    // (v_writelane / v_readlane in a kernel that is bound by its vector pipe); read from the argument segment at a
    // uniform offset as it is needed, none (-Rpass-analysis=kernel-resource-usage; profiles/r04_notes.md).
    if (pass == 2u) {
        MeshletCullViews one{};
        one.v[0] = p;
        ViewGroup g{};
        const int src = p.ms.sphere == nullptr ? 0 : (p.ms.cls0 != nullptr ? 2 : 1);
        const dim3 block(kEvWaves * 64);
        const bool ortho = p.ci.projection_type == 1;
        if (src == 2) {
            if (ortho) hipLaunchKernelGGL((meshlet_eval_views_kernel<2, 1, true, true>), grid, block, 0, s, one, g);
            else hipLaunchKernelGGL((meshlet_eval_views_kernel<2, 0, true, true>), grid, block, 0, s, one, g);
        } else if (src == 1) {
            if (ortho) hipLaunchKernelGGL((meshlet_eval_views_kernel<2, 1, true>), grid, block, 0, s, one, g);
            else hipLaunchKernelGGL((meshlet_eval_views_kernel<2, 0, true>), grid, block, 0, s, one, g);
        } else {
            if (ortho) hipLaunchKernelGGL((meshlet_eval_views_kernel<2, 1, false>), grid, block, 0, s, one, g);
            else hipLaunchKernelGGL((meshlet_eval_views_kernel<2, 0, false>), grid, block, 0, s, one, g);
        }
        return hipGetLastError();
    }
#endif
    switch (pass) {
    case 0: launch_eval_pass<0>(p, grid, s); break;
    case 1: launch_eval_pass<1>(p, grid, s); break;
    default: launch_eval_pass<2>(p, grid, s); break;
    }
    return hipGetLastError();
}

// orbit_cull_shard: the entity stage, the evaluation and the record list of a shard in one launch (pass 0, at most
// kShardMaxChunks chunks of 256 entity-draws; the caller checks both)
hipError_t launch_shard_cull(const FusedCullParams &f, uint32_t num_cus, hipStream_t s) {
#if !ORBIT_CONTRACT
    if (f.m.arith != 0u) return launch_shard_cull_contracted(f, num_cus, s);
#endif
    const MeshletCullParams &p = f.m;
    const uint32_t max_tiles = (p.dispatch_capacity + kTileRecords - 1) / kTileRecords;
    // every chunk needs a workgroup that may take it: at least as many workgroups as chunks
    const dim3 grid(max(eval_grid(0u, p.ms.sphere != nullptr, max_tiles, num_cus), min(f.ne_chunks, num_cus * (uint32_t)kEvWavesPerSimd))), block(kEvWaves * 64);
    ShardCullArgs a{};
    a.v[0] = f;
    const ViewGroup g{};
    const bool ortho = p.ci.projection_type == 1;
    const int src = p.ms.sphere == nullptr ? 0 : (p.ms.cls0 != nullptr ? 2 : 1);
    if (src == 2) {
        if (ortho) hipLaunchKernelGGL((shard_cull_kernel<1, true, true>), grid, block, 0, s, a, g);
        else hipLaunchKernelGGL((shard_cull_kernel<0, true, true>), grid, block, 0, s, a, g);
    } else if (src == 1) {
        if (ortho) hipLaunchKernelGGL((shard_cull_kernel<1, true, false>), grid, block, 0, s, a, g);
        else hipLaunchKernelGGL((shard_cull_kernel<0, true, false>), grid, block, 0, s, a, g);
    } else {
        if (ortho) hipLaunchKernelGGL((shard_cull_kernel<1, false, false>), grid, block, 0, s, a, g);
        else hipLaunchKernelGGL((shard_cull_kernel<0, false, false>), grid, block, 0, s, a, g);
    }
    return hipGetLastError();
}

// One evaluation launch for the m views g.idx[0..m) of `all`, which share the occlusion pass, the projection and the
// meshlet source `src` (0 = Meshlet buffer, 1 = derived streams, 2 = streams with alpha classes).
hipError_t launch_meshlet_eval_views(const MeshletCullViews &all, const ViewGroup &g, uint32_t m, uint32_t pass,
                                     bool ortho, int src, uint32_t max_tiles, uint32_t num_cus, hipStream_t s) {
#if !ORBIT_CONTRACT
    if (all.v[g.idx[0]].arith != 0u) return launch_meshlet_eval_views_contracted(all, g, m, pass, ortho, src, max_tiles, num_cus, s);
#endif
    // the views of a group share the device: each gets an equal part of the resident-wave budget
    uint32_t blocks = max(num_cus * (uint32_t)ev_waves_per_simd((int)pass, src != 0) / m, 1u);
    const uint32_t need = max((max_tiles + kEvWaves - 1) / kEvWaves, 1u);
    if (blocks > need) blocks = need;
    if (m == 1u) blocks = eval_grid(pass, src != 0, max_tiles, num_cus); // one view: the single-view launch's grid
    const dim3 grid(blocks, m), block(kEvWaves * 64);
#define ORBIT_EVAL_VIEWS(PASS)                                                                                        \
    if (src == 2) {                                                                                                   \
        if (ortho) hipLaunchKernelGGL((meshlet_eval_views_kernel<PASS, 1, true, true>), grid, block, 0, s, all, g);   \
        else hipLaunchKernelGGL((meshlet_eval_views_kernel<PASS, 0, true, true>), grid, block, 0, s, all, g);         \
    } else if (src == 1) {                                                                                            \
        if (ortho) hipLaunchKernelGGL((meshlet_eval_views_kernel<PASS, 1, true>), grid, block, 0, s, all, g);         \
        else hipLaunchKernelGGL((meshlet_eval_views_kernel<PASS, 0, true>), grid, block, 0, s, all, g);               \
    } else {                                                                                                          \
        if (ortho) hipLaunchKernelGGL((meshlet_eval_views_kernel<PASS, 1, false>), grid, block, 0, s, all, g);        \
        else hipLaunchKernelGGL((meshlet_eval_views_kernel<PASS, 0, false>), grid, block, 0, s, all, g);              \
    }
    if (pass == 1) {
        if (ortho) hipLaunchKernelGGL((meshlet_pass1_views_kernel<1>), grid, block, 0, s, all, g);
        else hipLaunchKernelGGL((meshlet_pass1_views_kernel<0>), grid, block, 0, s, all, g);
    } else if (pass == 0) {
        ORBIT_EVAL_VIEWS(0)
    } else {
        ORBIT_EVAL_VIEWS(2)
    }
#undef ORBIT_EVAL_VIEWS
    return hipGetLastError();
}

} // namespace orbit
