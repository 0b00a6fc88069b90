// meshlet_emit.hip — the emit launches of meshlet_cull: the survivors of the evaluation (meshlet_eval.hip) as
// MeshletDrawCommands at their canonical positions (shaders/meshlet_cull.comp:216-230 of the reference repo; the
// append order of its atomics replaced by the prefix sum over the tiles' survivor counts), or as the sharded
// engine's 8-B visible list.
//
//   meshlet_emit_kernel<false>   commands from the survivor payload the Meshlet-buffer evaluation kept
//   meshlet_emit_chain_kernel    commands from the derived streams: counts streamed, offsets derived along the
//                                upload chain (kernels.h MeshletStreamView)
//   meshlet_emit_kernel<true, *> the 8-B visible list {meshlet_index, entity_index}
#include "meshlet_common.h"
#include "handoff.h"

namespace orbit {

namespace {

// ---------------------------------------------------------------------------
// The prefix sum over the tiles' survivor counts, inside the CHAIN emit launch (the other emit launches keep the scan
// launch in front of them: scan.hip).  Rounds 1-3 ran its first level as a launch of its own between evaluation and
// emit: 4.8 us + a launch boundary at BASELINE config 5 for 196 KB of counts.  Now the first workgroups of the chain
// emit scan one chunk of kScanChunk tiles each (chunk-local exclusive prefixes to
// tile_base, stored write-through), publish the chunk's sum as sum << 1 | 1 in chunk_sums (zero since the
// evaluation launch of this cull cleared it), and every workgroup waits for the sums of all chunks — ONE hop, under
// which the launch's own ramp (records, ballots, links: nothing of which depends on the bases) can run — and builds the
// chunk bases in LDS as before.  A workgroup whose wait outlasts its patience does the missing scan itself (below).
// tile_base is then read with sc1 loads (another workgroup of this launch stored it).
// Returns the grand total; s_chunk_base[c] = exclusive prefix of chunk c for c < kEmitChunkTable.
__device__ __forceinline__ void emit_scan_chunk(const MeshletCullParams &p, uint32_t c, uint32_t ntiles,
                                                __amdgpu_buffer_rsrc_t CS, uint32_t *s_scan) {
    static_assert(kScanChunk == 4 * 256, "a thread of the emit block scans four tiles of a chunk");
    const uint32_t i = c * kScanChunk + threadIdx.x * 4u;
    // range-checked: quads past the one that holds the last tile read zeros (that quad lies inside the 256-B padding)
    const auto q = __builtin_amdgcn_raw_buffer_load_b128(rsrc_of(p.tile_counts, ((ntiles + 3u) & ~3u) * 4u), i * 4u, 0, 0);
    uint4 v = make_uint4(q[0], q[1], q[2], q[3]);
    v.x = i < ntiles ? v.x : 0u;
    v.y = i + 1u < ntiles ? v.y : 0u;
    v.z = i + 2u < ntiles ? v.z : 0u;
    v.w = i + 3u < ntiles ? v.w : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<4>(v.x + v.y + v.z + v.w, s_scan, &total);
    const v4u o = {ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z};
    // whole quads up to the one that holds the last tile (the scratch is padded to 256 B: that quad lies inside)
    const __amdgpu_buffer_rsrc_t TB = rsrc_of(p.tile_base + c * kScanChunk, min(kScanChunk, (ntiles - c * kScanChunk + 3u) & ~3u) * 4u);
    __builtin_amdgcn_raw_buffer_store_b128(o, TB, threadIdx.x * 16u, 0, kSc1);
    drain_stores();
    __syncthreads(); // every wave's prefixes are in memory
    st_sc1(CS, threadIdx.x == 0 ? c * 4u : kHandoffNoOffset, total << 1 | 1u);
}

__device__ __forceinline__ void emit_scan_duty(const MeshletCullParams &p, uint32_t ntiles, uint32_t nchunks, uint32_t *s_scan) {
    if (blockIdx.x == 0 && threadIdx.x < kTicketPools) p.tickets[threadIdx.x * kTicketStride] = 0u; // next evaluation
    const __amdgpu_buffer_rsrc_t CS = rsrc_of(p.chunk_sums, nchunks * 4u);
    for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) emit_scan_chunk(p, c, ntiles, CS, s_scan); // block-uniform
}

// Nothing here waits for another workgroup for good: a workgroup whose patience with a chunk's sum runs out scans that
// chunk itself (the same words, the same values: harmless twice).  The first workgroups of a launch are dispatched
// first and the wait is normally a few polls; but two launches of this kind sharing the device — two processes on one
// GPU — can each hold the slots the other's first workgroups are waiting for (found with bench.py --gpus 2 on one
// device: both ran into the poll bound), and no assumption about dispatch order or residency is worth a hang.
// (patience: p.scan_patience polls, 256 by default — each a memory round trip, some hundred microseconds in all)

__device__ __forceinline__ uint32_t emit_scan_wait(const MeshletCullParams &p, uint32_t ntiles, uint32_t nchunks,
                                                   uint32_t *s_chunk_base, uint32_t *s_scan) {
    const __amdgpu_buffer_rsrc_t CS = rsrc_of(p.chunk_sums, nchunks * 4u);
    bool missing = false;
    for (uint32_t c = threadIdx.x; c < nchunks; c += 256) {
        uint32_t v = ld_sc1(CS, c * 4u);
        for (uint32_t spins = 0; v == 0u && spins < p.scan_patience; spins++) {
            poll_pause();
            v = ld_sc1(CS, c * 4u);
        }
        missing = missing || v == 0u;
    }
    const uint64_t wave_missing = __ballot(missing);
    if ((threadIdx.x & 63) == 0) s_scan[threadIdx.x >> 6] = wave_missing != 0ull ? 1u : 0u;
    __syncthreads();
    const uint32_t any_missing = uniform(s_scan[0] | s_scan[1] | s_scan[2] | s_scan[3]);
    __syncthreads();
    if (any_missing != 0u) {
        for (uint32_t c = 0; c < nchunks; c++) {
            if (threadIdx.x == 0) s_chunk_base[0] = ld_sc1(CS, c * 4u);
            __syncthreads();
            const uint32_t have = uniform(s_chunk_base[0]);
            __syncthreads();
            if (have == 0u) emit_scan_chunk(p, c, ntiles, CS, s_scan);
        }
        drain_stores(); // the sums stored here are in memory before anyone of this workgroup reads them back below
        __syncthreads();
    }
    uint32_t grand_total = 0;
    for (uint32_t c0 = 0; c0 < nchunks; c0 += 256) { // block-uniform trip count
        const uint32_t c = c0 + threadIdx.x;
        const uint32_t v = ld_sc1(CS, c < nchunks ? c * 4u : kHandoffNoOffset); // (every sum is in: seen above, or stored here)
        if (c < nchunks && v == 0u) latch_status(p.status, ORBIT_E_TIMEOUT); // (cannot happen: reported rather than summed as zero)
        uint32_t sub;
        const uint32_t ex = block_exclusive_scan<4>(v >> 1, s_scan, &sub);
        if (c < nchunks && c < kEmitChunkTable) s_chunk_base[c] = grand_total + ex;
        grand_total += sub;
    }
    __syncthreads();
    return grand_total;
}

// One wave per tile (grid stride): commands of the tile's survivors at the
// scanned offset (:216-230).  Everything a tile needs is loaded ahead of its
// turn, so no memory round trip is exposed between tiles: the 16 records, count
// and chunk-local base two tiles ahead; the survivor payload (whose extent
// needs the count) one tile ahead.
struct EmitTile {
    uint4 rec; // lanes 0..15: the tile's dispatch records
    uint32_t count, base;
    uint32_t mword; // STREAM: lanes 0..15: the tile's should-draw ballots (every tile is rebuilt from them)
};

template <bool STREAM>
__device__ __forceinline__ EmitTile emit_prefetch(const MeshletCullParams &p, uint32_t tile, uint32_t ntiles,
                                                  uint32_t nrec, int lane) {
    EmitTile e;
    const uint4 *recs = reinterpret_cast<const uint4 *>(p.dispatch_buffer + ORBIT_DISPATCH_HEADER);
    const bool tv = tile < ntiles;
    const uint32_t t = tv ? tile : 0u;
    const uint32_t ri = t * kTileRecords + (lane & 15);
    e.rec = *((tv && ri < nrec) ? recs + ri : reinterpret_cast<const uint4 *>(p.zero_page));
    e.count = tv ? p.tile_counts[t] : 0u;
    e.base = p.tile_base[t]; // exclusive prefix inside the tile's scan chunk
    e.mword = 0u;
    if constexpr (STREAM)
        e.mword = *(tv ? p.tile_masks + ((size_t)t * (2 * kTileRows) + (lane & 15))
                       : reinterpret_cast<const uint32_t *>(p.zero_page));
    return e;
}

// Payload entries [first, first + 64) of a tile; lanes past the count read the zero page.
__device__ __forceinline__ Payload emit_load_payload(const MeshletCullParams &p, uint32_t tile, uint32_t count,
                                                   uint32_t first, int lane) {
    const uint32_t j = first + (uint32_t)lane;
    const bool has = j < count && count <= kPayloadCap;
    return *(has ? p.tile_payload + ((size_t)tile * kPayloadCap + j) : reinterpret_cast<const Payload *>(p.zero_page));
}


// Lane = survivor.  A tile with at most kPayloadCap survivors (the common case:
// ~50 at 10 % survival) is one contiguous payload run written by meshlet_eval in
// command order; the lane turns its entry into the 7 command words, and the run
// of 7-dword structs goes out through LDS so that each of the 7 stores covers
// 256 contiguous bytes instead of 64 addresses 28 B apart.  A denser tile is
// rebuilt from its ballots: survivors compacted into an LDS list of (row, lane)
// codes (ballot rank = list position, i.e. canonical order), Meshlet bytes
// 16..31 gathered.  The second scan level (chunk sums -> chunk bases, at most a
// few dozen values) is redone by every block in LDS instead of being a launch.
// (Draw commands of an evaluation that read the derived streams — no payload — come from meshlet_emit_chain_body below.)
__device__ __forceinline__ void meshlet_emit_body(const MeshletCullParams &p) {
    constexpr bool STREAM = false; // this body serves the evaluations that kept a payload
    __shared__ uint16_t s_list[4][kTileRows * 64];
    __shared__ uint32_t s_cmd[4][64 * 7];
    __shared__ uint32_t s_chunk_base[kEmitChunkTable];
    __shared__ uint32_t s_scan[256 / 64 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t *list = s_list[wave];
    uint32_t *cmd = s_cmd[wave];
    const uint32_t nrec = min(*reinterpret_cast<const uint32_t *>(p.dispatch_buffer), p.dispatch_capacity);
    const uint32_t ntiles = (nrec + kTileRecords - 1) / kTileRecords;
    const uint32_t nchunks = (ntiles + kScanChunk - 1) / kScanChunk;
    uint32_t grand_total = 0;
    for (uint32_t c0 = 0; c0 < nchunks; c0 += 256) { // block-uniform trip count
        const uint32_t c = c0 + threadIdx.x;
        const uint32_t v = c < nchunks ? p.chunk_sums[c] : 0u;
        uint32_t sub;
        const uint32_t ex = block_exclusive_scan<4>(v, s_scan, &sub);
        if (c < nchunks && c < kEmitChunkTable) s_chunk_base[c] = grand_total + ex;
        grand_total += sub;
    }
    __syncthreads();
    uint32_t *out = reinterpret_cast<uint32_t *>(p.draw_buffer + ORBIT_DRAW_HEADER);
    const uint32_t stride = gridDim.x * 4;
    uint32_t tile = blockIdx.x * 4 + wave;
    EmitTile cur = emit_prefetch<STREAM>(p, tile, ntiles, nrec, lane);
    EmitTile nxt = emit_prefetch<STREAM>(p, tile + stride, ntiles, nrec, lane);
    Payload cur_pay = Payload{0u, 0u, 0u};
    if constexpr (!STREAM) cur_pay = emit_load_payload(p, tile, cur.count, 0u, lane);
    for (; tile < ntiles; tile += stride) {
        const EmitTile nn = emit_prefetch<STREAM>(p, tile + 2 * stride, ntiles, nrec, lane);
        Payload nxt_pay = Payload{0u, 0u, 0u};
        if constexpr (!STREAM) nxt_pay = emit_load_payload(p, tile + stride, nxt.count, 0u, lane);
        const uint32_t n = cur.count;
        if (n != 0u) { // wave-uniform
            const uint32_t base = cur.base + s_chunk_base[tile / kScanChunk];
            const bool dense = STREAM || n > kPayloadCap; // wave-uniform
            if (dense) {
                const uint32_t mword = STREAM ? cur.mword : p.tile_masks[(size_t)tile * (2 * kTileRows) + (lane & 15)];
                uint32_t row_base = 0;
#pragma unroll
                for (int r = 0; r < (int)kTileRows; r++) {
                    const uint32_t lo = (uint32_t)__shfl((int)mword, 2 * r, 64);
                    const uint32_t hi = (uint32_t)__shfl((int)mword, 2 * r + 1, 64);
                    const uint64_t mask = ((uint64_t)hi << 32) | lo;
                    if ((mask >> lane) & 1ull) list[row_base + lane_prefix(mask)] = (uint16_t)(r * 64 + lane);
                    row_base += (uint32_t)__popcll(mask);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            for (uint32_t j0 = 0; j0 < n; j0 += 64) { // one trip unless more than 64 survivors
                const uint32_t j = j0 + lane;
                Payload pay; // payload layout (make_payload)
                if (dense) {
                    const uint32_t code = j < n ? list[j] : 0u;
                    const uint32_t rid = (code >> 6) * 2u + ((code >> 5) & 1u); // record 2r + half
                    const uint32_t mi = (uint32_t)__shfl((int)cur.rec.y, (int)rid, 64) + (code & 31u);
                    const uint4 b = *(j < n ? reinterpret_cast<const uint4 *>(p.meshlets + mi) + 1
                                            : reinterpret_cast<const uint4 *>(p.zero_page));
                    pay = make_payload(code & 31u, rid, b);
                } else {
                    pay = j0 == 0u ? cur_pay : emit_load_payload(p, tile, n, j0, lane);
                }
                const uint32_t pay_rid = pay.w & 15u;
                const uint32_t ent = (uint32_t)__shfl((int)cur.rec.x, (int)pay_rid, 64);
                const uint32_t pay_meshlet = (uint32_t)__shfl((int)cur.rec.y, (int)pay_rid, 64) + ((pay.w >> 4) & 31u);
                const uint32_t vertex_count = (pay.w >> 16) & 0xFFu;
                const uint32_t triangle_count = pay.w >> 24;
                uint32_t *c = cmd + lane * 7;
                c[0] = triangle_count * 3u;         // cmd_index_count
                c[1] = 1u;                          // cmd_instance_count
                c[2] = (pay.z + vertex_count) * 4u; // cmd_first_index
                c[3] = pay.z;                       // cmd_vertex_offset = int(data_offset)
                c[4] = ent;                         // cmd_first_instance = entity_index
                c[5] = pay.y;                       // meshlet_vertex_offset
                c[6] = pay_meshlet;                 // meshlet_index
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t first = base + j0; // slot of the trip's first command
                const uint32_t room = first < p.draw_capacity ? p.draw_capacity - first : 0u;
                const uint32_t words = min(min(n - j0, 64u), room) * 7u;
                uint32_t *dst = out + (size_t)first * 7u;
#pragma unroll
                for (uint32_t k = 0; k < 7; k++) {
                    const uint32_t w = k * 64u + (uint32_t)lane;
                    if (w < words) dst[w] = cmd[w];
                }
                __builtin_amdgcn_wave_barrier(); // cmd is rewritten by the next trip
            }
            __builtin_amdgcn_wave_barrier(); // list is reused by the next tile
        }
        cur = nxt;
        cur_pay = nxt_pay;
        nxt = nn;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // `count` header: subsumes the fill_buffer(count = 0) of draw_gen.rs:411-417
        *p.total = grand_total;
        *reinterpret_cast<uint32_t *>(p.draw_buffer) = min(grand_total, p.draw_capacity);
        if (grand_total > p.draw_capacity) latch_status(p.status, ORBIT_E_CAPACITY);
    }
}

// ---------------------------------------------------------------------------
// Draw commands from the derived streams.  meshlet_eval kept only ballots; a survivor's command needs its meshlet's
// vertex_offset, data_offset and the two counts (:216-230).  Gathering them per survivor (12 B out of a 128-B line
// each) moved 3x the bytes the commands hold; this body streams 2 B per meshlet of the records that have survivors
// instead and derives the rest, because the uploader lays a mesh's meshlets out as a chain (kernels.h
// MeshletStreamView): data_offset[i + 1] = data_offset[i] + vertex_count[i] + ceil(3 * triangle_count[i] / 4), one
// vertex_offset per submesh.
// Round 3 found the launch bound by vector-instruction issue (profiles/r03_notes.md: since round 4 it moves its bytes at
// the rate the chip streams at, profiles/r04_notes.md §6), so a wave works on a GROUP of two
// consecutive wave tiles — 32 dispatch records — at a time: everything that is per record runs on 32 lanes instead of
// 16, and everything that is per tile is paid once per two.
//   phase A  lane = two neighbouring meshlets of a record (a 16-lane DPP row = one record, four records per
//            instruction): data sizes from the counts (one 4-B load per lane), prefix sum inside each record (four
//            DPP steps, two quad-rows per scan as 16-bit halves), {prefix, counts} of every meshlet and the records'
//            totals to LDS; the survivors' (record, lane) codes peeled off the ballots in canonical order.
//   phase B  lane = record: a record's base is the gathered {vertex_offset, data_offset} of its first meshlet where a
//            chain starts (first record of the group with survivors, a jump in meshlet_offset, a predecessor without
//            survivors, a broken link since the predecessor's second meshlet: all known a group ahead, so the gather
//            is prefetched), else the predecessor's base + total — a segmented prefix sum over the 32 records.
//   phase C  lane = survivor: offsets = record base + prefix, valid while the record's link bits up to the lane are
//            all set; the few survivors behind a broken link gather their own 8 B.  Commands go out through LDS as
//            contiguous words (as in meshlet_emit_body).
// Pipeline per wave: records and ballots three groups ahead; link words two; counts and chain-start bases one.
// All loads are range-checked buffer loads (an offset outside the resource reads zero): no pointer selects, no
// branches around loads, 32-bit address arithmetic, and the stream's coverage check comes for free.
constexpr uint32_t kGroupTiles = 2, kGroupRecords = kGroupTiles * kTileRecords; // 32 records = 1024 meshlets
constexpr uint32_t kSlowWords = 16; // 512 iterations per wave: more than any grid the library launches leaves a wave
constexpr uint32_t kDynGroups = 16; // ticketed groups a wave takes inside its pipelined loop (power of two)

struct __attribute__((aligned(16))) ChainLds {
    uint4 rec[kGroupRecords];            // phase B -> C: {data_offset base, vertex_offset, entity_index, meshlet_offset}
    uint2 rec2[kGroupRecords];           // ... behind the record's first broken link: {data_offset base, vertex_offset}
    uint32_t rs[kGroupRecords];          // chain_stage: per record the byte offset of its first count (or "none")
    uint32_t row[kGroupRecords * 32];    // phase A -> C, per (record, lane): size prefix inside the record | counts << 16
    uint32_t link[kGroupRecords];        // link bits of the record's 32 lanes
    uint32_t tot[kGroupRecords / 2];     // [quad-row pair * 4 + sub]: data words of records 4 q + sub / 4 (q + 4) + sub (16:16)
    uint32_t slow[kSlowWords];           // bit i: the wave's i-th group needs the general form (process<!FAST>)
    uint32_t dyn[kDynGroups];            // the wave's ticketed groups (its iterations n_static ..)
    uint32_t cmd[64 * 7];
    uint16_t list[kGroupRecords * 32];
};


struct ChainRsrc {
    __amdgpu_buffer_rsrc_t recs, counts, bases, masks; // the dispatch records, the scan scratch
    __amdgpu_buffer_rsrc_t cnt, link, cmd, base32;     // the stream (based at its first meshlet)
    uint32_t first, count, first_word;
};

struct ChainGroup {
    uint32_t ent, y, z; // lane = record of the group (lanes 32.. repeat lanes 0..31): entity, meshlet offset, count
    uint32_t mword;     // the record's should-draw ballot
    uint32_t count;     // lane & 1: survivors of the group's tile (lane & 1)
    uint32_t base;      // chunk-local output slot of the group's first survivor
};

__device__ __forceinline__ ChainGroup chain_prefetch(const ChainRsrc &B, uint32_t group, int lane, bool with_base = true) {
    ChainGroup e;
    const uint32_t g = group < 0x007FFFFFu ? group : 0x007FFFFFu; // keeps g * 512 out of range instead of wrapping
    const uint32_t k = (uint32_t)lane & 31u;
    const auto r = __builtin_amdgcn_raw_buffer_load_b96(B.recs, (g * kGroupRecords + k) * 16u, 0, 0);
    e.ent = r[0], e.y = r[1], e.z = r[2];
    e.mword = __builtin_amdgcn_raw_buffer_load_b32(B.masks, (g * kGroupRecords + k) * 4u, 0, 0);
    e.count = __builtin_amdgcn_raw_buffer_load_b32(B.counts, (g * kGroupTiles + ((uint32_t)lane & 1u)) * 4u, 0, 0);
    // (stored by this launch: sc1; the launch's first prefetches run before the scan is known to be complete and fetch
    // their bases behind the wait)
    e.base = __builtin_amdgcn_raw_buffer_load_b32(B.bases, with_base ? g * kGroupTiles * 4u : kNoOffset, 0, kSc1);
    return e;
}

struct LinkStage {
    uint32_t l0, l1; // lane = record: the two link words that cover its lanes
};

struct ChainStage {
    uint32_t c2[kGroupRecords / 4]; // per quad-row: the counts of this lane's two meshlets (16:16; 0: not loaded)
    uint32_t link32;                // lane = record: bit l = link bit of meshlet (meshlet_offset + l)
    uint32_t bv, bd;                // lane = record: vertex_offset / data_offset of its first meshlet (chain starts only)
    uint32_t bv2, bd2;              // lane = record: those of the meshlet behind the record's first broken link, if any
    uint32_t flags;                 // lane = record: 1 = has survivors (counts loaded), 2 = chain start (bv, bd loaded)
};

// lane = record: does the emit read this record at all?  Records without survivors are skipped; a record that begins
// outside the stream reads zeros (ORBIT_E_RANGE is the evaluation's to latch).
__device__ __forceinline__ bool chain_has(const ChainRsrc &B, const ChainGroup &e) {
    return e.mword != 0u && e.z != 0u && e.y - B.first < B.count;
}

__device__ __forceinline__ LinkStage link_stage(const ChainRsrc &B, const ChainGroup &e) {
    const uint32_t wo = chain_has(B, e) ? ((e.y >> 5) - B.first_word) * 4u : kNoOffset;
    LinkStage k;
    k.l0 = __builtin_amdgcn_raw_buffer_load_b32(B.link, wo, 0, 0);
    k.l1 = __builtin_amdgcn_raw_buffer_load_b32(B.link, wo, 0 + 4, 0);
    return k;
}

// First lane of a record (1..31) whose link bit is clear, 32 if the links of lanes 1..31 all hold.
__device__ __forceinline__ uint32_t chain_first_break(uint32_t link32) {
    const uint32_t open = ~link32 & 0xFFFFFFFEu;
    return open != 0u ? (uint32_t)__builtin_ctz(open) : 32u;
}

// Lane i <- lane i - 1 of the wave (wave_shr:1); lane 0 gets `first`.
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t first) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xF, 0xF, false);
}

__device__ __forceinline__ ChainStage chain_stage(const ChainRsrc &B, ChainLds &S, const ChainGroup &e,
                                                  const LinkStage &k, int lane) {
    ChainStage s;
    const bool has = chain_has(B, e);
    if (lane < (int)kGroupRecords) S.rs[lane] = has ? (e.y - B.first) * 2u : kNoOffset;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // quad-row q: the wave's four 16-lane rows are records 4 q .. 4 q + 3, lane i of a row holds meshlets 2 i, 2 i + 1.
    // Lanes past a record's count read its neighbours' counts (or zeros past the stream): they feed nothing that is
    // used — a prefix of lanes behind them, and the total of a record that, being short, starts no continuation.
    const uint32_t sub = (uint32_t)lane >> 4, i4 = ((uint32_t)lane & 15u) * 4u;
    uint32_t off[kGroupRecords / 4];
#pragma unroll
    for (int q = 0; q < (int)kGroupRecords / 4; q++) off[q] = S.rs[4 * q + sub];
#pragma unroll
    for (int q = 0; q < (int)kGroupRecords / 4; q++)
        s.c2[q] = __builtin_amdgcn_raw_buffer_load_b32(B.cnt, off[q] + i4, 0, ORBIT_EMIT_CNT_AUX);
    // lane = record.  A record continues its predecessor's chain when that one was read too, ends where this one
    // begins, and every link from the predecessor's second meshlet to this record's first one holds; otherwise it
    // starts a chain from the gathered offsets of its first meshlet.
    s.link32 = __funnelshift_r(k.l0, k.l1, e.y & 31u);
    const bool full = has && e.z == 32u && (s.link32 | 1u) == 0xFFFFFFFFu; // the chain reaches the record's end
    const uint32_t prev_y = wave_shr1(e.y, 0u), prev_full = wave_shr1(full ? 1u : 0u, 0u);
    const bool first = ((uint32_t)lane & 31u) == 0u; // (lane 32 repeats record 0)
    const bool cont = has && !first && prev_full != 0u && e.y == prev_y + 32u && (s.link32 & 1u) != 0u;
    const bool start = has && !cont;
    s.flags = (has ? 1u : 0u) | (start ? 2u : 0u);
    // a chain's base: {vertex_offset, data_offset} of the record's first meshlet — from the dense base32 array where the
    // record starts on a multiple of 32 (8 B per record, neighbours in one line), gathered out of `cmd` (a 128-B line for
    // 8 B) where it does not.  Two loads from each array, the unused pair at an offset outside the resource: the
    // vector-memory code of the loop stays straight-line.
    const bool aligned = (e.y & 31u) == 0u;
    const uint32_t co = (start && !aligned) ? (e.y - B.first) * 12u : kNoOffset;
    const uint32_t bo = (start && aligned) ? ((e.y >> 5) - B.first_word) * 8u : kNoOffset;
    const uint32_t gv = __builtin_amdgcn_raw_buffer_load_b32(B.cmd, co, 0, 0);
    const uint32_t gd = __builtin_amdgcn_raw_buffer_load_b32(B.cmd, co, 0 + 4, 0);
    const uint32_t dv = __builtin_amdgcn_raw_buffer_load_b32(B.base32, bo, 0, 0);
    const uint32_t dd = __builtin_amdgcn_raw_buffer_load_b32(B.base32, bo, 0 + 4, 0);
    s.bv = gv | dv; // (the pair that was not asked for reads zeros)
    s.bd = gd | dd;
    // A submesh or allocation boundary INSIDE the record restarts the chain there: the offsets of the meshlet behind the
    // first broken link are gathered as well (none: nothing is read).  Survivors behind a second one gather their own.
    const uint32_t brk = chain_first_break(s.link32);
    const uint32_t co2 = (has && brk < 32u) ? (e.y + brk - B.first) * 12u : kNoOffset;
    s.bv2 = __builtin_amdgcn_raw_buffer_load_b32(B.cmd, co2, 0, 0);
    s.bd2 = __builtin_amdgcn_raw_buffer_load_b32(B.cmd, co2, 0 + 4, 0);
    return s;
}

// data sizes of two meshlets, packed 16:16, from their packed counts (meshlet_data_words, kernels.h)
__device__ __forceinline__ uint32_t chain_size2(uint32_t c2) {
    const uint32_t vc = c2 & 0x00FF00FFu, tc = (c2 >> 8) & 0x00FF00FFu;
    return vc + (((tc * 3u + 0x00030003u) >> 2) & 0x3FFF3FFFu);
}

// Inclusive scan (sum) inside each row of 16 lanes: row_shr 1 / 2 / 4 / 8.
__device__ __forceinline__ uint32_t row_inclusive_scan(uint32_t v) {
#define ORBIT_ROW_STEP(ctrl) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xF, 0xF, false)
    ORBIT_ROW_STEP(0x111);
    ORBIT_ROW_STEP(0x112);
    ORBIT_ROW_STEP(0x114);
    ORBIT_ROW_STEP(0x118);
#undef ORBIT_ROW_STEP
    return v;
}

__device__ __forceinline__ void meshlet_emit_chain_body(const MeshletCullParams &p) {
    __shared__ ChainLds s_chain[4];
    __shared__ uint32_t s_chunk_base[kEmitChunkTable];
    __shared__ uint32_t s_scan[256 / 64 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ChainLds &S = s_chain[wave];
    const uint32_t nrec = min(*reinterpret_cast<const uint32_t *>(p.dispatch_buffer), p.dispatch_capacity);
    const uint32_t ntiles = (nrec + kTileRecords - 1) / kTileRecords;
    const uint32_t ngroups = (ntiles + kGroupTiles - 1) / kGroupTiles;
    const uint32_t nchunks = (ntiles + kScanChunk - 1) / kScanChunk;
    static_assert(kScanChunk % kGroupTiles == 0, "a group's tiles share a scan chunk: its survivors are contiguous");
    emit_scan_duty(p, ntiles, nchunks, s_scan);
    ChainRsrc B;
    B.recs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.dispatch_buffer) + ORBIT_DISPATCH_HEADER, 0,
                                               nrec * 16u, kBufFlags);
    B.masks = __builtin_amdgcn_make_buffer_rsrc(p.tile_masks, 0, ntiles * 64u, kBufFlags);
    B.counts = __builtin_amdgcn_make_buffer_rsrc(p.tile_counts, 0, ntiles * 4u, kBufFlags);
    B.bases = __builtin_amdgcn_make_buffer_rsrc(p.tile_base, 0, ntiles * 4u, kBufFlags);
    B.first = p.ms.first;
    B.count = p.ms.count;
    B.first_word = p.ms.first >> 5;
    // + 2 bytes: the 4-B load of a lane whose first meshlet is the stream's last one (the arrays are padded)
    B.cnt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.ms.cnt + p.ms.first), 0, p.ms.count * 2u + 2u, kBufFlags);
    B.link = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.ms.link + B.first_word), 0,
                                               (((p.ms.first + p.ms.count + 31u) >> 5) - B.first_word + 2u) * 4u, kBufFlags);
    B.cmd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.ms.cmd + (size_t)p.ms.first * 3u), 0,
                                              p.ms.count * 12u, kBufFlags);
    B.base32 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint2 *>(p.ms.base32 + B.first_word), 0,
                                                 (((p.ms.first + p.ms.count + 31u) >> 5) - B.first_word) * 8u, kBufFlags);
    uint32_t *out = reinterpret_cast<uint32_t *>(p.draw_buffer + ORBIT_DRAW_HEADER);
    const uint32_t stride = gridDim.x * 4;
    const uint32_t sub = (uint32_t)lane >> 4, i16 = (uint32_t)lane & 15u;
    // One group's phases, from its prefetched records (`cur`) and staged counts / bases (`cs`).
    // FAST (the pipelined loop): straight-line vector-memory code.  The wave's vector-memory operations retire in
    // issue order, so waiting for a prefetched load also waits for every store issued before it — unless the compiler
    // knows exactly how many operations lie in between.  Hence, in that loop, no load or store in any branch: the same
    // fourteen stores per group whatever its survivors (a group without any stores into an empty window).  A group
    // that needs more — over 128 survivors, or a survivor behind a SECOND boundary inside one record — stores nothing
    // there, returns true, and is done again by the general form (!FAST) after the loop.
    auto process = [&](auto fast_tag, const ChainGroup &cur, const ChainStage &cs, uint32_t group) -> bool {
        constexpr bool FAST = decltype(fast_tag)::value;
        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)cur.count, 0) +
                           (uint32_t)__builtin_amdgcn_readlane((int)cur.count, 1);
        // wave-uniform, and provably so for the compiler (the store window's descriptor lives in SGPRs)
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)(cur.base + s_chunk_base[min(group * kGroupTiles / kScanChunk, kEmitChunkTable - 1u)]));
        // ---- phase A: quad-rows q and q + 4 share a scan (16-bit halves: a record's sizes sum to < 2^14)
#pragma unroll
        for (int q = 0; q < (int)kGroupRecords / 8; q++) {
            const uint32_t ca = cs.c2[q], cb = cs.c2[q + 4];
            const uint32_t sa = chain_size2(ca), sb = chain_size2(cb);         // {first, second} meshlet of the lane
            const uint32_t pa = (sa & 0xFFFFu) + (sa >> 16), pb = (sb & 0xFFFFu) + (sb >> 16);
            const uint32_t pair = pa | (pb << 16);
            const uint32_t incl = row_inclusive_scan(pair);
            const uint32_t excl = incl - pair;
            const uint32_t ea = excl & 0xFFFFu, eb = excl >> 16;
            // per meshlet: prefix | counts << 16 (the second meshlet's prefix includes the first one's size)
            uint2 wa, wb;
            wa.x = ea | (ca << 16);
            wa.y = ((ea + (sa & 0xFFFFu)) & 0xFFFFu) | (ca & 0xFFFF0000u);
            wb.x = eb | (cb << 16);
            wb.y = ((eb + (sb & 0xFFFFu)) & 0xFFFFu) | (cb & 0xFFFF0000u);
            *reinterpret_cast<uint2 *>(&S.row[q * 128 + lane * 2]) = wa;       // record 4 q + sub, lanes 2 i, 2 i + 1
            *reinterpret_cast<uint2 *>(&S.row[(q + 4) * 128 + lane * 2]) = wb;
            if (i16 == 15u) S.tot[q * 4 + sub] = incl;
        }
        // the survivors' codes (record * 32 + lane) in canonical order: lane i owns half a record's ballot and
        // peels its bits off one per trip
        {
            const uint32_t w = (uint32_t)__shfl((int)cur.mword, lane >> 1, 64);
            uint32_t bits = (w >> (16u * ((uint32_t)lane & 1u))) & 0xFFFFu;
            const uint32_t pc = (uint32_t)__popc(bits);
            uint32_t pos = wave_inclusive_scan(pc) - pc;
            const uint32_t code0 = (uint32_t)lane * 16u;
            while (__ballot(bits != 0u) != 0ull) {
                if (bits != 0u) {
                    S.list[pos] = (uint16_t)(code0 + (uint32_t)__builtin_ctz(bits));
                    pos++;
                    bits &= bits - 1u;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- phase B: lane = record (32 of them; lanes 32.. repeat)
        {
            const uint32_t k = (uint32_t)lane & 31u, kq = k >> 2;
            const uint32_t tot = (S.tot[(kq & 3u) * 4u + (k & 3u)] >> (16u * (kq >> 2))) & 0xFFFFu;
            const uint32_t prev_tot = wave_shr1(tot, 0u);
            const uint32_t brk = chain_first_break(cs.link32);
            const uint32_t brk_prefix = S.row[k * 32u + (brk & 31u)] & 0xFFFFu; // data words in front of that meshlet
            const bool cont = (cs.flags & 3u) == 1u; // has survivors and continues the predecessor's chain
            uint32_t f = cont ? 0u : 1u;             // segment head
            uint32_t a_d = f ? cs.bd : prev_tot, a_v = cs.bv; // bv / bd are zero unless the record starts a chain
#define ORBIT_SEG_STEP(ctrl, rows)                                                                              \
    {                                                                                                           \
        const uint32_t sd = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_d, ctrl, rows, 0xF, false);         \
        const uint32_t sv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_v, ctrl, rows, 0xF, false);         \
        const uint32_t sf = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, ctrl, rows, 0xF, false);           \
        a_d += f ? 0u : sd;                                                                                     \
        a_v += f ? 0u : sv;                                                                                     \
        f |= sf;                                                                                                \
    }
            // (a lane without a source in a step keeps what it has: record 16 has no predecessor in its row of 16
            // lanes and is still no head — its segment arrives with the row_bcast15 step)
            ORBIT_SEG_STEP(0x111, 0xF)
            ORBIT_SEG_STEP(0x112, 0xF)
            ORBIT_SEG_STEP(0x114, 0xF)
            ORBIT_SEG_STEP(0x118, 0xF)
            ORBIT_SEG_STEP(0x142, 0xA) // row_bcast15: records 16..31 continue from record 15 (lanes 48.. from 47)
#undef ORBIT_SEG_STEP
            if (lane < (int)kGroupRecords) {
                S.rec[lane] = make_uint4(a_d, a_v, cur.ent, cur.y);
                S.rec2[lane] = make_uint2(cs.bd2 - brk_prefix, cs.bv2);
                S.link[lane] = cs.link32;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- phase C: lane = survivor; one trip = up to 64 commands
        // is survivor j behind a second broken link of its record?  (then only its own gathered words are right)
        auto second_break = [&](uint32_t j) -> bool {
            const uint32_t code = j < n ? S.list[j] : 0u;
            const uint32_t lk = S.link[code >> 5], l = code & 31u;
            const uint32_t brk = chain_first_break(lk);
            const uint32_t need = ((2u << l) - 1u) & ~((2u << (brk & 31u)) - 1u); // links of lanes brk + 1 .. l
            return j < n && l >= brk && (lk & need) != need;
        };
        bool slow = false;
        if constexpr (FAST)
            slow = n > 128u || __ballot(second_break((uint32_t)lane) || second_break(64u + (uint32_t)lane)) != 0ull;
        auto trip = [&](uint32_t j0) {
            const uint32_t j = j0 + (uint32_t)lane;
            const uint32_t code = j < n ? S.list[j] : 0u;
            const uint32_t rid = code >> 5, l = code & 31u;
            const uint32_t pk = S.row[code];
            const uint4 R = S.rec[rid];
            const uint2 R2 = S.rec2[rid];
            const uint32_t mi = R.w + l;
            // in front of the record's first broken link: the record's base; behind it: the base gathered there
            const bool behind = l >= chain_first_break(S.link[rid]);
            uint32_t data = (behind ? R2.x : R.x) + (pk & 0xFFFFu), vtx = behind ? R2.y : R.y;
            if constexpr (!FAST) {
                const bool broken = second_break(j);
                if (__ballot(broken) != 0ull) { // rare
                    const uint32_t co = broken ? (mi - B.first) * 12u : kNoOffset;
                    const uint32_t gv = __builtin_amdgcn_raw_buffer_load_b32(B.cmd, co, 0, 0);
                    const uint32_t gd = __builtin_amdgcn_raw_buffer_load_b32(B.cmd, co, 0 + 4, 0);
                    vtx = broken ? gv : vtx;
                    data = broken ? gd : data;
                }
            }
            uint32_t vertex_count = (pk >> 16) & 0xFFu, triangle_count = pk >> 24;
            // ORBIT_E_RANGE (latched by the evaluation): the command of a meshlet outside the stream carries zeros
            if (!p.ms.covers(mi)) vtx = data = vertex_count = triangle_count = 0u;
            uint32_t *c = S.cmd + lane * 7;
            c[0] = triangle_count * 3u;         // cmd_index_count
            c[1] = 1u;                          // cmd_instance_count
            c[2] = (data + vertex_count) * 4u;  // cmd_first_index
            c[3] = data;                        // cmd_vertex_offset = int(data_offset)
            c[4] = R.z;                         // cmd_first_instance = entity_index
            c[5] = vtx;                         // meshlet_vertex_offset
            c[6] = mi;                          // meshlet_index
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // the trip's commands as one range-checked window: words past the list or the capacity are dropped
            const uint32_t first = base + j0; // slot of the trip's first command
            const uint32_t room = first < p.draw_capacity ? p.draw_capacity - first : 0u;
            const uint32_t left = (j0 < n && !slow) ? n - j0 : 0u;
            const uint32_t bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(min(min(left, 64u), room) * 28u));
            const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)first * 7u, 0, bytes, kBufFlags);
#pragma unroll
            for (uint32_t k = 0; k < 7; k++) {
                const uint32_t w = k * 64u + (uint32_t)lane;
                __builtin_amdgcn_raw_buffer_store_b32(S.cmd[w], win, w * 4u, 0, ORBIT_EMIT_STORE_AUX);
            }
            __builtin_amdgcn_wave_barrier(); // cmd is rewritten by the next trip
        };
        if constexpr (FAST) {
            trip(0u);
            trip(64u);
        } else {
            for (uint32_t j0 = 0u; j0 < n; j0 += 64u) trip(j0);
        }
        __builtin_amdgcn_wave_barrier(); // list / row / rec are reused by the next group
        return slow;
    };

    // The pipelined loop over the wave's groups: records and ballots three groups ahead, link words two, counts and
    // bases one.  A group past the end reads zero records: nothing else is loaded for it.
    // Which groups: a static grid-stride share first, tickets for the rest — as in the evaluation (meshlet_eval.hip).
    // With six groups per wave handed out statically the workgroups of a CU finished between 35 and 58 us of a 58-us
    // launch (per-workgroup stamps: the later a workgroup was dispatched, the later), i.e. the last third of the launch
    // ran at falling occupancy.  A claim is made four groups before the group is processed, so its round trip is never
    // waited for; pool = wave mod pools (a returning atomic on one word serves ~88 per us).
    // (the waves of the first workgroups — the ones with scan duty — are the ones with a group less where the static
    // groups do not divide evenly)
    const uint32_t wave_g = blockIdx.x * 4 + wave;
    const uint32_t group0 = stride - 1u - wave_g;
    const uint32_t full_rounds = ngroups / stride;
    const uint32_t n_static = full_rounds < 4u ? 0xFFFFFFFFu : max(full_rounds - min(max(full_rounds / 4u, 1u), 3u), 4u);
    const uint32_t npools = min(kEmitTicketPools, stride), pool = wave_g % npools;
    uint32_t *ticket_ctr = p.tickets + (kTicketPools + pool) * kTicketStride;
    constexpr uint32_t kNoGroup = 0x007FFFFFu, kNoTicket = 0x00400000u;
    bool dry = false, capped = false; // wave-uniform: the pool has no group left / this wave has had kDynGroups of them
    auto dyn_group = [&](uint32_t t) { return t >= kNoTicket ? kNoGroup : n_static * stride + t * npools + pool; };
    auto claim = [&](uint32_t iter) -> uint32_t { // for iteration `iter` of this wave; lane 0's VGPR carries the answer
        uint32_t raw = kNoTicket;
        if (iter >= n_static && !dry) {
            if (iter - n_static < kDynGroups) {
                raw = 0u;
                if (lane == 0) raw = atomicAdd(ticket_ctr, 1u);
            } else {
                capped = true;
            }
        }
        return raw;
    };
    uint32_t it = 0;
    uint32_t g_cur = group0, g_nxt = group0 + stride, g_nn = group0 + 2u * stride; // (n_static >= 4)
    uint32_t tk_raw = kNoTicket;                                                   // iteration 3 is static
    ChainGroup cur = chain_prefetch(B, g_cur, lane, false);
    ChainGroup nxt = chain_prefetch(B, g_nxt, lane, false);
    ChainGroup nn = chain_prefetch(B, g_nn, lane, false);
    ChainStage cs = chain_stage(B, S, cur, link_stage(B, cur), lane);
    LinkStage nk = link_stage(B, nxt);
    if (lane < (int)kSlowWords) S.slow[lane] = 0u;
    // the launch's ramp above ran under the scan of the first workgroups; now its result
    const uint32_t grand_total = emit_scan_wait(p, ntiles, nchunks, s_chunk_base, s_scan);
    {
        auto base_of = [&](uint32_t g) {
            return __builtin_amdgcn_raw_buffer_load_b32(B.bases, (g < kNoGroup ? g : kNoGroup) * kGroupTiles * 4u, 0, kSc1);
        };
        cur.base = base_of(g_cur);
        nxt.base = base_of(g_nxt);
        nn.base = base_of(g_nn);
    }
    auto step = [&]() __attribute__((always_inline)) {
        // the group three iterations ahead: static, or the ticket claimed one iteration ago
        uint32_t g3 = group0 + (it + 3u) * stride;
        if (it + 3u >= n_static) {
            g3 = dyn_group((uint32_t)__builtin_amdgcn_readfirstlane((int)tk_raw));
            if (g3 >= ngroups) dry = true;
            else if (lane == 0) S.dyn[(it + 3u - n_static) & (kDynGroups - 1u)] = g3; // (for the slow pass below)
        }
        tk_raw = claim(it + 4u);
        const ChainGroup n3 = chain_prefetch(B, g3, lane);
        const LinkStage nnk = link_stage(B, nn);
        const ChainStage ns = chain_stage(B, S, nxt, nk, lane);
        const bool slow = process(std::true_type{}, cur, cs, g_cur);
        // (the launchers size the grid so that a wave never has more than 32 * kSlowWords iterations)
        if (slow && lane == 0) S.slow[(it >> 5) & (kSlowWords - 1u)] |= 1u << (it & 31u);
        cur = nxt;
        cs = ns;
        nxt = nn;
        nk = nnk;
        nn = n3;
        g_cur = g_nxt;
        g_nxt = g_nn;
        g_nn = g3;
        it++;
    };
    // The first group outside the loop: the loop is then entered with its steady-state queue of loads and stores in
    // flight, and the compiler's wait counts inside it are the steady state's (from the prologue alone the link
    // words would look six operations old, and every iteration would drain its predecessor's stores to be safe).
    if (g_cur < ngroups) {
        step();
        while (g_cur < ngroups) step();
    }
    // The groups the loop left: one at a time, loads awaited as they come.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    auto one_group = [&](uint32_t g) {
        const ChainGroup e = chain_prefetch(B, g, lane);
        const ChainStage st = chain_stage(B, S, e, link_stage(B, e), lane);
        process(std::false_type{}, e, st, g);
    };
    for (uint32_t w = 0; w < kSlowWords; w++) {
        uint32_t bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.slow[w]);
        while (bits != 0u) {
            const uint32_t i = w * 32u + (uint32_t)__builtin_ctz(bits); // the wave's i-th iteration
            bits &= bits - 1u;
            one_group(i < n_static ? group0 + i * stride
                                   : (uint32_t)__builtin_amdgcn_readfirstlane((int)S.dyn[(i - n_static) & (kDynGroups - 1u)]));
        }
    }
    // a wave that has had its kDynGroups ticketed groups while its pool still had some (a launch with few waves running):
    // the rest of the pool, one group at a time
    while (capped) {
        uint32_t raw = 0u;
        if (lane == 0) raw = atomicAdd(ticket_ctr, 1u);
        const uint32_t g = dyn_group((uint32_t)__builtin_amdgcn_readfirstlane((int)raw));
        if (g >= ngroups) break;
        one_group(g);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // `count` header: subsumes the fill_buffer(count = 0) of draw_gen.rs:411-417
        *p.total = grand_total;
        *reinterpret_cast<uint32_t *>(p.draw_buffer) = min(grand_total, p.draw_capacity);
        if (grand_total > p.draw_capacity) latch_status(p.status, ORBIT_E_CAPACITY);
    }
}

__global__ __launch_bounds__(256) void meshlet_emit_kernel(const MeshletCullParams p) {
    meshlet_emit_body(p);
}
__global__ __launch_bounds__(256) void meshlet_emit_views_kernel(const MeshletCullViews all, const ViewGroup g) {
    meshlet_emit_body(all.v[g.idx[blockIdx.y]]);
}
// kChainWavesPerSimd resident waves per SIMD (three are 2 % slower, a fifth has neither the registers nor the LDS; the
// launch is bound by its bytes: profiles/r04_notes.md §6) — the register budget the
// compiler gets) and exactly that many blocks per CU in the grid — a block more per CU would run as a second round
constexpr int kChainWavesPerSimd = 4;
__global__ __launch_bounds__(256, kChainWavesPerSimd) void meshlet_emit_chain_kernel(const MeshletCullParams p) {
    meshlet_emit_chain_body(p);
}
__global__ __launch_bounds__(256, kChainWavesPerSimd) void meshlet_emit_chain_views_kernel(const MeshletCullViews all,
                                                                                          const ViewGroup g) {
    meshlet_emit_chain_body(all.v[g.idx[blockIdx.y]]);
}

} // namespace

// The emit launch behind evaluation + scan of one cull.  stream: the evaluation read the derived streams (no payload).
hipError_t launch_meshlet_emit(const MeshletCullParams &p, bool stream, uint32_t num_cus, hipStream_t s) {
    const uint32_t max_tiles = (p.dispatch_capacity + kTileRecords - 1) / kTileRecords;
    uint32_t eblocks = num_cus * 8u;
    const uint32_t eneed = (max_tiles + 3) / 4;
    if (eblocks > eneed) eblocks = eneed;
    if (eblocks == 0) eblocks = 1;
    if (stream) {
        // a wave per group of two tiles; never more than 32 * kSlowWords groups per wave
        const uint32_t gneed = (eneed + kGroupTiles - 1) / kGroupTiles;
        const uint32_t cblocks = max(max(min(num_cus * (uint32_t)kChainWavesPerSimd, gneed), (gneed + 32u * kSlowWords - kDynGroups - 1u) / (32u * kSlowWords - kDynGroups)), 1u);
        hipLaunchKernelGGL(meshlet_emit_chain_kernel, dim3(cblocks), dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL(meshlet_emit_kernel, dim3(eblocks), dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

// orbit_cull_views: one emit launch for the m views g.idx[0..m) of `all`, of n views in the call; their evaluations
// all kept a payload (stream = false) or all kept only ballots (true).
hipError_t launch_meshlet_emit_views(const MeshletCullViews &all, const ViewGroup &g, uint32_t m, uint32_t n, bool stream,
                                     uint32_t max_tiles_all, uint32_t num_cus, hipStream_t s) {
    uint32_t eblocks = max(num_cus * 8u / n, 1u);
    const uint32_t eneed = (max_tiles_all + 3) / 4;
    if (eblocks > eneed) eblocks = eneed;
    if (eblocks == 0) eblocks = 1;
    if (stream)
        hipLaunchKernelGGL(meshlet_emit_chain_views_kernel,
                           dim3(max(max(min(num_cus * (uint32_t)kChainWavesPerSimd / m, (eneed + kGroupTiles - 1) / kGroupTiles),
                                        ((eneed + kGroupTiles - 1) / kGroupTiles + 32u * kSlowWords - kDynGroups - 1u) / (32u * kSlowWords - kDynGroups)), 1u), m),
                           dim3(256), 0, s, all, g);
    else hipLaunchKernelGGL(meshlet_emit_views_kernel, dim3(eblocks, m), dim3(256), 0, s, all, g);
    return hipGetLastError();
}

} // namespace orbit
