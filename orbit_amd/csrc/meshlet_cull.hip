// meshlet_cull.hip — the launch protocol of one meshlet cull (src/passes/draw_gen.rs:382-435 of the reference repo:
// clear, one dispatch of shaders/meshlet_cull.comp over the entity stage's records) as this library runs it:
//
//   evaluation (meshlet_eval.hip) -> scan of the tiles' survivor counts (scan.hip; inside the emit launch where the
//   commands come from the derived streams) -> emit (meshlet_emit.hip)
//
// or, for the outputs that are not a draw-command list, evaluation -> list launches (meshlet_lists.hip); the record list
// of the sharded engine is written by the evaluation itself (meshlet_common.h tile_records_out).  Tile bases
// come from a prefix sum, so the command sequence is the canonical (record, lane) order with no atomics.  (A
// single-launch variant with a per-tile decoupled look-back was measured slower on the 50 M-meshlet scene and removed:
// profiles/r01_notes.md; scenes of the reference's own size take the one-launch path of cull_fused.hip.)
#include "kernels.h"

namespace orbit {

uint32_t max_dispatch_capacity() {
    const uint64_t cap = (uint64_t)kEmitChunkTable * kScanChunk * kTileRecords;
    return cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;
}

namespace {

// Prefix sum over the tiles' survivor counts + emit.  The chain emit (commands from the derived streams) scans inside
// its own launch (meshlet_emit.hip emit_scan_duty); the others get the first scan level (chunk-local prefixes + chunk
// sums) as a launch and redo the second in LDS.
hipError_t scan_and_emit(const MeshletCullParams &p, uint32_t max_tiles, uint32_t num_cus, hipStream_t s) {
    // the list-driven pass 1 keeps its payload whatever the meshlet source
    const bool stream = p.ci.occlusion_pass != 1 && p.ms.sphere != nullptr;
    if (!stream) {
        const hipError_t e = launch_scan_chunks(p.tile_counts, p.tile_base, reinterpret_cast<const uint32_t *>(p.dispatch_buffer),
                                                kTileRecords, max_tiles, p.chunk_sums, s, p.tickets, kTicketPools * kTicketStride);
        if (e != hipSuccess) return e;
    }
    return launch_meshlet_emit(p, stream, num_cus, s);
}

} // namespace

hipError_t launch_meshlet_scan_emit(const MeshletCullParams &p, uint32_t num_cus, hipStream_t s) {
    const uint32_t max_tiles = (p.dispatch_capacity + kTileRecords - 1) / kTileRecords;
    if ((max_tiles + kScanChunk - 1) / kScanChunk > kEmitChunkTable) return hipErrorInvalidValue;
    return scan_and_emit(p, max_tiles, num_cus, s);
}

hipError_t launch_meshlet_cull(const MeshletCullParams &p, uint32_t num_cus, hipStream_t s, hipEvent_t ev0,
                               hipEvent_t ev1) {
    const uint32_t max_tiles = (p.dispatch_capacity + kTileRecords - 1) / kTileRecords;
    // nothing is enqueued for a capacity the emit launch cannot address (orbit_ctx_create rejects such caps already)
    if ((max_tiles + kScanChunk - 1) / kScanChunk > kEmitChunkTable) return hipErrorInvalidValue;
    hipError_t e;
    if (ev0 && (e = hipEventRecord(ev0, s)) != hipSuccess) return e;
    if ((e = launch_meshlet_eval(p, num_cus, s)) != hipSuccess) return e;
    if (ev1 && (e = hipEventRecord(ev1, s)) != hipSuccess) return e;
    if (p.task_records) return launch_task_records(p, s);
    if (p.visible_list == 2u) {
        // the record list is complete: the evaluation wrote its entries and its last workgroup the header
        if (p.also_commands == nullptr) return hipSuccess;
        // the same evaluation's commands as well: scan + emit as in a plain cull, into the second buffer
        MeshletCullParams c = p;
        c.visible_list = 0u;
        c.draw_buffer = p.also_commands;
        c.draw_capacity = p.also_commands_capacity;
        return scan_and_emit(c, max_tiles, num_cus, s);
    }
    return scan_and_emit(p, max_tiles, num_cus, s);
}

// orbit_cull_views, meshlet stage.  The evaluation launches are grouped by (occlusion pass, projection, source) — a
// frame's early pass and its four cascades are two launches — scan and emit take all views at once.
hipError_t launch_meshlet_cull_views(const MeshletCullViews &all, uint32_t n, uint32_t num_cus, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const MeshletCullParams *host = all.v;
    auto tiles_of = [&](uint32_t i) { return (host[i].dispatch_capacity + kTileRecords - 1) / kTileRecords; };
    uint32_t max_tiles_all = 0;
    for (uint32_t i = 0; i < n; i++) {
        if ((tiles_of(i) + kScanChunk - 1) / kScanChunk > kEmitChunkTable) return hipErrorInvalidValue;
        max_tiles_all = max(max_tiles_all, tiles_of(i));
    }
    // a view evaluates from the derived meshlet streams when its block carries them and its pass streams (0 / 2):
    // 0 = Meshlet buffer, 1 = streams, 2 = streams with alpha classes
    auto source = [&](uint32_t i) -> int {
        if (host[i].ci.occlusion_pass == 1 || host[i].ms.sphere == nullptr) return 0;
        return host[i].ms.cls0 != nullptr ? 2 : 1;
    };
    bool done[ORBIT_MAX_CULL_VIEWS] = {};
    for (uint32_t i = 0; i < n; i++) {
        if (done[i]) continue;
        const uint32_t pass = host[i].ci.occlusion_pass, proj = host[i].ci.projection_type;
        const int src = source(i);
        ViewGroup g{};
        uint32_t m = 0, group_tiles = 0;
        for (uint32_t j = i; j < n; j++)
            if (!done[j] && host[j].ci.occlusion_pass == pass && host[j].ci.projection_type == proj && source(j) == src) {
                done[j] = true;
                g.idx[m++] = (uint8_t)j;
                group_tiles = max(group_tiles, tiles_of(j));
            }
        const hipError_t e = launch_meshlet_eval_views(all, g, m, pass, proj == 1, src, group_tiles, num_cus, s);
        if (e != hipSuccess) return e;
    }
    hipError_t e = hipSuccess;
    // the emit launch of the views whose evaluation kept a payload (behind the scan launch of those views), and of
    // those that kept only ballots (the chain emit: it scans itself)
    for (int stream = 0; stream < 2; stream++) {
        ViewGroup g{};
        uint32_t m = 0;
        for (uint32_t i = 0; i < n; i++)
            if ((source(i) != 0) == (stream != 0)) g.idx[m++] = (uint8_t)i;
        if (m == 0) continue;
        if (stream == 0) {
            ScanViews sv{};
            for (uint32_t k = 0; k < m; k++) {
                const uint32_t i = g.idx[k];
                sv.v[k] = {host[i].tile_counts, host[i].tile_base, reinterpret_cast<const uint32_t *>(host[i].dispatch_buffer),
                           host[i].chunk_sums, host[i].tickets, tiles_of(i)};
            }
            e = launch_scan_chunks_views(sv, m, kTileRecords, max_tiles_all, kTicketPools * kTicketStride, s);
            if (e != hipSuccess) return e;
        }
        e = launch_meshlet_emit_views(all, g, m, n, stream != 0, max_tiles_all, num_cus, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

} // namespace orbit
