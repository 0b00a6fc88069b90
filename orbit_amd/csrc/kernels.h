// kernels.h — host-callable launchers of the gfx950 kernels (internal to the
// library; the public surface is include/orbit_abi.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orbit_abi_ext.h"
#include "orbit_device.h"

namespace orbit {

// Meshlet-cull wave tile: 16 dispatch records (= 8 rows of 2 records on the
// 64 lanes of a wave, one meshlet per lane per row) = 512 meshlets.
constexpr uint32_t kTileRecords = 16;
constexpr uint32_t kTileRows = 8;
// meshlet_eval hands tiles out through kTicketPools counters (one 256-B line each); zero between evaluations
constexpr uint32_t kTicketPools = 8;
constexpr uint32_t kTicketStride = 64; // words
// ... followed by kEmitTicketPools counters of the chain emit's group tickets (meshlet_emit.hip), zeroed by the evaluation
constexpr uint32_t kEmitTicketPools = 16;
// The record list (visible_list == 2) is written by the evaluation itself — no launch follows it that could write the
// list's header or zero the tile tickets, so the evaluation's LAST workgroup does (meshlet_common.h records_finish):
// kListGroups 8-byte arrival words {arrivals << 40 | survivors} (workgroup b arrives at word b % kListGroups; the last one
// of a group carries the group's sum to the top word: nobody queues behind a thousand returning atomics on one word),
// every word on a 128-B line of its own; zero between launches (the last workgroup leaves them so).
constexpr uint32_t kListGroups = 32;
constexpr uint32_t kListSyncStride = 32; // words
constexpr uint32_t kListSyncWords = (kListGroups + 1) * kListSyncStride;

// what meshlet_eval keeps per survivor for the emit launch (meshlet_cull.hip make_payload)
struct Payload {
    uint32_t y, z, w; // Meshlet::vertex_offset, Meshlet::data_offset, {triangles:8 | vertices:8 | lane:5 << 4 | record:4}
};

// What the entity stage reads of a mesh, as a derived 32-B copy of the static 128-B MeshInfo
// (orbit_meshlet_stream_update_meshes): the entity test needs the bounding sphere (16 B at offset 0), lod_count (4 B at
// 56) and the chosen MeshLod (8 B at 64 + 8 lod) — 28 bytes of a 128-B line per entity-draw, the line being what HBM
// delivers (the launch moved 3.4 x its algorithmic bytes: profiles/r03_rdreq_sizes.txt).  With the table a mesh costs a
// quarter of a line; the MeshInfo itself is read only where a LOD other than 0 is picked, or where no update reached the
// mesh (valid == 0: the tables are zero-filled).
struct MeshSide {
    float sphere[4];
    uint32_t lod_count, lod0_offset, lod0_count, valid;
};
static_assert(sizeof(MeshSide) == 32, "MeshSide layout");
struct MeshSideView {
    const MeshSide *table; // nullptr: none (or it mirrors another mesh_info buffer)
    uint32_t count;        // meshes [0, count) have an entry
};

struct EntityCullParams {
    OrbitGpuCullInfo ci;
    const uint8_t *entity_draw_buffer;
    const OrbitMeshInfo *mesh_infos;
    MeshSideView mesh_side;
    uint8_t *dispatch_buffer;
    const OrbitEntityData *entities;
    uint32_t *visibility;
    PyramidView pyr;
    uint32_t draw_first;      // first entity-draw of this launch (multiple of 32)
    uint32_t draw_limit;      // draw_first + 256 * ceil(entity_draw_count / 256)
    uint32_t dispatch_capacity;
    uint32_t ne_chunks;       // ceil(entity_draw_count / 256): chunks of the launch (entity_common.h entity_chunks)
    uint32_t rec_shift;       // log2 of MESHLET_DISPATCH_SIZE: 5 (OrbitCaps.dispatch_size 32), 6 / 7 for 64 / 128 (pass 0)
    uint32_t arith;           // OrbitCaps.arith_profile: != 0 sends the launch to the contracted build of the kernels
    // context scratch of the two-launch form (entity_cull.hip: more than 128 chunks)
    OrbitMeshletDispatch *proto; // one proto-record per entity-draw of the launch
    uint32_t *block_sums; // records per block of 256 entity-draws; the emit launch prefix-sums them itself
    // context scratch of the one-launch form, zero between launches (the last workgroup of a launch leaves it so)
    uint32_t *sync;       // kFusedSyncWords counters (below)
    uint32_t *ent_flags;  // per chunk: records << 2 | state
    uint32_t *total;
    int32_t *status;
};

// Derived meshlet streams (orbit_meshlet_stream_*): the static 32-B Meshlet records re-laid out once, at upload, into
// what the streaming evaluation reads for EVERY meshlet (bounding sphere 16 B, cone 4 B, material index 2 B: 22 B) and
// what only a survivor's command needs.  The command words exist twice: `cmd` (12 B per meshlet: vertex_offset,
// data_offset, word 7) for gathers, and the chain form the emit launch streams — `cnt` (2 B: vertex_count |
// triangle_count << 8) plus one `link` bit per meshlet that says "this meshlet's vertex_offset equals its
// predecessor's and its data_offset is the predecessor's plus the predecessor's data size", which is how
// compute_meshlets / add_mesh lay a mesh out (src/assets/mesh.rs:309-316, assets/mod.rs:413-416).  Where the links
// hold, a record's offsets follow from ONE gathered base and a prefix sum over 2-B counts; where they do not, the
// survivor's 12 B are gathered from `cmd`.  Pointers are based at global meshlet index 0 (`link`: word 0 = meshlets
// 0..31); sphere == nullptr: no stream, evaluate the Meshlet buffer.
struct MeshletStreamView {
    const uint4 *sphere;  // Meshlet::bounding_sphere bits
    const uint32_t *cone; // cone_axis | cone_cutoff << 24
    const uint16_t *mat;  // material_index
    const uint32_t *cmd;  // 3 words per meshlet: vertex_offset, data_offset, word 7 again
    const uint16_t *cnt;  // vertex_count | triangle_count << 8
    const uint32_t *link; // bit (i & 31) of word (i >> 5): meshlet i continues meshlet i - 1's chain
    // {vertex_offset, data_offset} of every meshlet whose global index is a multiple of 32, addressed like `link`
    // (entry i >> 5): where a dispatch record starts on such an index — every record of a mesh whose meshlets start on
    // one — the chain's base is a dense 8-B read instead of a 128-B line gathered for 8 B out of `cmd`
    const uint2 *base32;
    // Alpha classes (orbit_meshlet_stream_set_materials): the alpha_mode of the meshlet's material as two bit planes
    // (class = bit of cls0 | bit of cls1 << 1, addressed like `link`); 0..2 = that alpha_mode, 3 = look the material
    // up.  Non-null only for a cull whose material buffer is the one the classes were derived from AND whose stream
    // is known to hold no meshlet of class 3: the evaluation then reads 20.25 B per meshlet instead of 22 and never
    // touches a material.
    const uint32_t *cls0, *cls1;
    uint32_t first, count; // global meshlet indices the arrays cover: [first, first + count)
    __host__ __device__ bool covers(uint32_t meshlet) const { return meshlet - first < count; }
};

// words of meshlet data a meshlet occupies: its vertex indices, then its triangle bytes rounded up to words
// (src/assets/mesh.rs:310-316)
__host__ __device__ inline uint32_t meshlet_data_words(uint32_t vertex_count, uint32_t triangle_count) {
    return vertex_count + ((triangle_count * 3u + 3u) >> 2);
}

struct MeshletCullParams {
    OrbitGpuCullInfo ci;
    const uint8_t *dispatch_buffer;
    const OrbitMeshlet *meshlets;
    MeshletStreamView ms;
    uint8_t *draw_buffer;
    const OrbitEntityData *entities;
    const OrbitMaterialData *materials;
    uint32_t *meshlet_visibility;
    PyramidView pyr;
    uint32_t dispatch_capacity;
    uint32_t draw_capacity;
    uint32_t material_count; // 0 = unknown (gather alpha modes), else entries in `materials`
    // The frustum a perspective camera hands the cull (forward.rs:264-268: left, right, bottom, top, near of
    // frustum_planes_from_matrix) is (a,0,c,0), (-a,0,c,0), (0,b,c',0), (0,-b,c',0), (0,0,e,w) — recognised on the host,
    // bit for bit.  For finite view-space centres the five plane tests then need 5 products and 5 sums instead of 15 +
    // 15, with the same outcome (the dropped terms are +-0; meshlet_cull.hip planes_std).  std_planes = 0: any other set.
    uint32_t std_planes;
    float stdp[6]; // a, c, b, c', e, w
    // mesh-shading path (orbit_meshlet_task_cull): non-null = survivors stay with their record, one
    // OrbitMeshTaskRecord per dispatch record, and no draw list is produced
    OrbitMeshTaskRecord *task_records;
    // sharded engine (orbit_meshlet_cull_visible_records / orbit_cull_shard): draw_buffer is the record list (8-B header,
    // 12 B {entity, meshlet offset, should-draw ballot} per dispatch record) and draw_capacity its capacity
    uint32_t arith;        // OrbitCaps.arith_profile: != 0 sends the evaluation to the contracted build of the kernels
    uint32_t visible_list; // 0: commands; 2: the record list, written by the evaluation launch itself (1 was the 8-B item
                           // list, retired in round 6)
    // visible_list == 2 only (orbit_meshlet_cull_records_and_commands): non-null = the same evaluation ALSO ends in the
    // MeshletDrawCommandBuffer (scan + emit behind the record launches) — a shard's own indirect draws next to the list
    // it sends
    uint8_t *also_commands;
    uint32_t also_commands_capacity;
    // context scratch
    uint32_t *tile_counts;
    Payload *tile_payload; // 128 x 12 B per wave tile: survivors in tile-local order (make_payload, meshlet_cull.hip)
    uint32_t *tile_masks; // 16 words (8 x 64-bit should-draw ballots) per wave tile
    uint32_t *tile_base;   // chunk-local exclusive prefix of tile_counts
    uint32_t *chunk_sums;  // per kScanChunk tiles; the emit launch prefix-sums them itself (LDS)
    uint32_t *total;
    uint32_t *tickets;     // (kTicketPools + kEmitTicketPools) x kTicketStride words (tile tickets, emit group tickets)
    uint32_t *list_sync;   // kListSyncWords: arrival counters and survivor sums of an evaluation that writes the record list
    const void *zero_page; // 256 zero bytes of context scratch: target of inactive lanes' loads
    unsigned long long *debug_cycles; // ORBIT_SP_DEBUG & 8: per-phase s_memtime sums (8 slots)
    uint32_t debug_flags; // ORBIT_SP_DEBUG (perf triage only): 8 = per-phase cycle stamps (pass 0 only)
    uint32_t scan_patience; // chain emit: polls before a waiting workgroup scans a missing chunk itself
    int32_t *status;
};

// One pyramid: every level has its own base pointer and row pitch (texels) — a packed chain is the special case
// level[k] = pyramid + mip_offset[k], pitch[k] = max(w0 >> k, 1).
struct DepthReduceParams {
    const float *depth;
    uint32_t depth_pitch; // texels per row of the depth buffer
    uint32_t screen_w, screen_h;
    uint32_t w0, h0, mips;
    float *level[ORBIT_MAX_PYRAMID_MIPS];
    uint32_t pitch[ORBIT_MAX_PYRAMID_MIPS];
};
// update_multiple_depth_pyramids::<C> (draw_gen.rs:569-628): C pyramids, one launch pair (blockIdx.z = pyramid)
constexpr uint32_t kMaxPyramidBatch = 8;
struct DepthReduceBatch {
    DepthReduceParams p[kMaxPyramidBatch];
    uint32_t count;
    uint32_t *tickets; // context scratch: kMaxPyramidBatch x kDepthTicketWords, zero between launches (depth_reduce.hip)
};
// per pyramid: one arrival counter for the rows and one per row of tile workgroups (at most 64 rows take the in-launch
// tail), 64 B apart
constexpr uint32_t kDepthTicketWords = (1u + 64u) * 16u;

struct ClusterMarkParams {
    OrbitMarkActivePush pc;
    const float *depth;
    uint32_t *masks;
    OrbitClusterDepthBounds *bounds;
    // orbit_compute_clusters: the launch also counts the active clusters per (z slice, chunk of 1024 tiles) — what the
    // compaction's first launch computes from the masks — into these words (zero before the launch; nullptr: no counting)
    uint32_t *chunk_counts;
};

struct ClusterCompactParams {
    uint32_t cc[3];
    const uint32_t *masks;
    uint8_t *unique;
    uint32_t index_capacity;
    uint32_t *chunk_counts; // cz * nchunks
    int32_t *status;
    bool counted;           // chunk_counts already holds the counts (the mark launch took them): one launch
};

// light indices the count launch keeps per cluster (256 B of scratch per cluster of caps.max_clusters): the write launch
// copies the lists of clusters that found at most this many (thread = cluster, kPlaceDirect words a round) and tests only
// the others again
constexpr uint32_t kHitCache = 64, kPlaceDirect = 16;
// clusters of padding behind the per-cluster scratch arrays of the assignment (counts, bases, AABBs, hit cache): its
// kernels read whole blocks / rounds of clusters speculatively (light_cluster.hip asserts its reach against this)
constexpr uint32_t kAssignPad = 256;
constexpr uint32_t kChunkSumStride = 32; // words between the sums of two chunks of kScanChunk clusters: a 128-B line each
struct ClusterAssignParams {
    OrbitClusterCullInfo info;
    const uint8_t *unique;
    const OrbitClusterDepthBounds *bounds;
    const OrbitLightData *lights;
    uint8_t *light_index_buffer;
    uint32_t light_index_capacity;
    uint32_t *offset_image;
    uint32_t max_clusters;
    // context scratch
    float4 *view_lights;     // (x, y, z, r*r) per light in view space
    uint32_t *light_flags;   // bit i set = light i is a point light
    uint32_t *counts;        // per active cluster: min(lights found, ORBIT_MAX_LIGHTS_PER_CLUSTER)
    uint32_t *hit_cache;     // per active cluster: its first kHitCache light indices, kept by the counting launch
    float *aabb;             // 6 floats per active cluster (compacted order), written by the coarse launch
    uint32_t *coarse;        // per group of 256 active clusters: 16 segments x coarse_seg candidate light indices
    float4 *coarse_lights;   // ... and, slot for slot, the candidates' view-space lights (no gather in the count launch)
    uint32_t *coarse_counts; // 16 per group
    float *group_box;        // 8 words per group: union AABB (6), any-NaN flag, unused
    uint32_t *group_order;   // the groups in the order the count launch takes them (longest candidate lists first)
    uint32_t coarse_seg;     // capacity of one segment (= lights per wave of the coarse filter, a multiple of 64)
    uint32_t *block_sums;    // the sum of the (capped) counts of every chunk of kScanChunk active clusters, kChunkSumStride
                             // words apart (count launch: one atomic per block; cleared by the AABB launch)
    uint32_t *block_base;    // the count launch's heavy blocks (block indices; a cluster with more than kHitCache lights)
    uint32_t *total;         // word 0: how many of those
    uint32_t *zero_words;    // orbit_compute_clusters: the mark launch's chunk counts, cleared again by the first
    uint32_t zero_count;     // launch here (consumed by then; nullptr / 0: nothing)
    int32_t *status;
    unsigned long long *debug_tests; // -DORBIT_TRIAGE builds: executed sphere-box tests per level (tools/cluster_tests.py)
};

// One-launch cull (cull_fused.hip): entity stage and meshlet stage of one view in one persistent kernel.  The meshlet
// side is the launches' parameter block (Meshlet buffer only: m.ms empty, no task records, no visible list); the entity
// side adds what entity_cull reads besides (ci, entities and the dispatch buffer are m's).
// kSyEntDone is 8 bytes: chunks whose records are in memory << 40 | records so far (handoff.h arrive / arrived_*)
constexpr uint32_t kSyEntTicket = 0, kSyEntDone = 32, kSyTileTicket = 96, kSyBlocksDone = 128;
constexpr uint32_t kFusedSyncWords = 160; // the counters above, each on a 128-B line of its own
constexpr uint32_t kFusedMaxEntityDraws = 16384; // culls of more entity-draws take the launches (abi.hip)
struct FusedCullParams {
    MeshletCullParams m;
    const uint8_t *entity_draw_buffer;
    const OrbitMeshInfo *mesh_infos;
    MeshSideView mesh_side;
    uint32_t *visibility; // entity bitset (occlusion passes)
    PyramidView e_pyr;    // the entity stage's pyramid (pass 2)
    uint32_t draw_first, draw_limit; // as EntityCullParams
    uint32_t e_dispatch_capacity;    // OrbitEntityCullBufs::dispatch_capacity (what the record stores are bounded by)
    uint32_t ne_chunks;              // ceil(entity_draw_count / 256) >= 1
    uint32_t *e_total;
    // context scratch, zero between launches (the last workgroup of a launch leaves it so)
    uint32_t *sync;       // kFusedSyncWords
    uint32_t *ent_flags;  // per entity chunk: records << 2 | state
    uint32_t *tile_flags; // per wave tile: survivors (or their inclusive prefix) << 2 | state
    uint32_t *done_flags; // shard launch only: kShardDoneWords x kShardDoneStride words, zero between launches
};
struct FusedCullViews {
    FusedCullParams v[ORBIT_MAX_CULL_VIEWS];
};
hipError_t launch_cull_fused_views(const FusedCullViews &all, const uint32_t *entity_draw_counts, uint32_t n,
                                   uint32_t num_cus, hipStream_t s);
// The shard launch of the sharded engine (orbit_cull_shard; meshlet_eval.hip): entity stage + evaluation + record list of
// one pass-0 view in ONE launch — f.m as for a cull with visible_list == 2 (a bound stream is used), the entity side as
// for the one-launch cull (f.tile_flags unused).  At most kShardMaxChunks chunks of 256 entity-draws: every chunk adds
// up the counts of the chunks before it, and beyond a few hundred the pair of entity launches is faster (r04 notes).
constexpr uint32_t kShardMaxChunks = 256;
// "every chunk is in" as the waiting workgroups of a shard launch read it: kShardDoneWords copies of {1 << 40 | records},
// each on a 128-B line of its own, stored by whoever completes the last chunk; workgroup b polls copy b % kShardDoneWords.
// (A thousand workgroups polling the arrival word itself put a thousand loads per microsecond on one line — the line the
// chunks' arrival atomics need: the entity phase took 20 us instead of 6.)
constexpr uint32_t kShardDoneWords = 64, kShardDoneStride = 32; // words
hipError_t launch_shard_cull(const FusedCullParams &f, uint32_t num_cus, hipStream_t s);
// scan + emit of an evaluation that has already run (the commands behind a shard launch's record list)
hipError_t launch_meshlet_scan_emit(const MeshletCullParams &p, uint32_t num_cus, hipStream_t s);

hipError_t launch_entity_cull(const EntityCullParams &p, uint32_t entity_draw_count, uint32_t num_cus, hipStream_t s);
// orbit_cull_views: the parameter blocks of all views travel BY VALUE in the kernel-argument segment (a few KB: HIP on
// gfx950 takes it — tools/ubench/kernarg_size.hip), so a view's block is read with scalar loads at a uniform offset
// (blockIdx.y), exactly like the single-view kernels read theirs.
// OrbitCaps.dispatch_size 64 / 128 (entity_cull.hip split_records_body): the caller's S-sized records -> records of 32
struct SplitRecordsParams {
    const uint8_t *src;    // the caller's dispatch buffer: {count,1,1} + 16-B records of up to 1 << rec_shift meshlets
    uint8_t *dst;          // context scratch: {count << (rec_shift - 5),1,1} + records of up to 32
    uint32_t src_capacity; // records `src` can hold
    uint32_t rec_shift;    // 6 / 7
};
struct SplitRecordsViews {
    SplitRecordsParams v[ORBIT_MAX_CULL_VIEWS];
};
hipError_t launch_split_records(const SplitRecordsParams &p, hipStream_t s);
hipError_t launch_split_records_views(const SplitRecordsViews &all, uint32_t n, hipStream_t s);

struct EntityCullViews {
    EntityCullParams v[ORBIT_MAX_CULL_VIEWS];
};
struct MeshletCullViews;
struct FusedCullViews;
struct ViewGroup;
#if !ORBIT_CONTRACT
// The translation units that evaluate geometry, compiled a second time with ORBIT_CONTRACT = 1 (*_contracted.hip: dots and
// matrix products as fma chains, OrbitCaps.arith_profile): the canonical launchers below hand a launch whose parameter
// block says `arith` to these.
hipError_t launch_entity_cull_contracted(const EntityCullParams &p, uint32_t entity_draw_count, uint32_t num_cus, hipStream_t s);
hipError_t launch_entity_cull_views_contracted(const EntityCullViews &all, uint32_t n, uint32_t max_entity_draw_count,
                                               uint32_t num_cus, hipStream_t s);
hipError_t launch_meshlet_eval_contracted(const MeshletCullParams &p, uint32_t num_cus, hipStream_t s);
hipError_t launch_meshlet_eval_views_contracted(const MeshletCullViews &all, const ViewGroup &g, uint32_t m, uint32_t pass,
                                                bool ortho, int src, uint32_t max_tiles, uint32_t num_cus, hipStream_t s);
hipError_t launch_shard_cull_contracted(const FusedCullParams &f, uint32_t num_cus, hipStream_t s);
hipError_t launch_cull_fused_views_contracted(const FusedCullViews &all, const uint32_t *entity_draw_counts, uint32_t n,
                                              uint32_t num_cus, hipStream_t s);
#endif
hipError_t launch_entity_cull_views(const EntityCullViews &all, uint32_t n, uint32_t max_entity_draw_count,
                                    uint32_t num_cus, hipStream_t s);
// ev0/ev1 (optional): recorded on `s` immediately before / after the dominant kernel of the op
hipError_t launch_meshlet_cull(const MeshletCullParams &p, uint32_t num_cus, hipStream_t s,
                               hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
// orbit_cull_views: the meshlet stage of the n views all.v[0..n)
struct MeshletCullViews {
    MeshletCullParams v[ORBIT_MAX_CULL_VIEWS];
};
hipError_t launch_meshlet_cull_views(const MeshletCullViews &all, uint32_t n, uint32_t num_cus, hipStream_t s);
// Views of one launch of the *_views kernels (orbit_cull_views): blockIdx.y = slot, idx[slot] = entry of the table.
struct ViewGroup {
    uint8_t idx[ORBIT_MAX_CULL_VIEWS];
};
// The launches launch_meshlet_cull / launch_meshlet_cull_views are made of (meshlet_eval.hip, meshlet_emit.hip,
// meshlet_lists.hip); grids are sized inside.
hipError_t launch_meshlet_eval(const MeshletCullParams &p, uint32_t num_cus, hipStream_t s);
hipError_t launch_meshlet_eval_views(const MeshletCullViews &all, const ViewGroup &g, uint32_t m, uint32_t pass,
                                     bool ortho, int src, uint32_t max_tiles, uint32_t num_cus, hipStream_t s);
hipError_t launch_meshlet_emit(const MeshletCullParams &p, bool stream, uint32_t num_cus, hipStream_t s);
hipError_t launch_meshlet_emit_views(const MeshletCullViews &all, const ViewGroup &g, uint32_t m, uint32_t n, bool stream,
                                     uint32_t max_tiles_all, uint32_t num_cus, hipStream_t s);
hipError_t launch_task_records(const MeshletCullParams &p, hipStream_t s);
// Largest dispatch capacity launch_meshlet_cull accepts (the emit launch keeps one base per scan chunk in LDS);
// checked by orbit_ctx_create against caps.max_dispatches, so no launch is ever enqueued for a larger one.
uint32_t max_dispatch_capacity();
// meshlets[first, first + count) -> the three streams (a plain re-layout, one pass)
// `ms`: the stream's arrays (writable through the const view) and its range; link bits are re-derived for
// [first, first + count] — the meshlet after the range chains from the range's last one.  `materials` (optional):
// the buffer the range's alpha classes are derived from (none: class 3, "look the material up").
hipError_t launch_meshlet_stream_build(const OrbitMeshlet *meshlets, uint64_t first, uint64_t count,
                                       const MeshletStreamView &ms, const OrbitMaterialData *materials,
                                       uint32_t material_count, uint32_t *other, hipStream_t s);
// ORBIT_E_STALE into *status if a meshlet of the stream's range differs from its derived copy (or, with `materials`,
// an alpha class from its material's alpha_mode): orbit_meshlet_stream_validate
hipError_t launch_meshlet_stream_validate(const OrbitMeshlet *meshlets, const MeshletStreamView &ms,
                                          const OrbitMaterialData *materials, uint32_t material_count, int32_t *status,
                                          hipStream_t s);
// the alpha classes of the whole stream, from its material indices (orbit_meshlet_stream_set_materials)
// meshes [first, first + count) of `mesh_infos` -> their 32-B side entries; and the comparison (ORBIT_E_STALE)
hipError_t launch_mesh_side_build(const OrbitMeshInfo *mesh_infos, uint32_t first, uint32_t count, MeshSide *table, hipStream_t s);
hipError_t launch_mesh_side_validate(const OrbitMeshInfo *mesh_infos, uint32_t first, uint32_t count, const MeshSide *table,
                                     int32_t *status, hipStream_t s);
hipError_t launch_meshlet_stream_classes(const MeshletStreamView &ms, const OrbitMaterialData *materials,
                                         uint32_t material_count, uint32_t *other, hipStream_t s);
// ms.cmd (optional): the derived command-word stream of `meshlets`, based at global meshlet index 0
// orbit_expand_visible_records: `records` = {records, survivors} header + 12-B {entity, meshlet_offset, mask} entries;
// block_pop: scratch, one word per 1024 records (max_blocks words)
hipError_t launch_visible_records_expand(const uint8_t *records, uint32_t *block_pop, uint32_t max_blocks,
                                         const OrbitMeshlet *meshlets, const MeshletStreamView &ms, uint8_t *draw_buffer,
                                         uint32_t draw_capacity, const void *zero_page, int32_t *status, hipStream_t s);
// orbit_compact_segments: `world` segments {count | header | segment_capacity items} -> {total | header | items}
hipError_t launch_compact_segments(const uint8_t *segments, uint32_t world, uint32_t segment_capacity, uint8_t *out,
                                   uint32_t out_capacity, uint32_t header_bytes, uint32_t stride, uint32_t num_cus,
                                   int32_t *status, hipStream_t s);
// orbit_exchange_list: scatter + device-side counts and completion (meshlet_lists.hip)
struct ExchangeListParams {
    const uint8_t *local_list;
    uint8_t *out[16];  // ORBIT_P2P_MAX_WORLD: rank r's output buffer as mapped here
    uint8_t *ctrl[16]; // rank r's control block as mapped here (ORBIT_P2P_CTRL_BYTES each, zero at first use)
    uint32_t rank, world, out_capacity, header_bytes, stride;
    int32_t *status;
};
hipError_t launch_exchange_list(const ExchangeListParams &p, uint32_t num_cus, hipStream_t s);
hipError_t launch_depth_reduce(const DepthReduceBatch &b, hipStream_t s);
hipError_t launch_cluster_mark(const ClusterMarkParams &p, hipStream_t s);
// tests: the hardware-log2 slice against the canonical one for every float in [lo, hi] (bit patterns); out: 3 x u64
hipError_t launch_log2_guard_check(uint32_t lo, uint32_t hi, float z_scale, float z_bias, unsigned long long *out, hipStream_t s);
hipError_t launch_cluster_compact(const ClusterCompactParams &p, hipStream_t s);
hipError_t launch_cluster_assign(const ClusterAssignParams &p, uint32_t num_cus, hipStream_t s);

// chunk bases the emit launches keep in LDS (1 Mi tiles = 16 Mi dispatch records)
constexpr uint32_t kEmitChunkTable = 1024;

// Exclusive scan of `n` words by one 1024-thread block; *total = sum.
hipError_t launch_scan_exclusive(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total, hipStream_t s);
// Two-level variant for long inputs: out_local[i] is the exclusive prefix inside
// chunk i / kScanChunk, chunk_base[c] the exclusive prefix of chunk c; n as below.
constexpr uint32_t kScanChunk = 1024; // 4096 -> 1024: four times the blocks, one pass each: -2 us per scan (profiles/r02_notes.md)
hipError_t launch_scan_two_level(const uint32_t *in, uint32_t *out_local, const uint32_t *n_src, uint32_t div,
                                 uint32_t n_max, uint32_t *chunk_sums, uint32_t *chunk_base, uint32_t *total,
                                 hipStream_t s);
// The first scan level of several inputs in one launch (blockIdx.y = input): orbit_cull_views
struct ScanViews {
    struct Item {
        const uint32_t *in;
        uint32_t *out_local;
        const uint32_t *n_src;
        uint32_t *chunk_sums;
        uint32_t *clear;
        uint32_t n_max;
    } v[8];
};
hipError_t launch_scan_chunks_views(const ScanViews &sv, uint32_t count, uint32_t div, uint32_t n_max_all,
                                    uint32_t clear_words, hipStream_t s);
// First level only (out_local and chunk_sums); the consumer scans the chunk sums itself.
// `clear` (optional): `clear_words` words zeroed by the launch as well (a fused memset for the caller's counters)
hipError_t launch_scan_chunks(const uint32_t *in, uint32_t *out_local, const uint32_t *n_src, uint32_t div,
                              uint32_t n_max, uint32_t *chunk_sums, hipStream_t s, uint32_t *clear = nullptr,
                              uint32_t clear_words = 0);

} // namespace orbit
