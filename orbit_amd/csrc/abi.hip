// abi.hip — implementation of the C ABI declared in include/orbit_abi.h.
//
// Argument validation mirrors the reference's host-side failure modes
// (assert!/unwrap in src/passes/draw_gen.rs:123-133,247,334,390) as status
// codes; nothing unwinds across the boundary.  There is no host fallback: a
// missing device or an unloadable code object is ORBIT_E_NO_DEVICE.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only: the symbols are resolved with dlsym (orbit_gather_visible)

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "kernels.h"

using namespace orbit;

// orbit_meshlet_stream_*: derived arrays of meshlets [first, first + capacity) of the buffer at `source`
struct OrbitMeshletStream {
    int device = 0;
    uint64_t first = 0, capacity = 0;
    uint4 *sphere = nullptr;
    uint32_t *cone = nullptr;
    uint16_t *mat = nullptr;
    uint32_t *cmd = nullptr;
    uint16_t *cnt = nullptr;
    uint32_t *link = nullptr, *cls0 = nullptr, *cls1 = nullptr; // word 0 = meshlets (first & ~31) .. +31
    uint2 *base32 = nullptr;                                     // entry 0 = meshlet (first & ~31)
    // A stream may be bound to several contexts, each with its own lock: its own state has its own.  The device-side
    // order of an update against the culls that read it is the caller's, like the order of the meshlet upload itself.
    std::mutex mu;
    const void *source = nullptr;          // Meshlet buffer of the updates so far (nullptr: never updated)
    uint64_t valid_lo = 0, valid_hi = 0;   // hull of the ranges derived from `source`: what a cull may read
    const void *materials = nullptr;       // orbit_meshlet_stream_set_materials: what the alpha classes mirror; read on
                                           // the DEVICE by every later update (the caller forgets it before freeing it)
    uint32_t material_count = 0;
    // "Some meshlet has class 3" (a material beyond the table, an alpha_mode > 2): written on the device by the launches
    // that derive classes, copied to pinned host memory behind them.  A cull reads the classes only once that copy has
    // landed and says no (hipEventQuery, never a wait); until then, and for streams with such meshlets, the
    // evaluation goes through the material indices, which is always right.
    uint32_t *d_other = nullptr, *h_other = nullptr;
    hipEvent_t other_event = nullptr;
    bool other_pending = false;  // a derivation is under way (or its outcome unknowable): do not trust the classes
    bool other_recorded = false; // other_event was recorded for the pending derivation (on a stream that really runs)
    std::atomic<int> bindings{0};          // contexts it is bound to (orbit_meshlet_stream_destroy refuses while > 0)
    // orbit_meshlet_stream_update_meshes: 32-B side entries of the meshes of `mesh_source` (kernels.h MeshSide);
    // allocated (zero-filled) by the first update, meshes [0, mesh_hi) may be read
    MeshSide *mesh_side = nullptr;
    uint32_t mesh_capacity = 0, mesh_hi = 0;
    const void *mesh_source = nullptr;
    uint64_t mesh_side_culls = 0;          // entity culls that were handed the table
};

// meshes the side table holds when none was asked for more: 32 MB (the reference's MAX_MESH_COUNT is 10 000)
constexpr uint32_t kMeshSideMinCapacity = 1u << 20;

struct OrbitCtx {
    int device = 0;
    uint32_t num_cus = 0;
    OrbitCaps caps{};
    std::mutex mu; // entry points are thread-safe per ctx (pass closures are Send + Sync, context.rs:617-620)
    // scratch (device)
    uint8_t *arena = nullptr;
    size_t arena_bytes = 0;
    OrbitMeshletDispatch *e_proto = nullptr;
    uint32_t *e_block_sums = nullptr, *e_total = nullptr;
    Payload *m_tile_payload = nullptr;
    uint32_t *m_tile_masks = nullptr, *m_chunk_sums = nullptr;
    uint32_t *m_tile_counts = nullptr, *m_tile_base = nullptr, *m_total = nullptr;
    uint32_t *x_block_pop = nullptr; // orbit_expand_visible_records: survivors per 1024 records of the list
    uint32_t *c_chunk = nullptr; // compact: its own chunk counts | the ones a counting mark launch left (c_chunk_words each)
    size_t c_chunk_words = 0;
    float4 *a_view_lights = nullptr, *a_coarse_lights = nullptr;
    uint32_t *a_light_flags = nullptr, *a_counts = nullptr, *a_block_sums = nullptr, *a_block_base = nullptr,
             *a_total = nullptr, *a_coarse = nullptr, *a_coarse_counts = nullptr, *a_hit_cache = nullptr;
    float *a_aabb = nullptr, *a_group_box = nullptr;
    uint32_t *a_group_order = nullptr;
    uint32_t a_coarse_seg = 0;
    uint32_t *m_tickets = nullptr, *m_list_sync = nullptr, *f_done = nullptr;
    uint32_t *f_sync = nullptr, *f_ent_flags = nullptr, *f_tile_flags = nullptr; // one-launch cull (cull_fused.hip)
    uint32_t *d_tickets = nullptr; // depth_reduce: one arrival counter per pyramid of a batch
    uint8_t *m_split = nullptr;    // dispatch_size 64 / 128: the caller's records as records of 32 (entity_cull.hip split_records_body)
    uint32_t rec_shift = 5;        // log2 of caps.dispatch_size
    uint64_t fused_culls = 0;                                                    // views culled by it so far
    uint64_t shard_culls = 0;                                                    // orbit_cull_shard calls that took ONE launch
    uint32_t *g_counts = nullptr; // gather_visible: per-rank command counts
    int32_t *status = nullptr;
    uint32_t debug_flags = 0;
    uint32_t scan_patience = 256; // meshlet_emit.hip emit_scan_wait (orbit_debug_set_scan_patience)
    unsigned long long *debug_cycles = nullptr;
    void *zero_page = nullptr;
    // measurement hook: HIP event pairs around the dominant kernel (meshlet_eval)
    uint32_t profiling = 0;   // 0 off, n: every n-th meshlet cull is timed
    uint32_t prof_calls = 0;
    std::vector<hipEvent_t> prof_events; // pairs, in record order
    size_t prof_used = 0;
    // orbit_cull_views: child contexts (own scan scratch) for views 1.., created on first use
    std::vector<OrbitCtx *> view_ctx;
    hipStream_t side_stream[2] = {nullptr, nullptr}; // orbit_frame_late: the chains beside the caller's stream
    hipEvent_t side_event[3] = {nullptr, nullptr, nullptr}; // fork, join of chain B, join of chain C
    OrbitMeshletStream *meshlet_stream = nullptr; // orbit_ctx_bind_meshlet_stream
    uint64_t stream_culls = 0;                          // meshlet culls launched from it
    uint64_t class_culls = 0;                           // ... of which with its alpha classes
    char err[512] = {0};
};

// orbit_expand_visible_records: a list of up to kExpandBlocks * 1024 records (64 M) can be expanded
constexpr uint32_t kExpandBlocks = 65536;

namespace {

thread_local char g_err[512] = "no error";

int32_t fail(OrbitCtx *ctx, int32_t code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    snprintf(g_err, sizeof(g_err), "%s", buf);
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s", buf);
    return code;
}

int32_t hip_fail(OrbitCtx *ctx, hipError_t e, const char *what) {
    return fail(ctx, ORBIT_E_HIP, "%s: %s", what, hipGetErrorString(e));
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// hipMemset of device memory returns before the fill has run (it is ordered on the null stream), and the streams the
// library is called on later need not wait for the null stream (hipStreamNonBlocking: torch's side streams are): a
// context's first cull could run while its scratch was still being cleared under it (found by
// tests/test_concurrent_gpu.py once the arena had grown by a gigabyte).  Every fill of memory that launches on OTHER
// streams will use is therefore waited for before the call that made it returns.
hipError_t memset_now(void *ptr, int value, size_t bytes) {
    hipError_t e = hipMemset(ptr, value, bytes);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}

uint32_t mip_levels_from_size(uint32_t max_size) { // src/math.rs:18-20
    uint32_t l = 0;
    while ((max_size >> (l + 1)) != 0) l++;
    return l + 1;
}

uint32_t next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

void fill_pyramid_desc(uint32_t w0, uint32_t h0, OrbitDepthPyramidDesc *d) {
    memset(d, 0, sizeof(*d));
    d->width = w0;
    d->height = h0;
    d->mip_levels = mip_levels_from_size(w0 > h0 ? w0 : h0);
    uint32_t off = 0;
    for (uint32_t k = 0; k < d->mip_levels && k < ORBIT_MAX_PYRAMID_MIPS; k++) {
        d->mip_offset[k] = off;
        d->mip_width[k] = (w0 >> k) ? (w0 >> k) : 1; // image.rs:533
        d->mip_height[k] = (h0 >> k) ? (h0 >> k) : 1;
        off += d->mip_width[k] * d->mip_height[k];
    }
    d->total_texels = off;
}

int32_t check_cull_info(OrbitCtx *ctx, const OrbitGpuCullInfo *ci) {
    if (!ci) return fail(ctx, ORBIT_E_INVALID, "cull_info is NULL");
    if (ci->cull_plane_count > ORBIT_MAX_CULL_PLANES) // assert!, draw_gen.rs:247,334,390
        return fail(ctx, ORBIT_E_PLANES, "cull_plane_count %u > %d", ci->cull_plane_count, ORBIT_MAX_CULL_PLANES);
    if (ci->projection_type > 1) return fail(ctx, ORBIT_E_INVALID, "projection_type %u", ci->projection_type);
    if (ci->occlusion_pass > 2) return fail(ctx, ORBIT_E_INVALID, "occlusion_pass %u", ci->occlusion_pass);
    return ORBIT_OK;
}

// The stream's arrays based at global meshlet index 0, like meshlet_buffer itself, over the range derived so far.
MeshletStreamView stream_arrays(const OrbitMeshletStream *ms, uint64_t lo, uint64_t hi) {
    MeshletStreamView v{};
    v.sphere = ms->sphere - ms->first;
    v.cone = ms->cone - ms->first;
    v.mat = ms->mat - ms->first;
    v.cmd = ms->cmd - 3u * ms->first;
    v.cnt = ms->cnt - ms->first;
    v.link = ms->link - (ms->first >> 5);
    v.base32 = ms->base32 - (ms->first >> 5);
    v.cls0 = ms->cls0 - (ms->first >> 5);
    v.cls1 = ms->cls1 - (ms->first >> 5);
    v.first = (uint32_t)lo;
    v.count = (uint32_t)(hi - lo);
    return v;
}

// What a cull of `meshlet_buffer` (with `material_buffer`) may take from the stream bound to its context: nothing
// unless the stream mirrors that very buffer; the alpha classes only if they mirror that very material buffer.
MeshletStreamView stream_view_for(OrbitMeshletStream *ms, const void *meshlet_buffer, const void *material_buffer) {
    if (!ms) return MeshletStreamView{};
    std::lock_guard<std::mutex> lock(ms->mu);
    if (ms->source == nullptr || ms->source != meshlet_buffer || ms->valid_hi == ms->valid_lo) return MeshletStreamView{};
    MeshletStreamView v = stream_arrays(ms, ms->valid_lo, ms->valid_hi);
    bool classes = ms->materials != nullptr && ms->materials == material_buffer;
    // (no query while the flag's read-back was only captured into a graph, not run: other_recorded is false then)
    if (classes && ms->other_pending && ms->other_recorded && hipEventQuery(ms->other_event) == hipSuccess)
        ms->other_pending = false;
    if (classes && (ms->other_pending || *ms->h_other != 0u)) classes = false;
    if (!classes) v.cls0 = v.cls1 = nullptr;
    return v;
}

// The mesh side table an entity cull of `mesh_info_buffer` may read: only the one derived from that very buffer.
MeshSideView mesh_side_for(OrbitMeshletStream *ms, const void *mesh_info_buffer) {
    if (!ms) return MeshSideView{nullptr, 0u};
    std::lock_guard<std::mutex> lock(ms->mu);
    if (ms->mesh_side == nullptr || ms->mesh_source != mesh_info_buffer || ms->mesh_hi == 0u) return MeshSideView{nullptr, 0u};
    ms->mesh_side_culls++;
    return MeshSideView{ms->mesh_side, ms->mesh_hi};
}

// Enqueues the read-back of the stream's class flag behind the launch that may have set it.  On a stream that is being
// captured into a graph nothing runs now and an event query would invalidate the capture: the outcome stays unknown
// (the culls keep reading material indices) until a derivation runs on a live stream.
hipError_t read_back_class_flag(OrbitMeshletStream *ms, hipStream_t s) {
    ms->other_pending = true;
    ms->other_recorded = false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return hipSuccess;
    hipError_t e = hipMemcpyAsync(ms->h_other, ms->d_other, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipEventRecord(ms->other_event, s);
    if (e == hipSuccess) ms->other_recorded = true;
    return e;
}

PyramidView make_pyramid_view(const float *texels, const uint32_t size[2],
                              const OrbitDepthPyramidLevel *levels = nullptr) {
    PyramidView v;
    v.texels = texels;
    v.levels = levels;
    v.w0 = size[0];
    v.h0 = size[1];
    v.mips = (size[0] | size[1]) ? mip_levels_from_size(size[0] > size[1] ? size[0] : size[1]) : 0;
    return v;
}

} // namespace

extern "C" {

uint32_t orbit_abi_version(void) { return ORBIT_ABI_VERSION; }

void orbit_default_caps(OrbitCaps *caps) {
    if (!caps) return;
    memset(caps, 0, sizeof(*caps));
    caps->max_entities = 100000;    // src/scene.rs:303
    caps->max_dispatches = 1000000; // src/passes/draw_gen.rs:16
    caps->max_draws = 1000000;      // src/passes/draw_gen.rs:15
    caps->max_lights = 2000;        // src/scene.rs:304
    caps->max_clusters = 240 * 135 * 32; // 1920x1080, 8 px tiles, 32 slices (cluster.rs:23-33)
    caps->dispatch_size = ORBIT_MESHLET_DISPATCH_SIZE;
}

const char *orbit_last_error(const OrbitCtx *ctx) { return ctx ? ctx->err : g_err; }

int32_t orbit_ctx_create(int32_t device_id, const OrbitCaps *caps_in, OrbitCtx **out_ctx) {
    if (!out_ctx) return fail(nullptr, ORBIT_E_INVALID, "out_ctx is NULL");
    *out_ctx = nullptr;
    OrbitCaps caps;
    if (caps_in) caps = *caps_in;
    else orbit_default_caps(&caps);
    if (caps.dispatch_size != 32u && caps.dispatch_size != 64u && caps.dispatch_size != 128u)
        return fail(nullptr, ORBIT_E_INVALID, "dispatch_size %u: 32, 64 or 128 (src/graphics/device.rs:369-372)",
                    caps.dispatch_size);
    const uint32_t rec_shift = caps.dispatch_size == 32u ? 5u : caps.dispatch_size == 64u ? 6u : 7u;
    // the meshlet stage works on records of 32 whatever the caller's records hold: its scratch is sized for those
    const uint64_t md32 = (uint64_t)caps.max_dispatches << (rec_shift - 5u);
    if (md32 > max_dispatch_capacity())
        return fail(nullptr, ORBIT_E_CAPACITY, "caps.max_dispatches %u x dispatch_size %u / 32 > %u (chunk-base table of the emit launch)",
                    caps.max_dispatches, caps.dispatch_size, max_dispatch_capacity());
    if (caps.arith_profile > ORBIT_ARITH_CONTRACTED)
        return fail(nullptr, ORBIT_E_INVALID, "arith_profile %u (0 canonical, 1 contracted)", caps.arith_profile);
    if (caps.max_dispatches > max_dispatch_capacity())
        return fail(nullptr, ORBIT_E_CAPACITY, "caps.max_dispatches %u > %u (chunk-base table of the emit launch)",
                    caps.max_dispatches, max_dispatch_capacity());
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev)
        return fail(nullptr, ORBIT_E_NO_DEVICE, "no usable HIP device %d (count %d): %s; there is no CPU fallback",
                    device_id, ndev, e == hipSuccess ? "ok" : hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceProperties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, ORBIT_E_NO_DEVICE, "device %d is %s; this library carries gfx950 code objects only",
                    device_id, prop.gcnArchName);
    e = hipSetDevice(device_id);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipSetDevice");

    OrbitCtx *ctx = new (std::nothrow) OrbitCtx();
    if (!ctx) return fail(nullptr, ORBIT_E_HIP, "out of host memory");
    ctx->device = device_id;
    ctx->num_cus = (uint32_t)prop.multiProcessorCount;
    ctx->caps = caps;

    // one arena for all scan scratch, carved at 256-B alignment
    const size_t ent = align_up((size_t)caps.max_entities, 256);
    const size_t mtiles = ((size_t)md32 + kTileRecords - 1) / kTileRecords + 1;
    const size_t cchunks = ((size_t)caps.max_clusters / 1024 + 64) * 32;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t o = off;
        off = align_up(off + bytes, 256);
        return o;
    };
    const size_t o_status = carve(256);
    const size_t o_dbg = carve(256 + 16 * 8192); // phase sums + per-wave {begin, end} wall-clock stamps (triage builds)
    const size_t o_zero = carve(256);
    const size_t o_gc = carve(256);
    const size_t o_tk = carve((kTicketPools + kEmitTicketPools) * kTicketStride * 4);
    const size_t o_ls = carve(kListSyncWords * 4);
    const size_t o_sd = carve((size_t)kShardDoneWords * kShardDoneStride * 4);
    const size_t o_fsy = carve(kFusedSyncWords * 4), o_fef = carve((ent / 256 + 2) * 4);
    const size_t o_dtk = carve((size_t)kMaxPyramidBatch * kDepthTicketWords * 4);
    const size_t o_eproto = carve(ent * sizeof(OrbitMeshletDispatch));
    const size_t o_ebs = carve((ent / 256 + 1) * 4), o_et = carve(256);
    const size_t o_mtc = carve(mtiles * 4), o_mtb = carve(mtiles * 4), o_mt = carve(256);
    const size_t o_ftf = carve(((size_t)md32 / 4 + 2) * 4); // one flag per tile of 4 records (cull_fused.hip)
    const size_t o_split = carve(rec_shift > 5u ? ORBIT_DISPATCH_HEADER + 16 * (size_t)md32 : 0); // dispatch_size 64 / 128
    const size_t o_xbp = carve((size_t)kExpandBlocks * 4);
    const size_t o_mtm = carve(mtiles * 64);
    const size_t o_mtp = carve(mtiles * 128 * sizeof(Payload));
    const size_t mchunks = mtiles / kScanChunk + 2;
    const size_t o_mcs = carve(mchunks * 4);
    const size_t o_cc = carve((2 * cchunks + 64) * 4);
    const size_t o_avl = carve((size_t)caps.max_lights * 16 + 1024);
    const size_t o_alf = carve(((size_t)caps.max_lights / 32 + 64) * 4);
    const size_t o_acn = carve(((size_t)caps.max_clusters + kAssignPad) * 4);
    // assign: block_sums holds the sums of the chunks' counts (a line each); block_base the list of heavy blocks
    const size_t o_abs = carve((((size_t)caps.max_clusters / kScanChunk + 2) * kChunkSumStride + 64) * 4);
    const size_t o_abb = carve(((size_t)caps.max_clusters + kAssignPad) * 4), o_at = carve(256);
    // light assignment: cached cluster AABBs, and per group of 256 active clusters the coarse candidate lists
    const size_t agroups = (size_t)caps.max_clusters / 256 + 1;
    const size_t aseg = (((size_t)caps.max_lights + 15) / 16 + 63) / 64 * 64;
    const size_t o_aab = carve(((size_t)caps.max_clusters + kAssignPad) * 6 * 4);
    const size_t o_aco = carve(agroups * 16 * aseg * 4), o_acl = carve(agroups * 16 * aseg * 16), o_acc = carve(agroups * 16 * 4), o_agb = carve(agroups * 32), o_ago = carve((agroups + 1) * 4);
    const size_t o_ahc = carve(((size_t)caps.max_clusters + kAssignPad) * kHitCache * 4);
    ctx->arena_bytes = off;
    // (Where the arena lands physically was suspected of the Meshlet-buffer evaluation's two speeds, 311 against 333-343 us
    // by process at BASELINE config 5.  Round 6 allocated four candidate arenas here and timed a probe of that
    // evaluation's write pattern against each: the probe does not tell them apart — 14.3-14.9 us on all — and engines on
    // probed and unprobed arenas ran the same 328-353 us; five engines alive on one buffer 348-352: the spread is between
    // boxes and processes, not between arenas of a process.  Removed; profiles/r06_notes.md.)
    e = hipMalloc((void **)&ctx->arena, ctx->arena_bytes);
    if (e != hipSuccess) {
        delete ctx;
        return hip_fail(nullptr, e, "hipMalloc(scratch arena)");
    }
    e = memset_now(ctx->arena, 0, ctx->arena_bytes);
    if (e != hipSuccess) {
        (void)hipFree(ctx->arena);
        delete ctx;
        return hip_fail(nullptr, e, "hipMemset(scratch arena)");
    }
    uint8_t *a = ctx->arena;
    ctx->status = (int32_t *)(a + o_status);
    ctx->debug_cycles = (unsigned long long *)(a + o_dbg);
    ctx->zero_page = (void *)(a + o_zero);
    ctx->e_proto = (OrbitMeshletDispatch *)(a + o_eproto);
    ctx->e_block_sums = (uint32_t *)(a + o_ebs);
    ctx->e_total = (uint32_t *)(a + o_et);
    ctx->m_tile_counts = (uint32_t *)(a + o_mtc);
    ctx->m_tile_base = (uint32_t *)(a + o_mtb);
    ctx->x_block_pop = (uint32_t *)(a + o_xbp);
    ctx->m_total = (uint32_t *)(a + o_mt);
    ctx->m_tile_masks = (uint32_t *)(a + o_mtm);
    ctx->m_tile_payload = (Payload *)(a + o_mtp);
    ctx->m_chunk_sums = (uint32_t *)(a + o_mcs);
    ctx->c_chunk = (uint32_t *)(a + o_cc);
    ctx->c_chunk_words = cchunks;
    ctx->a_view_lights = (float4 *)(a + o_avl);
    ctx->a_light_flags = (uint32_t *)(a + o_alf);
    ctx->a_counts = (uint32_t *)(a + o_acn);
    ctx->a_block_sums = (uint32_t *)(a + o_abs);
    ctx->a_block_base = (uint32_t *)(a + o_abb);
    ctx->a_total = (uint32_t *)(a + o_at);
    ctx->a_aabb = (float *)(a + o_aab);
    ctx->a_coarse = (uint32_t *)(a + o_aco);
    ctx->a_coarse_lights = (float4 *)(a + o_acl);
    ctx->a_hit_cache = (uint32_t *)(a + o_ahc);
    ctx->a_coarse_counts = (uint32_t *)(a + o_acc);
    ctx->a_group_box = (float *)(a + o_agb);
    ctx->a_group_order = (uint32_t *)(a + o_ago);
    ctx->a_coarse_seg = (uint32_t)aseg;
    ctx->g_counts = (uint32_t *)(a + o_gc);
    ctx->m_tickets = (uint32_t *)(a + o_tk);
    ctx->m_list_sync = (uint32_t *)(a + o_ls);
    ctx->f_done = (uint32_t *)(a + o_sd);
    ctx->f_sync = (uint32_t *)(a + o_fsy);
    ctx->d_tickets = (uint32_t *)(a + o_dtk);
    ctx->f_ent_flags = (uint32_t *)(a + o_fef);
    ctx->f_tile_flags = (uint32_t *)(a + o_ftf);
    ctx->m_split = rec_shift > 5u ? a + o_split : nullptr;
    ctx->rec_shift = rec_shift;
#ifdef ORBIT_TRIAGE // perf-triage builds (tools/mkvariant_any.sh); the product library reads no environment variable
    const char *dbg = getenv("ORBIT_SP_DEBUG");
    ctx->debug_flags = dbg ? (uint32_t)atoi(dbg) : 0u;
#endif
    snprintf(ctx->err, sizeof(ctx->err), "no error");
    // scan scratch of views 1.. of orbit_cull_views, when the caller announced them
    if (caps.cull_path > 2u) {
        (void)orbit_ctx_destroy(ctx);
        return fail(nullptr, ORBIT_E_INVALID, "caps.cull_path %u (0 auto, 1 launch chain, 2 one-launch cull)", caps.cull_path);
    }
    if (caps.max_views > ORBIT_MAX_CULL_VIEWS) {
        (void)orbit_ctx_destroy(ctx);
        return fail(nullptr, ORBIT_E_CAPACITY, "caps.max_views %u > %u", caps.max_views, (unsigned)ORBIT_MAX_CULL_VIEWS);
    }
    OrbitCaps child_caps = caps; // (a view's scratch: the culls' only — no cluster or light scratch, the bulk of a context's)
    child_caps.max_views = 0, child_caps.max_clusters = 0, child_caps.max_lights = 0;
    for (uint32_t v = 1; v < caps.max_views; v++) {
        OrbitCtx *child = nullptr;
        const int32_t rc = orbit_ctx_create(device_id, &child_caps, &child);
        if (rc != ORBIT_OK) {
            (void)orbit_ctx_destroy(ctx);
            return rc;
        }
        ctx->view_ctx.push_back(child);
    }
    *out_ctx = ctx;
    return ORBIT_OK;
}

int32_t orbit_ctx_destroy(OrbitCtx *ctx) {
    if (!ctx) return ORBIT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (hipEvent_t ev : ctx->prof_events) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->side_event)
        if (ev) (void)hipEventDestroy(ev);
    for (hipStream_t st : ctx->side_stream)
        if (st) (void)hipStreamDestroy(st);
    for (OrbitCtx *child : ctx->view_ctx) (void)orbit_ctx_destroy(child);
    if (ctx->meshlet_stream) ctx->meshlet_stream->bindings.fetch_sub(1);
    if (ctx->arena) (void)hipFree(ctx->arena);
    delete ctx;
    return ORBIT_OK;
}

int32_t orbit_ctx_status(OrbitCtx *ctx, void *stream, int32_t sync) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    hipStream_t s = (hipStream_t)stream;
    if (sync) {
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) return hip_fail(ctx, e, "hipStreamSynchronize");
    }
    int32_t v = 0;
    hipError_t e = hipMemcpy(&v, ctx->status, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipMemcpy(status)");
    for (OrbitCtx *child : ctx->view_ctx) { // latches of the views that ran on the children's scratch
        int32_t cv = 0;
        e = hipMemcpy(&cv, child->status, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return hip_fail(ctx, e, "hipMemcpy(status)");
        if (cv != 0) {
            (void)memset_now(child->status, 0, 4);
            if (v == 0) v = cv;
        }
    }
    if (v != 0) {
        e = memset_now(ctx->status, 0, 4);
        if (e != hipSuccess) return hip_fail(ctx, e, "hipMemset(status)");
        return fail(ctx, v, v == ORBIT_E_CAPACITY ? "an append overflowed a caller buffer (entries dropped)"
                            : v == ORBIT_E_RANGE  ? "a meshlet outside the bound meshlet stream was culled or expanded"
                            : v == ORBIT_E_STALE  ? "the bound meshlet stream no longer mirrors its meshlet buffer (update missing)"
                                                  : "device-latched error %d", v);
    }
    return ORBIT_OK;
}

// ------------------------------------------------------------- measurement hooks
int32_t orbit_ctx_profile(OrbitCtx *ctx, int32_t enable) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->profiling = enable > 0 ? (uint32_t)enable : 0u;
    ctx->prof_used = 0;
    ctx->prof_calls = 0;
    return ORBIT_OK;
}

// The hook's event pairs are created on first use (hipEventCreate + an event's first record: tens of microseconds of
// host time each) — a caller that times a region creates them BEFORE it: `pairs` pairs exist afterwards, each recorded
// once on `stream`.
int32_t orbit_ctx_profile_reserve(OrbitCtx *ctx, uint32_t pairs, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (pairs > 65536u) return fail(ctx, ORBIT_E_CAPACITY, "profile_reserve: %u pairs", pairs);
    while (ctx->prof_events.size() < 2u * (size_t)pairs) {
        hipEvent_t ev;
        hipError_t e = hipEventCreate(&ev);
        if (e != hipSuccess) return hip_fail(ctx, e, "hipEventCreate");
        ctx->prof_events.push_back(ev);
        e = hipEventRecord(ev, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(ctx, e, "hipEventRecord");
    }
    return ORBIT_OK;
}

int32_t orbit_ctx_profile_read(OrbitCtx *ctx, float *avg_ms, uint32_t *launches) {
    if (!ctx || !avg_ms || !launches) return fail(ctx, ORBIT_E_INVALID, "profile_read: NULL argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    double sum = 0.0;
    uint32_t n = 0;
    for (size_t i = 0; i + 1 < ctx->prof_used; i += 2) {
        hipError_t e = hipEventSynchronize(ctx->prof_events[i + 1]);
        if (e != hipSuccess) return hip_fail(ctx, e, "hipEventSynchronize");
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, ctx->prof_events[i], ctx->prof_events[i + 1]);
        if (e != hipSuccess) return hip_fail(ctx, e, "hipEventElapsedTime");
        sum += ms;
        n++;
    }
    *avg_ms = n ? (float)(sum / n) : 0.f;
    *launches = n;
    ctx->prof_used = 0;
    return ORBIT_OK;
}

// perf triage only (not part of the public ABI): per-wave {begin, end} wall_clock64 stamps of the last phase-stamped
// meshlet_eval launch (ORBIT_SP_DEBUG=8), 100 MHz ticks; out holds 2 * 8192 values
int32_t orbit_debug_read_wave_stamps(OrbitCtx *ctx, unsigned long long *out) {
    if (!ctx || !out) return ORBIT_E_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return ORBIT_E_HIP;
    if (hipMemcpy(out, ctx->debug_cycles + 32, 16 * 8192, hipMemcpyDeviceToHost) != hipSuccess) return ORBIT_E_HIP;
    return ORBIT_OK;
}

// triage only (not part of the public ABI): the progress marks of a -DORBIT_FUSED_DEBUG build of cull_fused.hip, copied
// on a stream of their own so that they can be read while a launch is still running; out holds 8192 values
int32_t orbit_debug_read_fused_marks(OrbitCtx *ctx, unsigned long long *out) {
    if (!ctx || !out) return ORBIT_E_INVALID;
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return ORBIT_E_HIP;
    hipError_t e = hipMemcpyAsync(out, ctx->debug_cycles + 64, 8192 * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
    return e == hipSuccess ? ORBIT_OK : ORBIT_E_HIP;
}

// tests only (not part of the public ABI): how many polls a workgroup of the chain emit waits for a chunk's sum before
// it scans the chunk itself (meshlet_emit.hip emit_scan_wait; default 256).  0 makes every waiting workgroup do so at
// once — the path that otherwise only runs when the launch's first workgroups are not being dispatched.
int32_t orbit_debug_set_scan_patience(OrbitCtx *ctx, uint32_t polls) {
    if (!ctx) return ORBIT_E_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->scan_patience = polls;
    return ORBIT_OK;
}

// tests only (not part of the public ABI): cluster_mark's z slice through the hardware log2 with its guard band
// (orbit_device.h depth_slice) against the canonical software form, for every float bit pattern in [lo_bits, hi_bits]:
// out = {mismatches, samples decided by the canonical path, bits of the largest |v_log_f32 - log2c| seen}
int32_t orbit_debug_log2_guard(OrbitCtx *ctx, uint32_t lo_bits, uint32_t hi_bits, float z_scale, float z_bias,
                               unsigned long long out[3]) {
    if (!ctx || !out || hi_bits < lo_bits) return ORBIT_E_INVALID;
    std::lock_guard<std::mutex> lock(ctx->mu);
    unsigned long long *d = ctx->debug_cycles + 8;
    if (hipMemset(d, 0, 24) != hipSuccess) return ORBIT_E_HIP;
    if (launch_log2_guard_check(lo_bits, hi_bits, z_scale, z_bias, d, nullptr) != hipSuccess) return ORBIT_E_HIP;
    if (hipDeviceSynchronize() != hipSuccess) return ORBIT_E_HIP;
    if (hipMemcpy(out, d, 24, hipMemcpyDeviceToHost) != hipSuccess) return ORBIT_E_HIP;
    if (hipMemset(d, 0, 24) != hipSuccess) return ORBIT_E_HIP;
    return ORBIT_OK;
}

// perf triage only (not part of the public ABI): reads and clears the per-phase cycle sums
int32_t orbit_debug_read_cycles(OrbitCtx *ctx, unsigned long long out[8]) {
    if (!ctx || !out) return ORBIT_E_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return ORBIT_E_HIP;
    if (hipMemcpy(out, ctx->debug_cycles, 64, hipMemcpyDeviceToHost) != hipSuccess) return ORBIT_E_HIP;
    if (hipMemset(ctx->debug_cycles, 0, 64) != hipSuccess) return ORBIT_E_HIP;
    return ORBIT_OK;
}

// perf triage only (not part of the public ABI): the seven wall-clock stamps a -DORBIT_TRIAGE_STAMPS build of
// depth_reduce.hip leaves in the unused words of pyramid 0's arrival counters (tools/depth_stamps.py)
int32_t orbit_debug_read_depth_stamps(OrbitCtx *ctx, unsigned long long out[7]) {
    if (!ctx || !out) return ORBIT_E_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return ORBIT_E_HIP;
    if (hipMemcpy(out, ctx->d_tickets + 2, 56, hipMemcpyDeviceToHost) != hipSuccess) return ORBIT_E_HIP;
    return ORBIT_OK;
}

// perf triage only (not part of the public ABI; all zeros unless the library was built with -DORBIT_TRIAGE): reads and
// clears the executed sphere-box test counts of the cluster assignment — coarse tests, coarse passes, count launch
// {filter, cluster} tests, write launch {filter, cluster} tests, write-launch blocks served from the hit cache, the
// largest candidate list of a group
// (the count launch's per-block {begin, end} wall-clock stamps land where orbit_debug_read_wave_stamps reads)
int32_t orbit_debug_read_cluster_tests(OrbitCtx *ctx, unsigned long long out[8]) {
    if (!ctx || !out) return ORBIT_E_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return ORBIT_E_HIP;
    if (hipMemcpy(out, ctx->debug_cycles + 16, 64, hipMemcpyDeviceToHost) != hipSuccess) return ORBIT_E_HIP;
    if (hipMemset(ctx->debug_cycles + 16, 0, 64) != hipSuccess) return ORBIT_E_HIP;
    return ORBIT_OK;
}

// ------------------------------------------------------------------ entity_cull
// Validates one entity cull and fills its parameter block; `scratch` owns the scan scratch the launch will use
// (`ctx` itself, or one of its view children), errors are reported on `ctx`.  Caller holds ctx->mu.
static int32_t entity_cull_params(OrbitCtx *ctx, OrbitCtx *scratch, const OrbitGpuCullInfo *ci,
                                  const OrbitEntityCullBufs *b, uint32_t draw_first, uint32_t entity_draw_count,
                                  bool exact_range, EntityCullParams &p) {
    int32_t rc = check_cull_info(ctx, ci);
    if (rc) return rc;
    if (ctx->rec_shift > 5u && ci->occlusion_pass != 0)
        return fail(ctx, ORBIT_E_INVALID,
                    "occlusion_pass %u with dispatch_size %u: the reference's visibility words are consistent for 32 only — an "
                    "entity gets ceil(meshlets / 32) words (src/scene.rs:427) while entity_cull.comp:222 advances a record's "
                    "word offset by meshlet_count / S; other dispatch sizes serve occlusion pass 0",
                    ci->occlusion_pass, ctx->caps.dispatch_size);
    if (!b) return fail(ctx, ORBIT_E_INVALID, "bufs is NULL");
    if (!b->entity_draw_buffer || !b->mesh_info_buffer || !b->meshlet_dispatch_buffer || !b->entity_buffer)
        return fail(ctx, ORBIT_E_MISSING, "entity_cull: a required buffer is NULL");
    if (ci->occlusion_pass != 0 && !b->visibility_buffer)
        return fail(ctx, ORBIT_E_MISSING, "occlusion_pass %u needs visibility_buffer", ci->occlusion_pass);
    if (ci->occlusion_pass == 2 &&
        ((!b->depth_pyramid && !b->depth_pyramid_levels) || !b->depth_pyramid_size[0] || !b->depth_pyramid_size[1]))
        return fail(ctx, ORBIT_E_MISSING, "occlusion_pass 2 needs depth_pyramid");
    if (draw_first % 32u) return fail(ctx, ORBIT_E_INVALID, "draw_first %u is not a multiple of 32", draw_first);
    if (entity_draw_count > ctx->caps.max_entities)
        return fail(ctx, ORBIT_E_CAPACITY, "entity_draw_count %u > caps.max_entities %u", entity_draw_count,
                    ctx->caps.max_entities);
    p.ci = *ci;
    p.entity_draw_buffer = (const uint8_t *)b->entity_draw_buffer;
    p.mesh_infos = (const OrbitMeshInfo *)b->mesh_info_buffer;
    p.mesh_side = mesh_side_for(ctx->meshlet_stream, b->mesh_info_buffer);
    p.dispatch_buffer = (uint8_t *)b->meshlet_dispatch_buffer;
    p.entities = (const OrbitEntityData *)b->entity_buffer;
    p.visibility = b->visibility_buffer;
    p.pyr = make_pyramid_view(b->depth_pyramid, b->depth_pyramid_size, b->depth_pyramid_levels);
    p.draw_first = draw_first;
    // whole-buffer call: the reference's grid of ceil(n/256) x 256 invocations (draw_gen.rs:377);
    // range call: exactly [draw_first, draw_first + n) so that shards never overlap
    const uint64_t limit = exact_range ? (uint64_t)draw_first + entity_draw_count
                                       : ((uint64_t)entity_draw_count + 255u) / 256u * 256u;
    p.draw_limit = limit > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)limit;
    p.dispatch_capacity = b->dispatch_capacity;
    p.ne_chunks = (entity_draw_count + 255u) / 256u;
    p.rec_shift = ctx->rec_shift;
    p.arith = ctx->caps.arith_profile;
    p.proto = scratch->e_proto;
    p.block_sums = scratch->e_block_sums;
    p.sync = scratch->f_sync;
    p.ent_flags = scratch->f_ent_flags;
    p.total = scratch->e_total;
    p.status = scratch->status;
    return ORBIT_OK;
}

static int32_t entity_cull_impl(OrbitCtx *ctx, const OrbitGpuCullInfo *ci, const OrbitEntityCullBufs *b,
                                uint32_t draw_first, uint32_t entity_draw_count, bool exact_range, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    EntityCullParams p;
    const int32_t rc = entity_cull_params(ctx, ctx, ci, b, draw_first, entity_draw_count, exact_range, p);
    if (rc) return rc;
    if (p.mesh_side.table && ctx->caps.validate_streams) { // debug: is the mesh side table still a mirror?
        const hipError_t ve = launch_mesh_side_validate(p.mesh_infos, 0u, p.mesh_side.count, p.mesh_side.table, ctx->status,
                                                        (hipStream_t)stream);
        if (ve != hipSuccess) return hip_fail(ctx, ve, "launch mesh_side_validate");
    }
    const hipError_t e = launch_entity_cull(p, entity_draw_count, ctx->num_cus, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch entity_cull");
    return ORBIT_OK;
}

int32_t orbit_entity_cull(OrbitCtx *ctx, const OrbitGpuCullInfo *cull_info, const OrbitEntityCullBufs *bufs,
                          uint32_t entity_draw_count, void *stream) {
    return entity_cull_impl(ctx, cull_info, bufs, 0u, entity_draw_count, false, stream);
}

int32_t orbit_entity_cull_range(OrbitCtx *ctx, const OrbitGpuCullInfo *cull_info, const OrbitEntityCullBufs *bufs,
                                uint32_t draw_first, uint32_t draw_count, void *stream) {
    return entity_cull_impl(ctx, cull_info, bufs, draw_first, draw_count, true, stream);
}

// ----------------------------------------------------------------- meshlet_cull
} // extern "C"

namespace {

// orbit_meshlet_cull (task_records == nullptr: survivors -> MeshletDrawCommandBuffer) and
// orbit_meshlet_task_cull (survivors stay with their record: one OrbitMeshTaskRecord each).
// Validation + parameter block of one meshlet cull (see entity_cull_params).  Caller holds ctx->mu.
int32_t meshlet_cull_params(OrbitCtx *ctx, OrbitCtx *scratch, const OrbitGpuCullInfo *ci, const OrbitMeshletCullBufs *b,
                            OrbitMeshTaskRecord *task_records, bool task_path, void *visible_buffer,
                            uint32_t visible_capacity, uint32_t visible_path /* 0: commands, 2: records */, MeshletCullParams &p) {
    int32_t rc = check_cull_info(ctx, ci);
    if (rc) return rc;
    if (!b) return fail(ctx, ORBIT_E_INVALID, "bufs is NULL");
    if (!b->meshlet_dispatch_buffer || !b->meshlet_buffer || !b->entity_buffer || !b->material_buffer ||
        (task_path ? (void *)task_records : visible_path ? visible_buffer : b->draw_commands_buffer) == nullptr)
        return fail(ctx, ORBIT_E_MISSING, "meshlet_cull: a required buffer is NULL");
    if (ctx->rec_shift > 5u) { // dispatch_size 64 / 128: pass 0 into a MeshletDrawCommandBuffer, nothing else
        if (ci->occlusion_pass != 0)
            return fail(ctx, ORBIT_E_INVALID,
                        "occlusion_pass %u with dispatch_size %u: the reference's visibility words are consistent for 32 only "
                        "(src/scene.rs:427 against meshlet_cull.comp:129-134, 233-254); other dispatch sizes serve pass 0",
                        ci->occlusion_pass, ctx->caps.dispatch_size);
        if (task_path)
            return fail(ctx, ORBIT_E_INVALID,
                        "the mesh-shading path with dispatch_size %u: MeshTaskPayload.meshlet_indices has 32 entries whatever "
                        "the dispatch size (shaders/include/types.glsl:196-200) — a record with more survivors writes past it",
                        ctx->caps.dispatch_size);
        if (visible_path)
            return fail(ctx, ORBIT_E_INVALID, "the sharded engine's record list is defined for dispatch_size 32 (one 32-bit "
                                              "ballot per record); this context has %u", ctx->caps.dispatch_size);
    }
    const bool meshlet_occ = ci->meshlet_visibility_buffer != ORBIT_NONE;
    if (meshlet_occ && ci->occlusion_pass != 0 && !b->meshlet_visibility_buffer)
        return fail(ctx, ORBIT_E_MISSING, "cull_info declares a meshlet visibility buffer but the pointer is NULL");
    if (meshlet_occ && ci->occlusion_pass == 2 &&
        ((!b->depth_pyramid && !b->depth_pyramid_levels) || !b->depth_pyramid_size[0] || !b->depth_pyramid_size[1]))
        return fail(ctx, ORBIT_E_MISSING, "occlusion_pass 2 needs depth_pyramid");
    if (b->dispatch_capacity > ctx->caps.max_dispatches)
        return fail(ctx, ORBIT_E_CAPACITY, "dispatch_capacity %u > caps.max_dispatches %u", b->dispatch_capacity,
                    ctx->caps.max_dispatches);
    p.ci = *ci;
    p.dispatch_buffer = (const uint8_t *)b->meshlet_dispatch_buffer;
    p.meshlets = (const OrbitMeshlet *)b->meshlet_buffer;
    p.ms = stream_view_for(ctx->meshlet_stream, b->meshlet_buffer, b->material_buffer);
    p.draw_buffer = (uint8_t *)(visible_path ? visible_buffer : b->draw_commands_buffer);
    p.entities = (const OrbitEntityData *)b->entity_buffer;
    p.materials = (const OrbitMaterialData *)b->material_buffer;
    p.meshlet_visibility = b->meshlet_visibility_buffer;
    p.pyr = make_pyramid_view(b->depth_pyramid, b->depth_pyramid_size, b->depth_pyramid_levels);
    p.dispatch_capacity = b->dispatch_capacity;
    if (ctx->rec_shift > 5u) { // the launches read the records of 32 that split_params_for's launch derives
        p.dispatch_buffer = scratch->m_split;
        p.dispatch_capacity = b->dispatch_capacity << (ctx->rec_shift - 5u);
    }
    p.draw_capacity = visible_path ? visible_capacity : b->draw_capacity;
    p.visible_list = visible_path;
    p.arith = ctx->caps.arith_profile;
    p.material_count = b->material_count;
    {   // the symmetric five-plane frustum, recognised bit for bit (kernels.h MeshletCullParams::std_planes)
        auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
        auto zero = [&](float f) { return (bits(f) & 0x7FFFFFFFu) == 0u; };
        auto fin = [&](float f) { return (bits(f) & 0x7F800000u) != 0x7F800000u; };
        const float(*pl)[4] = ci->cull_planes;
        bool ok = ci->cull_plane_count == 5;
        ok = ok && zero(pl[0][1]) && zero(pl[0][3]) && zero(pl[1][1]) && zero(pl[1][3]) &&
             bits(pl[1][0]) == (bits(pl[0][0]) ^ 0x80000000u) && bits(pl[1][2]) == bits(pl[0][2]);
        ok = ok && zero(pl[2][0]) && zero(pl[2][3]) && zero(pl[3][0]) && zero(pl[3][3]) &&
             bits(pl[3][1]) == (bits(pl[2][1]) ^ 0x80000000u) && bits(pl[3][2]) == bits(pl[2][2]);
        ok = ok && zero(pl[4][0]) && zero(pl[4][1]);
        const float v[6] = {pl[0][0], pl[0][2], pl[2][1], pl[2][2], pl[4][2], pl[4][3]};
        for (float f : v) ok = ok && fin(f);
        p.std_planes = ok ? 1u : 0u;
        for (int k = 0; k < 6; k++) p.stdp[k] = ok ? v[k] : 0.0f;
    }
    p.task_records = task_records;
    p.also_commands = nullptr;
    p.also_commands_capacity = 0u;
    p.tile_counts = scratch->m_tile_counts;
    p.tile_masks = scratch->m_tile_masks;
    p.tile_payload = scratch->m_tile_payload;
    p.tile_base = scratch->m_tile_base;
    p.chunk_sums = scratch->m_chunk_sums;
    p.total = scratch->m_total;
    p.tickets = scratch->m_tickets;
    p.list_sync = scratch->m_list_sync;
    p.debug_flags = scratch->debug_flags;
    p.debug_cycles = scratch->debug_cycles;
    p.scan_patience = ctx->scan_patience;
    p.zero_page = scratch->zero_page;
    p.status = scratch->status;
    return ORBIT_OK;
}

// dispatch_size 64 / 128: the launch that turns the caller's S-sized records into the records of 32 the meshlet stage
// reads (entity_cull.hip split_records_body), on `scratch`'s buffer.
SplitRecordsParams split_params_for(const OrbitCtx *ctx, const OrbitCtx *scratch, const OrbitMeshletCullBufs *b) {
    SplitRecordsParams sp;
    sp.src = (const uint8_t *)b->meshlet_dispatch_buffer;
    sp.dst = scratch->m_split;
    sp.src_capacity = b->dispatch_capacity;
    sp.rec_shift = ctx->rec_shift;
    return sp;
}

int32_t meshlet_cull_impl(OrbitCtx *ctx, const OrbitGpuCullInfo *ci, const OrbitMeshletCullBufs *b,
                          OrbitMeshTaskRecord *task_records, bool task_path, void *stream,
                          void *visible_buffer = nullptr, uint32_t visible_capacity = 0, uint32_t visible_path = 0u,
                          bool also_commands = false) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    MeshletCullParams p;
    const int32_t rc = meshlet_cull_params(ctx, ctx, ci, b, task_records, task_path, visible_buffer, visible_capacity,
                                           visible_path, p);
    if (rc) return rc;
    if (also_commands) { // the record list AND the commands of the same evaluation (sharded engine)
        if (!b->draw_commands_buffer) return fail(ctx, ORBIT_E_MISSING, "records_and_commands: draw_commands_buffer is NULL");
        p.also_commands = (uint8_t *)b->draw_commands_buffer;
        p.also_commands_capacity = b->draw_capacity;
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ctx->profiling && (ctx->prof_calls++ % ctx->profiling) == 0) {
        if (ctx->prof_used + 2 > ctx->prof_events.size()) {
            for (int i = 0; i < 2; i++) {
                hipEvent_t ev;
                if (hipEventCreate(&ev) != hipSuccess) return fail(ctx, ORBIT_E_HIP, "hipEventCreate");
                ctx->prof_events.push_back(ev);
            }
        }
        ev0 = ctx->prof_events[ctx->prof_used];
        ev1 = ctx->prof_events[ctx->prof_used + 1];
        ctx->prof_used += 2;
    }
    if (p.ms.sphere && ctx->caps.validate_streams && ci->occlusion_pass != 1) { // debug: is the stream still a mirror?
        const hipError_t ve = launch_meshlet_stream_validate(p.meshlets, p.ms, p.ms.cls0 ? p.materials : nullptr,
                                                             ctx->meshlet_stream->material_count, ctx->status,
                                                             (hipStream_t)stream);
        if (ve != hipSuccess) return hip_fail(ctx, ve, "launch meshlet_stream_validate");
    }
    hipError_t e = hipSuccess;
    if (ctx->rec_shift > 5u) e = launch_split_records(split_params_for(ctx, ctx, b), (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch split_records");
    e = launch_meshlet_cull(p, ctx->num_cus, (hipStream_t)stream, ev0, ev1);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch meshlet_cull");
    if (p.ms.sphere && ci->occlusion_pass != 1) {
        ctx->stream_culls++;
        if (p.ms.cls0) ctx->class_culls++;
    }
    return ORBIT_OK;
}

} // namespace

extern "C" {

int32_t orbit_meshlet_cull(OrbitCtx *ctx, const OrbitGpuCullInfo *ci, const OrbitMeshletCullBufs *b, void *stream) {
    return meshlet_cull_impl(ctx, ci, b, nullptr, false, stream);
}

} // extern "C"

namespace {

// One orbit_cull_views call, validated and laid out, not yet enqueued (orbit_frame_late validates all its groups before
// it forks anything).  Scratch: view i runs on scratch set `scratch_base + i` (0 = the context itself, k = child k - 1).
struct PreparedCullViews {
    EntityCullViews ev{};
    MeshletCullViews mv{};
    FusedCullViews fv{};
    SplitRecordsViews sv{}; // dispatch_size 64 / 128: entry k serves mv.v[k]
    uint32_t draws[ORBIT_MAX_CULL_VIEWS] = {};
    uint32_t count = 0, n_mesh = 0, max_draws = 0;
    bool fused = false;
};

OrbitCtx *scratch_of(OrbitCtx *ctx, uint32_t k) { return k == 0 ? ctx : ctx->view_ctx[k - 1]; }

// Caller holds ctx->mu.  Allocates missing scratch sets (the only allocation an enqueue call can ever make).
int32_t prepare_cull_views(OrbitCtx *ctx, const OrbitCullView *views, uint32_t count, uint32_t scratch_base,
                           PreparedCullViews &pc) {
    if (!views || count == 0) return fail(ctx, ORBIT_E_MISSING, "cull_views: no views");
    if (count > ORBIT_MAX_CULL_VIEWS || scratch_base + count > ORBIT_MAX_CULL_VIEWS)
        return fail(ctx, ORBIT_E_CAPACITY, "cull_views: %u views (max %u)", scratch_base + count, (unsigned)ORBIT_MAX_CULL_VIEWS);
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipSetDevice");
    // scan scratch per view (views 1.. on child contexts), created once
    while (ctx->view_ctx.size() + 1 < scratch_base + count) {
        OrbitCtx *child = nullptr;
        OrbitCaps child_caps = ctx->caps;
        child_caps.max_views = 0, child_caps.max_clusters = 0, child_caps.max_lights = 0;
        const int32_t rc = orbit_ctx_create(ctx->device, &child_caps, &child);
        if (rc != ORBIT_OK) return fail(ctx, rc, "cull_views: scratch for view %zu: %s", ctx->view_ctx.size() + 1, g_err);
        ctx->view_ctx.push_back(child);
    }
    // every view's parameter block, validated before anything is enqueued; the blocks travel by value in the
    // kernels' argument segments (a few KB)
    pc.count = count;
    for (uint32_t i = 0; i < count; i++) {
        const OrbitCullView &v = views[i];
        OrbitCtx *scratch = scratch_of(ctx, scratch_base + i);
        int32_t rc = entity_cull_params(ctx, scratch, v.cull_info, &v.entity, 0u, v.entity_draw_count, false, pc.ev.v[i]);
        if (rc == ORBIT_OK && !v.skip_meshlet_stage) {
            rc = meshlet_cull_params(ctx, scratch, v.cull_info, &v.meshlet, nullptr, false, nullptr, 0, false, pc.mv.v[pc.n_mesh]);
            if (rc == ORBIT_OK && ctx->rec_shift > 5u) pc.sv.v[pc.n_mesh] = split_params_for(ctx, scratch, &v.meshlet);
            if (rc == ORBIT_OK && v.meshlet.meshlet_dispatch_buffer != v.entity.meshlet_dispatch_buffer)
                rc = fail(ctx, ORBIT_E_INVALID, "view %u: the meshlet stage must read the entity stage's dispatch buffer", i);
            pc.n_mesh++;
        }
        if (rc != ORBIT_OK) return rc;
        pc.max_draws = v.entity_draw_count > pc.max_draws ? v.entity_draw_count : pc.max_draws;
    }
    // Views of the reference's own size (src/scene.rs:303, assets/mod.rs:202) are bound by launch latency, not by
    // bytes: all of them together as ONE launch per (pass, projection) (cull_fused.hip) — same outputs.  It evaluates
    // the 32-B Meshlet buffer whatever stream is bound (the survivors' command words are then in the row registers).
    // (a context that validates its stream on every cull keeps the chain: the one launch never reads the stream, and
    // the validation and ORBIT_E_RANGE are what such a context exists for)
    // (and a context of another dispatch size: the one launch hands 32-meshlet records over inside itself)
    bool fused = ctx->caps.cull_path != 1u && pc.n_mesh == count && ctx->rec_shift == 5u &&
                 !(ctx->caps.validate_streams != 0u && ctx->meshlet_stream != nullptr);
    for (uint32_t i = 0; i < count && fused; i++)
        fused = views[i].entity_draw_count != 0u &&
                (ctx->caps.cull_path == 2u || views[i].entity_draw_count <= kFusedMaxEntityDraws);
    pc.fused = fused;
    if (fused) {
        for (uint32_t i = 0; i < count; i++) {
            OrbitCtx *scratch = scratch_of(ctx, scratch_base + i);
            FusedCullParams &f = pc.fv.v[i];
            f.m = pc.mv.v[i];
            f.m.ms = MeshletStreamView{};
            f.entity_draw_buffer = pc.ev.v[i].entity_draw_buffer;
            f.mesh_infos = pc.ev.v[i].mesh_infos;
            f.mesh_side = pc.ev.v[i].mesh_side;
            f.visibility = pc.ev.v[i].visibility;
            f.e_pyr = pc.ev.v[i].pyr;
            f.draw_first = pc.ev.v[i].draw_first;
            f.draw_limit = pc.ev.v[i].draw_limit;
            f.e_dispatch_capacity = pc.ev.v[i].dispatch_capacity;
            f.ne_chunks = (views[i].entity_draw_count + 255u) / 256u;
            f.e_total = pc.ev.v[i].total;
            f.sync = scratch->f_sync;
            f.ent_flags = scratch->f_ent_flags;
            f.tile_flags = scratch->f_tile_flags;
            pc.draws[i] = views[i].entity_draw_count;
        }
    }
    return ORBIT_OK;
}

int32_t launch_prepared_cull_views(OrbitCtx *ctx, const PreparedCullViews &pc, hipStream_t s) {
    hipError_t e;
    if (pc.fused) {
        e = launch_cull_fused_views(pc.fv, pc.draws, pc.count, ctx->num_cus, s);
        if (e != hipSuccess) return hip_fail(ctx, e, "launch cull_fused");
        ctx->fused_culls += pc.count;
        return ORBIT_OK;
    }
    e = launch_entity_cull_views(pc.ev, pc.count, pc.max_draws, ctx->num_cus, s);
    if (e == hipSuccess && ctx->rec_shift > 5u && pc.n_mesh) e = launch_split_records_views(pc.sv, pc.n_mesh, s);
    if (e == hipSuccess) e = launch_meshlet_cull_views(pc.mv, pc.n_mesh, ctx->num_cus, s);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch cull_views");
    for (uint32_t i = 0; i < pc.n_mesh; i++)
        if (pc.mv.v[i].ms.sphere && pc.mv.v[i].ci.occlusion_pass != 1) {
            ctx->stream_culls++;
            if (pc.mv.v[i].ms.cls0) ctx->class_culls++;
        }
    return ORBIT_OK;
}

} // namespace

extern "C" {

int32_t orbit_cull_views(OrbitCtx *ctx, const OrbitCullView *views, uint32_t count, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    PreparedCullViews pc;
    const int32_t rc = prepare_cull_views(ctx, views, count, 0u, pc);
    if (rc != ORBIT_OK) return rc;
    return launch_prepared_cull_views(ctx, pc, (hipStream_t)stream);
}

int32_t orbit_meshlet_stream_create(OrbitCtx *ctx, uint64_t first_meshlet, uint64_t capacity,
                                    OrbitMeshletStream **out_stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!out_stream) return fail(ctx, ORBIT_E_MISSING, "meshlet_stream_create: out_stream is NULL");
    *out_stream = nullptr;
    // 32-bit meshlet indices; the kernels address the arrays through buffer resources (32-bit byte offsets: 16 B per
    // meshlet stay below 2 GiB, so the "no access" offset of meshlet_cull.hip lies outside every array)
    if (capacity == 0 || capacity > (1ull << 27) || first_meshlet + capacity > 0xFFFFFFFFull)
        return fail(ctx, ORBIT_E_INVALID, "meshlet_stream_create: range [%llu, +%llu) (at most 2^27 meshlets per stream)",
                    (unsigned long long)first_meshlet, (unsigned long long)capacity);
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipSetDevice");
    OrbitMeshletStream *ms = new (std::nothrow) OrbitMeshletStream();
    if (!ms) return fail(ctx, ORBIT_E_HIP, "out of host memory");
    ms->device = ctx->device;
    ms->first = first_meshlet;
    ms->capacity = capacity;
    // one bit per meshlet in words aligned to global index 0, plus the word behind a record that ends the range
    const size_t bit_words = (size_t)(((first_meshlet + capacity + 31u) >> 5) - (first_meshlet >> 5)) + 2u;
    struct {
        void **p;
        size_t bytes;
        int fill;
    } arrays[] = {{(void **)&ms->sphere, capacity * sizeof(uint4), 0},
                  {(void **)&ms->cone, capacity * sizeof(uint32_t), 0},
                  {(void **)&ms->mat, capacity * sizeof(uint16_t) + 256u, 0},
                  {(void **)&ms->cmd, capacity * 12u, 0},
                  {(void **)&ms->cnt, capacity * sizeof(uint16_t) + 256u, 0},
                  {(void **)&ms->link, bit_words * sizeof(uint32_t), 0},
                  {(void **)&ms->base32, bit_words * sizeof(uint2), 0},
                  {(void **)&ms->cls0, bit_words * sizeof(uint32_t), 0xFF},  // class 3: look the material up
                  {(void **)&ms->cls1, bit_words * sizeof(uint32_t), 0xFF}};
    // Zero-filled: a meshlet inside the derived range that no update has reached (a gap between two uploads) is a
    // defined, empty meshlet — never uninitialised memory.
    for (auto &a : arrays) {
        if (e == hipSuccess) e = hipMalloc(a.p, a.bytes);
        if (e == hipSuccess) e = hipMemset(*a.p, a.fill, a.bytes); // (waited for below)
    }
    if (e == hipSuccess) e = hipMalloc((void **)&ms->d_other, 256);
    if (e == hipSuccess) e = memset_now(ms->d_other, 0, 256); // ... and every fill above: updates launch on the caller's streams
    if (e == hipSuccess) e = hipHostMalloc((void **)&ms->h_other, sizeof(uint32_t), hipHostMallocDefault);
    if (e == hipSuccess) *ms->h_other = 0u;
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ms->other_event, hipEventDisableTiming);
    if (e != hipSuccess) {
        orbit_meshlet_stream_destroy(ms);
        return hip_fail(ctx, e, "meshlet_stream_create: hipMalloc (36.65 B per meshlet)");
    }
    *out_stream = ms;
    return ORBIT_OK;
}

int32_t orbit_meshlet_stream_update(OrbitCtx *ctx, OrbitMeshletStream *ms, const void *meshlet_buffer, uint64_t first,
                                    uint64_t count, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ms || !meshlet_buffer) return fail(ctx, ORBIT_E_MISSING, "meshlet_stream_update: NULL argument");
    if (ms->device != ctx->device) return fail(ctx, ORBIT_E_INVALID, "meshlet_stream_update: stream lives on device %d", ms->device);
    if (first < ms->first || count > ms->capacity || first - ms->first > ms->capacity - count)
        return fail(ctx, ORBIT_E_CAPACITY, "meshlet_stream_update: [%llu, +%llu) outside the stream's [%llu, +%llu)",
                    (unsigned long long)first, (unsigned long long)count, (unsigned long long)ms->first,
                    (unsigned long long)ms->capacity);
    std::lock_guard<std::mutex> slock(ms->mu);
    // What was derived from another buffer says nothing about this one: the readable range starts over.  (The arrays
    // keep the other buffer's values outside [first, first + count); they are outside the new range too.)
    const bool same = ms->source == meshlet_buffer && ms->valid_hi > ms->valid_lo;
    const uint64_t lo = same ? (first < ms->valid_lo ? first : ms->valid_lo) : first;
    const uint64_t hi = same ? (first + count > ms->valid_hi ? first + count : ms->valid_hi) : first + count;
    // the link bit in front of / behind the range looks at the neighbour's derived copy: only inside the same buffer's range
    MeshletStreamView v = stream_arrays(ms, same ? ms->valid_lo : first, same ? ms->valid_hi : first + count);
    if (v.first > first) v.count += v.first - (uint32_t)first, v.first = (uint32_t)first;
    if ((uint64_t)v.first + v.count < first + count) v.count = (uint32_t)(first + count - v.first);
    hipError_t e = launch_meshlet_stream_build((const OrbitMeshlet *)meshlet_buffer, first, count, v,
                                               (const OrbitMaterialData *)ms->materials, ms->material_count, ms->d_other,
                                               (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch meshlet_stream_build");
    if (ms->materials != nullptr && count != 0) { // the range's classes were derived: did a class 3 appear?
        e = read_back_class_flag(ms, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(ctx, e, "meshlet_stream_update: class flag read-back");
    }
    if (count != 0) { // the stream mirrors the buffer only once the launch is enqueued
        ms->source = meshlet_buffer;
        ms->valid_lo = lo;
        ms->valid_hi = hi;
    }
    return ORBIT_OK;
}

int32_t orbit_meshlet_stream_set_materials(OrbitCtx *ctx, OrbitMeshletStream *ms, const void *material_buffer,
                                           uint32_t material_count, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::unique_lock<std::mutex> lock(ctx->mu);
    if (!ms) return fail(ctx, ORBIT_E_MISSING, "meshlet_stream_set_materials: stream is NULL");
    if (ms->device != ctx->device)
        return fail(ctx, ORBIT_E_INVALID, "meshlet_stream_set_materials: stream lives on device %d", ms->device);
    if (material_buffer && material_count == 0)
        return fail(ctx, ORBIT_E_INVALID, "meshlet_stream_set_materials: material_count is 0");
    std::unique_lock<std::mutex> slock(ms->mu);
    // every meshlet the arrays hold, derived or not (the indices of never-derived ones are zero)
    ms->other_pending = true; // no cull trusts the classes until the flag of THIS derivation has come back
    ms->other_recorded = false;
    hipError_t e = hipMemsetAsync(ms->d_other, 0, sizeof(uint32_t), (hipStream_t)stream);
    if (e == hipSuccess)
        e = launch_meshlet_stream_classes(stream_arrays(ms, ms->first, ms->first + ms->capacity),
                                          (const OrbitMaterialData *)material_buffer, material_buffer ? material_count : 0u,
                                          ms->d_other, (hipStream_t)stream);
    if (e == hipSuccess) e = read_back_class_flag(ms, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch meshlet_stream_classes");
    ms->materials = material_buffer;
    ms->material_count = material_buffer ? material_count : 0u;
    // Resolved before the call returns (an upload-time call; nothing to wait for while `stream` is only being
    // captured): the first cull after set_materials takes the class kernel or the index kernel because of what the
    // stream holds, never because of when the flag's copy happened to land.  The wait itself is made WITHOUT the two
    // locks — it lasts as long as everything already queued on `stream`, and other threads' enqueues on this context
    // (or on any context the stream is bound to) have nothing to do with it; nothing is written behind it: the next
    // cull's event query finds the event complete and clears `other_pending` itself (stream_view_for).
    const bool wait = ms->other_recorded;
    hipEvent_t ev = ms->other_event;
    slock.unlock();
    lock.unlock();
    if (wait) {
        e = hipEventSynchronize(ev);
        if (e != hipSuccess) {
            lock.lock();
            return hip_fail(ctx, e, "hipEventSynchronize(alpha-class flag)");
        }
    }
    return ORBIT_OK;
}

int32_t orbit_meshlet_stream_update_meshes(OrbitCtx *ctx, OrbitMeshletStream *ms, const void *mesh_info_buffer,
                                           uint32_t first_mesh, uint32_t count, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ms) return fail(ctx, ORBIT_E_MISSING, "meshlet_stream_update_meshes: stream is NULL");
    if (ms->device != ctx->device)
        return fail(ctx, ORBIT_E_INVALID, "meshlet_stream_update_meshes: stream lives on device %d", ms->device);
    std::lock_guard<std::mutex> slock(ms->mu);
    if (!mesh_info_buffer) { // forget: entity culls read the MeshInfos again
        ms->mesh_source = nullptr;
        ms->mesh_hi = 0u;
        return ORBIT_OK;
    }
    if ((uint64_t)first_mesh + count > 0xFFFFFFFFull) return fail(ctx, ORBIT_E_INVALID, "meshlet_stream_update_meshes: range");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipSetDevice");
    if (ms->mesh_side == nullptr) { // the first update allocates (an upload-time call, like orbit_meshlet_stream_create)
        const uint32_t cap = first_mesh + count > kMeshSideMinCapacity ? first_mesh + count : kMeshSideMinCapacity;
        e = hipMalloc((void **)&ms->mesh_side, (size_t)cap * sizeof(MeshSide));
        if (e == hipSuccess) e = memset_now(ms->mesh_side, 0, (size_t)cap * sizeof(MeshSide));
        if (e != hipSuccess) {
            (void)hipFree(ms->mesh_side);
            ms->mesh_side = nullptr;
            return hip_fail(ctx, e, "meshlet_stream_update_meshes: hipMalloc (32 B per mesh)");
        }
        ms->mesh_capacity = cap;
    }
    if (first_mesh + count > ms->mesh_capacity)
        return fail(ctx, ORBIT_E_CAPACITY, "meshlet_stream_update_meshes: meshes [%u, +%u) beyond the table's %u", first_mesh,
                    count, ms->mesh_capacity);
    if (ms->mesh_source != mesh_info_buffer && ms->mesh_hi != 0u) { // another buffer: what was derived says nothing about it
        e = hipMemsetAsync(ms->mesh_side, 0, (size_t)ms->mesh_hi * sizeof(MeshSide), (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(ctx, e, "meshlet_stream_update_meshes: hipMemsetAsync");
        ms->mesh_hi = 0u;
    }
    e = launch_mesh_side_build((const OrbitMeshInfo *)mesh_info_buffer, first_mesh, count, ms->mesh_side, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch mesh_side_build");
    if (count != 0u) {
        ms->mesh_source = mesh_info_buffer;
        if (first_mesh + count > ms->mesh_hi) ms->mesh_hi = first_mesh + count;
    }
    return ORBIT_OK;
}

uint64_t orbit_ctx_mesh_side_culls(const OrbitCtx *ctx) {
    if (!ctx || !ctx->meshlet_stream) return 0;
    std::lock_guard<std::mutex> lock(ctx->meshlet_stream->mu);
    return ctx->meshlet_stream->mesh_side_culls;
}

int32_t orbit_meshlet_stream_validate(OrbitCtx *ctx, OrbitMeshletStream *ms, const void *meshlet_buffer,
                                      const void *material_buffer, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ms || !meshlet_buffer) return fail(ctx, ORBIT_E_MISSING, "meshlet_stream_validate: NULL argument");
    if (ms->device != ctx->device)
        return fail(ctx, ORBIT_E_INVALID, "meshlet_stream_validate: stream lives on device %d", ms->device);
    {   // the mesh side table against the mesh_info buffer it was derived from (if any)
        std::lock_guard<std::mutex> slock(ms->mu);
        if (ms->mesh_side != nullptr && ms->mesh_source != nullptr && ms->mesh_hi != 0u) {
            const hipError_t me = launch_mesh_side_validate((const OrbitMeshInfo *)ms->mesh_source, 0u, ms->mesh_hi, ms->mesh_side,
                                                            ctx->status, (hipStream_t)stream);
            if (me != hipSuccess) return hip_fail(ctx, me, "launch mesh_side_validate");
        }
    }
    MeshletStreamView v = stream_view_for(ms, meshlet_buffer, nullptr);
    if (!v.sphere) return ORBIT_OK; // mirrors another buffer (or nothing): no cull of this one reads it
    // the classes are checked against the buffer they were derived from, whether or not a cull would read them yet
    const bool classes = material_buffer != nullptr && ms->materials == material_buffer;
    if (classes) {
        const MeshletStreamView all = stream_arrays(ms, ms->first, ms->first + ms->capacity);
        v.cls0 = all.cls0, v.cls1 = all.cls1;
    }
    const hipError_t e = launch_meshlet_stream_validate((const OrbitMeshlet *)meshlet_buffer, v,
                                                        classes ? (const OrbitMaterialData *)material_buffer : nullptr,
                                                        ms->material_count, ctx->status, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch meshlet_stream_validate");
    return ORBIT_OK;
}

int32_t orbit_meshlet_stream_destroy(OrbitMeshletStream *ms) {
    if (!ms) return ORBIT_OK;
    if (ms->bindings.load() > 0)
        return fail(nullptr, ORBIT_E_INVALID, "meshlet_stream_destroy: the stream is still bound to %d context(s)",
                    ms->bindings.load());
    (void)hipFree(ms->sphere);
    (void)hipFree(ms->cone);
    (void)hipFree(ms->mat);
    (void)hipFree(ms->cmd);
    (void)hipFree(ms->cnt);
    (void)hipFree(ms->link);
    (void)hipFree(ms->base32);
    (void)hipFree(ms->cls0);
    (void)hipFree(ms->cls1);
    (void)hipFree(ms->d_other);
    (void)hipFree(ms->mesh_side);
    if (ms->h_other) (void)hipHostFree(ms->h_other);
    if (ms->other_event) (void)hipEventDestroy(ms->other_event);
    delete ms;
    return ORBIT_OK;
}

int32_t orbit_ctx_bind_meshlet_stream(OrbitCtx *ctx, OrbitMeshletStream *ms) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (ms && ms->device != ctx->device)
        return fail(ctx, ORBIT_E_INVALID, "bind_meshlet_stream: stream lives on device %d", ms->device);
    if (ms) ms->bindings.fetch_add(1);
    if (ctx->meshlet_stream) ctx->meshlet_stream->bindings.fetch_sub(1);
    ctx->meshlet_stream = ms;
    return ORBIT_OK;
}

uint64_t orbit_ctx_fused_culls(const OrbitCtx *ctx) { return ctx ? ctx->fused_culls : 0; }
uint64_t orbit_ctx_meshlet_stream_culls(const OrbitCtx *ctx) { return ctx ? ctx->stream_culls : 0; }
uint64_t orbit_ctx_meshlet_class_culls(const OrbitCtx *ctx) { return ctx ? ctx->class_culls : 0; }

int32_t orbit_meshlet_task_cull(OrbitCtx *ctx, const OrbitGpuCullInfo *ci, const OrbitMeshletCullBufs *b,
                                OrbitMeshTaskRecord *task_records, void *stream) {
    return meshlet_cull_impl(ctx, ci, b, task_records, true, stream);
}

int32_t orbit_meshlet_cull_visible_records(OrbitCtx *ctx, const OrbitGpuCullInfo *ci, const OrbitMeshletCullBufs *b,
                                           void *record_buffer, uint32_t record_capacity, void *stream) {
    return meshlet_cull_impl(ctx, ci, b, nullptr, false, stream, record_buffer, record_capacity, 2u);
}

int32_t orbit_meshlet_cull_records_and_commands(OrbitCtx *ctx, const OrbitGpuCullInfo *ci, const OrbitMeshletCullBufs *b,
                                                void *record_buffer, uint32_t record_capacity, void *stream) {
    return meshlet_cull_impl(ctx, ci, b, nullptr, false, stream, record_buffer, record_capacity, 2u, true);
}

// One shard's cull of the sharded engine as one call: orbit_entity_cull_range + orbit_meshlet_cull_visible_records (or
// _records_and_commands) — and, for pass 0 and at most kShardMaxChunks x 256 entity-draws, ONE launch for both stages
// and the list (+ the emit launch for the commands).
int32_t orbit_cull_shard(OrbitCtx *ctx, const OrbitGpuCullInfo *ci, const OrbitEntityCullBufs *eb, uint32_t draw_first,
                         uint32_t draw_count, const OrbitMeshletCullBufs *mb, void *record_buffer,
                         uint32_t record_capacity, uint32_t with_commands, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (ctx->rec_shift != 5u)
        return fail(ctx, ORBIT_E_INVALID, "cull_shard: the sharded engine's record list is defined for dispatch_size 32; this "
                                          "context has %u", ctx->caps.dispatch_size);
    EntityCullParams ep;
    MeshletCullParams p;
    int32_t rc = entity_cull_params(ctx, ctx, ci, eb, draw_first, draw_count, true, ep);
    if (rc == ORBIT_OK) rc = meshlet_cull_params(ctx, ctx, ci, mb, nullptr, false, record_buffer, record_capacity, 2u, p);
    if (rc != ORBIT_OK) return rc;
    if (mb->meshlet_dispatch_buffer != eb->meshlet_dispatch_buffer)
        return fail(ctx, ORBIT_E_INVALID, "cull_shard: the meshlet stage must read the entity stage's dispatch buffer");
    if (with_commands) {
        if (!mb->draw_commands_buffer) return fail(ctx, ORBIT_E_MISSING, "cull_shard: draw_commands_buffer is NULL");
        p.also_commands = (uint8_t *)mb->draw_commands_buffer;
        p.also_commands_capacity = mb->draw_capacity;
    }
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ctx->profiling && (ctx->prof_calls++ % ctx->profiling) == 0) {
        if (ctx->prof_used + 2 > ctx->prof_events.size()) {
            for (int i = 0; i < 2; i++) {
                hipEvent_t ev;
                if (hipEventCreate(&ev) != hipSuccess) return fail(ctx, ORBIT_E_HIP, "hipEventCreate");
                ctx->prof_events.push_back(ev);
            }
        }
        ev0 = ctx->prof_events[ctx->prof_used];
        ev1 = ctx->prof_events[ctx->prof_used + 1];
        ctx->prof_used += 2;
    }
    if (p.ms.sphere && ctx->caps.validate_streams) { // debug: is the stream still a mirror?
        const hipError_t ve = launch_meshlet_stream_validate(p.meshlets, p.ms, p.ms.cls0 ? p.materials : nullptr,
                                                             ctx->meshlet_stream->material_count, ctx->status, s);
        if (ve != hipSuccess) return hip_fail(ctx, ve, "launch meshlet_stream_validate");
    }
    if (ep.mesh_side.table && ctx->caps.validate_streams) { // ... and the mesh side table (as orbit_entity_cull does)
        const hipError_t ve = launch_mesh_side_validate(ep.mesh_infos, 0u, ep.mesh_side.count, ep.mesh_side.table, ctx->status, s);
        if (ve != hipSuccess) return hip_fail(ctx, ve, "launch mesh_side_validate");
    }
    const uint32_t chunks = (draw_count + 255u) / 256u;
    const bool one_launch = ci->occlusion_pass == 0u && chunks >= 1u && chunks <= kShardMaxChunks && ctx->caps.cull_path != 1u;
    hipError_t e;
    if (one_launch) {
        FusedCullParams f{};
        f.m = p;
        f.entity_draw_buffer = ep.entity_draw_buffer;
        f.mesh_infos = ep.mesh_infos;
        f.mesh_side = ep.mesh_side;
        f.visibility = ep.visibility;
        f.e_pyr = ep.pyr;
        f.draw_first = ep.draw_first;
        f.draw_limit = ep.draw_limit;
        f.e_dispatch_capacity = ep.dispatch_capacity;
        f.ne_chunks = chunks;
        f.e_total = ep.total;
        f.sync = ctx->f_sync;
        f.ent_flags = ctx->f_ent_flags;
        f.tile_flags = ctx->f_tile_flags;
        f.done_flags = ctx->f_done;
        if (ev0 && (e = hipEventRecord(ev0, s)) != hipSuccess) return hip_fail(ctx, e, "hipEventRecord");
        e = launch_shard_cull(f, ctx->num_cus, s);
        if (e != hipSuccess) return hip_fail(ctx, e, "launch shard_cull");
        if (ev1 && (e = hipEventRecord(ev1, s)) != hipSuccess) return hip_fail(ctx, e, "hipEventRecord");
        if (with_commands) { // scan + emit of the same evaluation's ballots, into the command buffer
            MeshletCullParams c = p;
            c.visible_list = 0u;
            c.draw_buffer = p.also_commands;
            c.draw_capacity = p.also_commands_capacity;
            e = launch_meshlet_scan_emit(c, ctx->num_cus, s);
            if (e != hipSuccess) return hip_fail(ctx, e, "launch scan + emit");
        }
        ctx->shard_culls++;
    } else {
        e = launch_entity_cull(ep, draw_count, ctx->num_cus, s);
        if (e != hipSuccess) return hip_fail(ctx, e, "launch entity_cull");
        e = launch_meshlet_cull(p, ctx->num_cus, s, ev0, ev1);
        if (e != hipSuccess) return hip_fail(ctx, e, "launch meshlet_cull");
    }
    if (p.ms.sphere && ci->occlusion_pass != 1) {
        ctx->stream_culls++;
        if (p.ms.cls0) ctx->class_culls++;
    }
    return ORBIT_OK;
}

uint64_t orbit_ctx_shard_culls(const OrbitCtx *ctx) { return ctx ? ctx->shard_culls : 0; }

int32_t orbit_expand_visible_records(OrbitCtx *ctx, const void *record_buffer, const void *meshlet_buffer,
                                     void *draw_commands_buffer, uint32_t draw_capacity, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!record_buffer || !meshlet_buffer || !draw_commands_buffer)
        return fail(ctx, ORBIT_E_MISSING, "expand_visible_records: NULL argument");
    const MeshletStreamView view = stream_view_for(ctx->meshlet_stream, meshlet_buffer, nullptr);
    if (view.sphere && ctx->caps.validate_streams)
        (void)launch_meshlet_stream_validate((const OrbitMeshlet *)meshlet_buffer, view, nullptr, 0, ctx->status,
                                             (hipStream_t)stream);
    const hipError_t e = launch_visible_records_expand((const uint8_t *)record_buffer, ctx->x_block_pop, kExpandBlocks,
                                                       (const OrbitMeshlet *)meshlet_buffer, view,
                                                       (uint8_t *)draw_commands_buffer, draw_capacity, ctx->zero_page,
                                                       ctx->status, (hipStream_t)stream);
    if (view.cmd) ctx->stream_culls++;
    if (e != hipSuccess) return hip_fail(ctx, e, "launch visible_records_expand");
    return ORBIT_OK;
}

// ----------------------------------------------------------------- depth_reduce
int32_t orbit_depth_pyramid_desc(uint32_t sw, uint32_t sh, OrbitDepthPyramidDesc *desc) {
    if (!desc || sw == 0 || sh == 0) return fail(nullptr, ORBIT_E_INVALID, "depth_pyramid_desc: bad argument");
    uint32_t w0 = next_pow2(sw) / 2, h0 = next_pow2(sh) / 2; // draw_gen.rs:458
    if (w0 == 0) w0 = 1;
    if (h0 == 0) h0 = 1;
    fill_pyramid_desc(w0, h0, desc);
    return ORBIT_OK;
}

int32_t orbit_depth_pyramid_desc_from_mip0(uint32_t w0, uint32_t h0, OrbitDepthPyramidDesc *desc) {
    if (!desc || w0 == 0 || h0 == 0) return fail(nullptr, ORBIT_E_INVALID, "depth_pyramid_desc: bad argument");
    fill_pyramid_desc(w0, h0, desc);
    return ORBIT_OK;
}

// Validates a batch of pyramids and fills the launch's parameter block (ctx->mu held; nothing is enqueued).
static int32_t prepare_depth_reduce(OrbitCtx *ctx, const OrbitDepthReduceItem *items, uint32_t count, DepthReduceBatch &b) {
    if (!items || count == 0) return fail(ctx, ORBIT_E_MISSING, "depth_reduce: no items");
    if (count > ORBIT_MAX_PYRAMID_BATCH)
        return fail(ctx, ORBIT_E_CAPACITY, "depth_reduce: %u pyramids in one batch (max %u)", count,
                    (unsigned)ORBIT_MAX_PYRAMID_BATCH);
    static_assert(ORBIT_MAX_PYRAMID_BATCH == kMaxPyramidBatch, "batch size");
    b = DepthReduceBatch{};
    b.count = count;
    b.tickets = ctx->d_tickets;
    for (uint32_t i = 0; i < count; i++) {
        const OrbitDepthReduceItem &it = items[i];
        if (!it.depth || (!it.pyramid) == (!it.levels))
            return fail(ctx, ORBIT_E_MISSING, "depth_reduce item %u: depth, and exactly one of pyramid / levels", i);
        OrbitDepthPyramidDesc d;
        const int32_t rc = orbit_depth_pyramid_desc(it.screen_width, it.screen_height, &d);
        if (rc) return rc;
        DepthReduceParams &p = b.p[i];
        p.depth = it.depth;
        p.depth_pitch = it.depth_row_pitch ? it.depth_row_pitch : it.screen_width;
        if (p.depth_pitch < it.screen_width)
            return fail(ctx, ORBIT_E_INVALID, "depth_reduce item %u: row pitch %u < width %u", i, p.depth_pitch,
                        it.screen_width);
        p.screen_w = it.screen_width;
        p.screen_h = it.screen_height;
        p.w0 = d.width;
        p.h0 = d.height;
        p.mips = d.mip_levels;
        for (uint32_t k = 0; k < d.mip_levels; k++) {
            if (it.pyramid) {
                p.level[k] = it.pyramid + d.mip_offset[k];
                p.pitch[k] = d.mip_width[k];
            } else {
                if (!it.levels[k].texels || it.levels[k].row_pitch < d.mip_width[k])
                    return fail(ctx, ORBIT_E_INVALID, "depth_reduce item %u: level %u is NULL or its pitch %u < %u", i, k,
                                it.levels[k].row_pitch, d.mip_width[k]);
                p.level[k] = it.levels[k].texels;
                p.pitch[k] = it.levels[k].row_pitch;
            }
        }
    }
    return ORBIT_OK;
}

int32_t orbit_depth_reduce_multi(OrbitCtx *ctx, const OrbitDepthReduceItem *items, uint32_t count, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DepthReduceBatch b;
    const int32_t rc = prepare_depth_reduce(ctx, items, count, b);
    if (rc != ORBIT_OK) return rc;
    const hipError_t e = launch_depth_reduce(b, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch depth_reduce");
    return ORBIT_OK;
}

int32_t orbit_depth_reduce(OrbitCtx *ctx, const float *depth, uint32_t sw, uint32_t sh, float *pyramid,
                           void *stream) {
    OrbitDepthReduceItem it{};
    it.depth = depth;
    it.screen_width = sw;
    it.screen_height = sh;
    it.pyramid = pyramid;
    if (ctx && (!depth || !pyramid)) {
        std::lock_guard<std::mutex> lock(ctx->mu);
        return fail(ctx, ORBIT_E_MISSING, "depth_reduce: NULL buffer");
    }
    return orbit_depth_reduce_multi(ctx, &it, 1, stream);
}

// ---------------------------------------------------------------- light_cluster
// The three stages behind their entry points (ctx->mu held).  `count_chunks`: the mark launch also takes the
// compaction's chunk counts (orbit_compute_clusters; the words are zero: cleared at creation and by every assignment
// that follows a counting mark); `counted`: the compaction finds them there; `clear_counts`: the assignment's first
// launch clears them again.
static int32_t cluster_mark_locked(OrbitCtx *ctx, const OrbitMarkActivePush *push, const float *depth,
                                   uint32_t *tile_depth_slice_mask, OrbitClusterDepthBounds *depth_bounds, bool count_chunks,
                                   void *stream) {
    if (!push || !depth || !tile_depth_slice_mask || !depth_bounds)
        return fail(ctx, ORBIT_E_MISSING, "cluster_mark: NULL argument");
    if (push->cluster_count[2] > 32 || push->tile_size_px == 0 || push->depth_buffer_sample_count == 0)
        return fail(ctx, ORBIT_E_INVALID, "cluster_mark: z slices %u (> 32), tile %u or samples %u invalid",
                    push->cluster_count[2], push->tile_size_px, push->depth_buffer_sample_count);
    ClusterMarkParams p;
    p.pc = *push;
    p.depth = depth;
    p.masks = tile_depth_slice_mask;
    p.bounds = depth_bounds;
    p.chunk_counts = count_chunks ? ctx->c_chunk + ctx->c_chunk_words : nullptr;
    hipError_t e = launch_cluster_mark(p, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch cluster_mark");
    return ORBIT_OK;
}

static int32_t cluster_compact_check(OrbitCtx *ctx, const uint32_t cluster_count[3], const uint32_t *masks,
                                     void *unique_cluster_buffer) {
    if (!cluster_count || !masks || !unique_cluster_buffer)
        return fail(ctx, ORBIT_E_MISSING, "cluster_compact: NULL argument");
    const uint64_t total = (uint64_t)cluster_count[0] * cluster_count[1] * cluster_count[2];
    if (cluster_count[2] > 32 || total > ctx->caps.max_clusters)
        return fail(ctx, ORBIT_E_CAPACITY, "cluster grid %ux%ux%u exceeds caps.max_clusters %u or 32 slices",
                    cluster_count[0], cluster_count[1], cluster_count[2], ctx->caps.max_clusters);
    return ORBIT_OK;
}

static int32_t cluster_compact_locked(OrbitCtx *ctx, const uint32_t cluster_count[3], const uint32_t *masks,
                                      void *unique_cluster_buffer, uint32_t index_capacity, bool counted, void *stream) {
    const int32_t rc = cluster_compact_check(ctx, cluster_count, masks, unique_cluster_buffer);
    if (rc != ORBIT_OK) return rc;
    ClusterCompactParams p;
    memcpy(p.cc, cluster_count, 12);
    p.masks = masks;
    p.unique = (uint8_t *)unique_cluster_buffer;
    p.index_capacity = index_capacity;
    p.chunk_counts = counted ? ctx->c_chunk + ctx->c_chunk_words : ctx->c_chunk;
    p.status = ctx->status;
    p.counted = counted;
    hipError_t e = launch_cluster_compact(p, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch cluster_compact");
    return ORBIT_OK;
}

static int32_t cluster_assign_check(OrbitCtx *ctx, const OrbitClusterCullInfo *info, const void *unique_cluster_buffer,
                                    const OrbitClusterDepthBounds *depth_bounds, const OrbitLightData *lights,
                                    void *light_index_buffer, uint32_t *cluster_offset_image) {
    if (!info || !unique_cluster_buffer || !depth_bounds || !light_index_buffer || !cluster_offset_image)
        return fail(ctx, ORBIT_E_MISSING, "cluster_assign: NULL argument");
    if (info->global_light_count > 0 && !lights) return fail(ctx, ORBIT_E_MISSING, "cluster_assign: lights is NULL");
    if (info->global_light_count > ctx->caps.max_lights)
        return fail(ctx, ORBIT_E_CAPACITY, "light count %u > caps.max_lights %u", info->global_light_count,
                    ctx->caps.max_lights);
    const uint64_t total = (uint64_t)info->cluster_count[0] * info->cluster_count[1] * info->cluster_count[2];
    if (total > ctx->caps.max_clusters)
        return fail(ctx, ORBIT_E_CAPACITY, "cluster grid exceeds caps.max_clusters %u", ctx->caps.max_clusters);
    return ORBIT_OK;
}

static int32_t cluster_assign_locked(OrbitCtx *ctx, const OrbitClusterCullInfo *info, const void *unique_cluster_buffer,
                                     const OrbitClusterDepthBounds *depth_bounds, const OrbitLightData *lights,
                                     void *light_index_buffer, uint32_t light_index_capacity, uint32_t *cluster_offset_image,
                                     uint32_t clear_counts, void *stream) {
    const int32_t rc = cluster_assign_check(ctx, info, unique_cluster_buffer, depth_bounds, lights, light_index_buffer,
                                            cluster_offset_image);
    if (rc != ORBIT_OK) return rc;
    const uint64_t total = (uint64_t)info->cluster_count[0] * info->cluster_count[1] * info->cluster_count[2];
    ClusterAssignParams p;
    p.info = *info;
    p.unique = (const uint8_t *)unique_cluster_buffer;
    p.bounds = depth_bounds;
    p.lights = lights;
    p.light_index_buffer = (uint8_t *)light_index_buffer;
    p.light_index_capacity = light_index_capacity;
    p.offset_image = cluster_offset_image;
    p.max_clusters = (uint32_t)total;
    p.view_lights = ctx->a_view_lights;
    p.light_flags = ctx->a_light_flags;
    p.counts = ctx->a_counts;
    p.aabb = ctx->a_aabb;
    p.coarse = ctx->a_coarse;
    p.coarse_lights = ctx->a_coarse_lights;
    p.hit_cache = ctx->a_hit_cache;
    p.coarse_counts = ctx->a_coarse_counts;
    p.group_box = ctx->a_group_box;
    p.group_order = ctx->a_group_order;
    p.coarse_seg = ctx->a_coarse_seg;
    p.block_sums = ctx->a_block_sums;
    p.block_base = ctx->a_block_base;
    p.total = ctx->a_total;
    p.zero_words = clear_counts ? ctx->c_chunk + ctx->c_chunk_words : nullptr;
    p.zero_count = clear_counts;
    p.status = ctx->status;
    p.debug_tests = ctx->debug_cycles + 16;
    hipError_t e = launch_cluster_assign(p, ctx->num_cus, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch cluster_assign");
    return ORBIT_OK;
}

int32_t orbit_cluster_mark(OrbitCtx *ctx, const OrbitMarkActivePush *push, const float *depth,
                           uint32_t *tile_depth_slice_mask, OrbitClusterDepthBounds *depth_bounds, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    return cluster_mark_locked(ctx, push, depth, tile_depth_slice_mask, depth_bounds, false, stream);
}

int32_t orbit_cluster_compact(OrbitCtx *ctx, const uint32_t cluster_count[3], const uint32_t *masks,
                              void *unique_cluster_buffer, uint32_t index_capacity, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    return cluster_compact_locked(ctx, cluster_count, masks, unique_cluster_buffer, index_capacity, false, stream);
}

int32_t orbit_cluster_assign(OrbitCtx *ctx, const OrbitClusterCullInfo *info, const void *unique_cluster_buffer,
                             const OrbitClusterDepthBounds *depth_bounds, const OrbitLightData *lights,
                             void *light_index_buffer, uint32_t light_index_capacity, uint32_t *cluster_offset_image,
                             void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    return cluster_assign_locked(ctx, info, unique_cluster_buffer, depth_bounds, lights, light_index_buffer,
                                 light_index_capacity, cluster_offset_image, 0u, stream);
}

// Everything orbit_compute_clusters would refuse, refused before its first launch (ctx->mu held): the mark launch leaves
// the compaction's chunk counts behind, and only the assignment's first launch clears them again.
static int32_t compute_clusters_check(OrbitCtx *ctx, const OrbitMarkActivePush *push, const OrbitClusterCullInfo *info,
                                      const float *depth, const OrbitLightData *lights, uint32_t *tile_depth_slice_mask,
                                      OrbitClusterDepthBounds *depth_bounds, void *unique_cluster_buffer,
                                      void *light_index_buffer, uint32_t *cluster_offset_image) {
    if (!push || !info) return fail(ctx, ORBIT_E_MISSING, "compute_clusters: NULL parameter block");
    for (int i = 0; i < 3; i++)
        if (push->cluster_count[i] != info->cluster_count[i])
            return fail(ctx, ORBIT_E_INVALID, "compute_clusters: cluster_count[%d] differs between push (%u) and info (%u)",
                        i, push->cluster_count[i], info->cluster_count[i]);
    if (!depth || !tile_depth_slice_mask || !depth_bounds) return fail(ctx, ORBIT_E_MISSING, "cluster_mark: NULL argument");
    if (push->cluster_count[2] > 32 || push->tile_size_px == 0 || push->depth_buffer_sample_count == 0)
        return fail(ctx, ORBIT_E_INVALID, "cluster_mark: z slices %u (> 32), tile %u or samples %u invalid",
                    push->cluster_count[2], push->tile_size_px, push->depth_buffer_sample_count);
    int32_t rc = cluster_compact_check(ctx, push->cluster_count, tile_depth_slice_mask, unique_cluster_buffer);
    if (rc == ORBIT_OK)
        rc = cluster_assign_check(ctx, info, unique_cluster_buffer, depth_bounds, lights, light_index_buffer, cluster_offset_image);
    return rc;
}

static int32_t compute_clusters_launch(OrbitCtx *ctx, const OrbitMarkActivePush *push, const OrbitClusterCullInfo *info,
                                       const float *depth, const OrbitLightData *lights, uint32_t *tile_depth_slice_mask,
                                       OrbitClusterDepthBounds *depth_bounds, void *unique_cluster_buffer,
                                       uint32_t index_capacity, void *light_index_buffer, uint32_t light_index_capacity,
                                       uint32_t *cluster_offset_image, void *stream) {
    // stream order is the only dependency between the stages (cluster.rs:380-395)
    int32_t rc = cluster_mark_locked(ctx, push, depth, tile_depth_slice_mask, depth_bounds, true, stream);
    if (rc != ORBIT_OK) return rc; // (refused before its launch: nothing was counted)
    const uint32_t tiles = push->cluster_count[0] * push->cluster_count[1];
    const uint32_t count_words = push->cluster_count[2] * ((tiles + 1023u) / 1024u);
    rc = cluster_compact_locked(ctx, push->cluster_count, tile_depth_slice_mask, unique_cluster_buffer, index_capacity, true,
                                stream);
    if (rc == ORBIT_OK)
        rc = cluster_assign_locked(ctx, info, unique_cluster_buffer, depth_bounds, lights, light_index_buffer,
                                   light_index_capacity, cluster_offset_image, count_words, stream);
    if (rc != ORBIT_OK) // a launch failed behind the counting mark: the words must not stay
        (void)hipMemsetAsync(ctx->c_chunk + ctx->c_chunk_words, 0, (size_t)count_words * 4u, (hipStream_t)stream);
    return rc;
}

int32_t orbit_compute_clusters(OrbitCtx *ctx, const OrbitMarkActivePush *push, const OrbitClusterCullInfo *info,
                               const float *depth, const OrbitLightData *lights, uint32_t *tile_depth_slice_mask,
                               OrbitClusterDepthBounds *depth_bounds, void *unique_cluster_buffer,
                               uint32_t index_capacity, void *light_index_buffer, uint32_t light_index_capacity,
                               uint32_t *cluster_offset_image, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    const int32_t rc = compute_clusters_check(ctx, push, info, depth, lights, tile_depth_slice_mask, depth_bounds,
                                              unique_cluster_buffer, light_index_buffer, cluster_offset_image);
    if (rc != ORBIT_OK) return rc;
    return compute_clusters_launch(ctx, push, info, depth, lights, tile_depth_slice_mask, depth_bounds, unique_cluster_buffer,
                                   index_capacity, light_index_buffer, light_index_capacity, cluster_offset_image, stream);
}

// The late half of a frame — everything the renderer records between "the depth buffer exists" and the forward pass
// (src/app.rs:1151-1212: the late cull of render_depth_prepass, render_shadows' cascade culls, compute_clusters) — as ONE
// call whose independent chains run side by side: {pyramids -> pass-2 culls} on the caller's stream, {cascade culls} and
// {compute_clusters} on two streams of the context, forked behind what the caller had enqueued and joined before the call
// returns control of the stream (events only: capturable).  Each chain is a handful of dependent latency-bound launches on
// a device that is 95 % idle during any one of them; serially they cost their sum, side by side the longest.
int32_t orbit_frame_late(OrbitCtx *ctx, const OrbitFrameLate *f, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!f) return fail(ctx, ORBIT_E_MISSING, "frame_late: NULL descriptor");
    if (f->late_view_count + f->cascade_view_count > ORBIT_MAX_CULL_VIEWS)
        return fail(ctx, ORBIT_E_CAPACITY, "frame_late: %u views (max %u)", f->late_view_count + f->cascade_view_count,
                    (unsigned)ORBIT_MAX_CULL_VIEWS);
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipSetDevice");
    // everything is validated (and every scratch set exists) before anything is enqueued or forked
    DepthReduceBatch pyr;
    PreparedCullViews late, casc;
    int32_t rc = ORBIT_OK;
    if (f->pyramid_count) rc = prepare_depth_reduce(ctx, f->pyramids, f->pyramid_count, pyr);
    if (rc == ORBIT_OK && f->late_view_count) rc = prepare_cull_views(ctx, f->late_views, f->late_view_count, 0u, late);
    if (rc == ORBIT_OK && f->cascade_view_count)
        rc = prepare_cull_views(ctx, f->cascade_views, f->cascade_view_count, f->late_view_count, casc);
    const OrbitClusterFrame *c = f->clusters;
    if (rc == ORBIT_OK && c)
        rc = compute_clusters_check(ctx, c->push, c->info, c->depth, c->lights, c->tile_depth_slice_mask, c->depth_bounds,
                                    c->unique_cluster_buffer, c->light_index_buffer, c->cluster_offset_image);
    if (rc != ORBIT_OK) return rc;
    const bool chain_a = f->pyramid_count || f->late_view_count, chain_b = f->cascade_view_count != 0, chain_c = c != nullptr;
    // the side streams and the four events: created by the first call that forks (never while anything is enqueued)
    if ((chain_b || chain_c) && !ctx->side_stream[0]) {
        for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipStreamCreateWithFlags(&ctx->side_stream[i], hipStreamNonBlocking);
        for (int i = 0; i < 3 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ctx->side_event[i], hipEventDisableTiming);
        if (e != hipSuccess) return hip_fail(ctx, e, "frame_late: side streams");
    }
    hipStream_t s = (hipStream_t)stream;
    // The LONGEST chain stays on the caller's stream: a forked chain starts a fork's latency late (the side stream waits
    // for an event of `s`: a cross-queue dependency, ~5 us) and its join is a wait in `s` — both are hidden only behind
    // work `s` itself still has.  Measured on configs 3 + 4 as a replayed graph: the cluster chain (59 us alone) forked
    // beside {pyramid, late cull} on `s` took 76 us; on `s` with the others forked, what the chain itself takes.  Weight =
    // dependent launches of the chain.
    const uint32_t w_a = (f->pyramid_count ? 1u : 0u) + (f->late_view_count ? (late.fused ? 1u : 5u) : 0u);
    const uint32_t w_b = chain_b ? (casc.fused ? 1u : 5u) : 0u, w_c = chain_c ? 6u : 0u;
    hipStream_t sa = s, sb = s, sc = s;
    {
        const uint32_t heaviest = (w_c >= w_a && w_c >= w_b) ? 2u : (w_a >= w_b ? 0u : 1u);
        int side = 0;
        if (chain_a && heaviest != 0u) sa = ctx->side_stream[side++];
        if (chain_b && heaviest != 1u) sb = ctx->side_stream[side++];
        if (chain_c && heaviest != 2u) sc = ctx->side_stream[side++];
    }
    const bool fork_a = chain_a && sa != s, fork_b = chain_b && sb != s, fork_c = chain_c && sc != s;
    if (fork_a || fork_b || fork_c) {
        e = hipEventRecord(ctx->side_event[0], s);
        if (e == hipSuccess && fork_a) e = hipStreamWaitEvent(sa, ctx->side_event[0], 0);
        if (e == hipSuccess && fork_b) e = hipStreamWaitEvent(sb, ctx->side_event[0], 0);
        if (e == hipSuccess && fork_c) e = hipStreamWaitEvent(sc, ctx->side_event[0], 0);
        if (e != hipSuccess) return hip_fail(ctx, e, "frame_late: fork");
    }
    // Enqueue order.  Eagerly the call is bound by the HOST (fourteen launches and seven event operations are ~45 us of
    // enqueueing for chains of 36 and 60 us): the forked chains go first — they run while the host still enqueues the
    // long one — 80.8 us for configs 3 + 4 against 89.0 the other way round and 89.6 serially.  While the stream is being
    // CAPTURED the order only decides which branch the graph runs on the launch stream, and that should be the long one:
    // replayed, 81.6 us against 93.3 (profiles/r06_notes.md, all four arrangements).
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &capture) == hipSuccess && capture != hipStreamCaptureStatusNone;
    auto run_a = [&]() -> int32_t {
        if (f->pyramid_count) {
            const hipError_t le = launch_depth_reduce(pyr, sa);
            if (le != hipSuccess) return hip_fail(ctx, le, "launch depth_reduce");
        }
        return f->late_view_count ? launch_prepared_cull_views(ctx, late, sa) : ORBIT_OK;
    };
    auto run_b = [&]() -> int32_t { return launch_prepared_cull_views(ctx, casc, sb); };
    auto run_c = [&]() -> int32_t {
        return compute_clusters_launch(ctx, c->push, c->info, c->depth, c->lights, c->tile_depth_slice_mask, c->depth_bounds,
                                       c->unique_cluster_buffer, c->index_capacity, c->light_index_buffer,
                                       c->light_index_capacity, c->cluster_offset_image, sc);
    };
    for (int pass = 0; pass < 2 && rc == ORBIT_OK; pass++) {
        const bool forked_now = capturing ? pass == 1 : pass == 0; // eager: forked chains first; captured: the one on `s` first
        if (rc == ORBIT_OK && chain_a && fork_a == forked_now) rc = run_a();
        if (rc == ORBIT_OK && chain_b && fork_b == forked_now) rc = run_b();
        if (rc == ORBIT_OK && chain_c && fork_c == forked_now) rc = run_c();
    }
    // join — also behind a launch error: a forked stream must come back (a capture would otherwise be left unjoined)
    int ev = 1;
    for (hipStream_t side : {fork_a ? sa : nullptr, fork_b ? sb : nullptr, fork_c ? sc : nullptr}) {
        if (!side) continue;
        e = hipEventRecord(ctx->side_event[ev], side);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, ctx->side_event[ev], 0);
        if (e != hipSuccess && rc == ORBIT_OK) rc = hip_fail(ctx, e, "frame_late: join");
        ev++;
    }
    return rc;
}

// ------------------------------------------------------------------- multi-GPU
void orbit_shard_range(uint32_t n, uint32_t rank, uint32_t world, uint32_t *begin, uint32_t *end) {
    // contiguous ranges cut at multiples of 32 so entity-bitset words are rank-private
    if (world == 0) world = 1;
    const uint64_t words = ((uint64_t)n + 31u) / 32u;
    uint64_t b = words * rank / world * 32u, e = words * (rank + 1ull) / world * 32u;
    if (b > n) b = n;
    if (e > n) e = n;
    if (begin) *begin = (uint32_t)b;
    if (end) *end = (uint32_t)e;
}

} // extern "C"

namespace {

// RCCL entry points, resolved from the copy already loaded in the process (the one the caller's
// communicator belongs to); liborbit_cull.so itself does not link RCCL.
struct Rccl {
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
};

const Rccl &rccl() {
    static const Rccl table = [] {
        Rccl t;
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (h) break;
        }
        if (!h)
            for (const char *name : {"librccl.so.1", "librccl.so"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h) break;
            }
        if (!h) return t;
        t.all_gather = (decltype(t.all_gather))dlsym(h, "ncclAllGather");
        t.send = (decltype(t.send))dlsym(h, "ncclSend");
        t.recv = (decltype(t.recv))dlsym(h, "ncclRecv");
        t.group_start = (decltype(t.group_start))dlsym(h, "ncclGroupStart");
        t.group_end = (decltype(t.group_end))dlsym(h, "ncclGroupEnd");
        t.error_string = (decltype(t.error_string))dlsym(h, "ncclGetErrorString");
        t.ok = t.all_gather && t.send && t.recv && t.group_start && t.group_end && t.error_string;
        return t;
    }();
    return table;
}

constexpr uint32_t kMaxGatherWorld = 64; // counts scratch: one 256-B carve

__global__ void write_u32_kernel(uint32_t *dst, uint32_t v) { *dst = v; }

} // namespace

namespace {

// Rank-ordered all-gather of {u32 count @0 | header_bytes | items of `stride` bytes} buffers.
int32_t gather_lists(OrbitCtx *ctx, void *nccl_comm, uint32_t rank, uint32_t world, const void *local_draw_buffer,
                     void *out_draw_buffer, uint32_t out_capacity, void *stream, size_t header_bytes, size_t stride) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!nccl_comm || !local_draw_buffer || !out_draw_buffer)
        return fail(ctx, ORBIT_E_MISSING, "gather_visible: NULL argument");
    if (world == 0 || world > kMaxGatherWorld || rank >= world)
        return fail(ctx, ORBIT_E_INVALID, "gather_visible: rank %u of world %u (max %u)", rank, world, kMaxGatherWorld);
    const Rccl &nc = rccl();
    if (!nc.ok) return fail(ctx, ORBIT_E_COMM, "gather_visible: librccl is not loadable in this process");
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipSetDevice");
#define ORBIT_NCCL(call, what)                                                                  \
    do {                                                                                        \
        const ncclResult_t r_ = (call);                                                         \
        if (r_ != ncclSuccess) return fail(ctx, ORBIT_E_COMM, "%s: %s", what, nc.error_string(r_)); \
    } while (0)
    // 1. counts of all ranks (4 B each), device -> host: the message sizes
    ORBIT_NCCL(nc.all_gather(local_draw_buffer, ctx->g_counts, 1, ncclUint32, comm, s), "ncclAllGather(counts)");
    uint32_t counts[kMaxGatherWorld];
    e = hipMemcpyAsync(counts, ctx->g_counts, world * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipMemcpyAsync(counts)");
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipStreamSynchronize(counts)");
    uint64_t prefix[kMaxGatherWorld + 1];
    prefix[0] = 0;
    for (uint32_t r = 0; r < world; r++) prefix[r + 1] = prefix[r] + counts[r];
    if (prefix[world] > out_capacity)
        return fail(ctx, ORBIT_E_CAPACITY, "gather_visible: %llu commands > out_capacity %u",
                    (unsigned long long)prefix[world], out_capacity);
    // 2. header = total; 3. every list straight into out + prefix[rank]
    uint8_t *out = (uint8_t *)out_draw_buffer;
    const uint8_t *mine = (const uint8_t *)local_draw_buffer + header_bytes;
    hipLaunchKernelGGL(write_u32_kernel, dim3(1), dim3(1), 0, s, (uint32_t *)out, (uint32_t)prefix[world]);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(ctx, e, "launch write_u32");
    if (counts[rank] > 0) {
        e = hipMemcpyAsync(out + header_bytes + stride * prefix[rank], mine, stride * counts[rank],
                           hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return hip_fail(ctx, e, "hipMemcpyAsync(own list)");
    }
    if (world > 1) {
        ORBIT_NCCL(nc.group_start(), "ncclGroupStart");
        // an error inside the group must still close it (an open group would swallow the communicator's next calls)
        ncclResult_t bad = ncclSuccess;
        const char *bad_what = "";
        for (uint32_t peer = 0; peer < world && bad == ncclSuccess; peer++) {
            if (peer == rank) continue;
            if (counts[rank] > 0) {
                bad = nc.send(mine, stride * counts[rank], ncclUint8, (int)peer, comm, s);
                bad_what = "ncclSend";
            }
            if (bad == ncclSuccess && counts[peer] > 0) {
                bad = nc.recv(out + header_bytes + stride * prefix[peer], stride * counts[peer], ncclUint8, (int)peer,
                              comm, s);
                bad_what = "ncclRecv";
            }
        }
        const ncclResult_t ended = nc.group_end();
        if (bad != ncclSuccess) return fail(ctx, ORBIT_E_COMM, "%s: %s", bad_what, nc.error_string(bad));
        ORBIT_NCCL(ended, "ncclGroupEnd");
    }
#undef ORBIT_NCCL
    return ORBIT_OK;
}

} // namespace

extern "C" {

// ------------------------------------------------------------------- exchange without a host round trip
int32_t orbit_p2p_alloc(OrbitCtx *ctx, uint64_t bytes, void **out_ptr, uint8_t out_handle[ORBIT_P2P_HANDLE_BYTES]) {
    static_assert(sizeof(hipIpcMemHandle_t) == ORBIT_P2P_HANDLE_BYTES, "handle size");
    if (!ctx || !out_ptr || !out_handle || bytes == 0) return fail(ctx, ORBIT_E_INVALID, "p2p_alloc: bad argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    *out_ptr = nullptr;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipSetDevice");
    // Fine-grained device memory: what a peer GPU stores into it over xGMI — the list segments, and the counts and
    // completion words this device's waiting kernel polls while it runs — must be visible to this device without a
    // kernel boundary in between.  Ordinary (coarse-grained) device memory is only coherent between devices at kernel
    // boundaries: the device's L2 may keep serving a polled control word it cached before the peer's store arrived.
    void *ptr = nullptr;
    e = hipExtMallocWithFlags(&ptr, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipExtMallocWithFlags(p2p buffer, fine-grained)");
    e = memset_now(ptr, 0, bytes);
    if (e == hipSuccess) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t *>(out_handle), ptr);
    if (e != hipSuccess) {
        (void)hipFree(ptr);
        return hip_fail(ctx, e, "hipIpcGetMemHandle (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
    }
    *out_ptr = ptr;
    return ORBIT_OK;
}

int32_t orbit_p2p_free(OrbitCtx *ctx, void *ptr) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ptr) return ORBIT_OK;
    const hipError_t e = hipFree(ptr);
    return e == hipSuccess ? ORBIT_OK : hip_fail(ctx, e, "hipFree(p2p buffer)");
}

int32_t orbit_p2p_open(OrbitCtx *ctx, const uint8_t handle[ORBIT_P2P_HANDLE_BYTES], void **out_peer_ptr) {
    if (!ctx || !handle || !out_peer_ptr) return fail(ctx, ORBIT_E_INVALID, "p2p_open: bad argument");
    std::lock_guard<std::mutex> lock(ctx->mu);
    *out_peer_ptr = nullptr;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipSetDevice");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void *ptr = nullptr;
    e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipIpcOpenMemHandle");
    *out_peer_ptr = ptr;
    return ORBIT_OK;
}

int32_t orbit_p2p_close(OrbitCtx *ctx, void *peer_ptr) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!peer_ptr) return ORBIT_OK;
    const hipError_t e = hipIpcCloseMemHandle(peer_ptr);
    return e == hipSuccess ? ORBIT_OK : hip_fail(ctx, e, "hipIpcCloseMemHandle");
}

int32_t orbit_exchange_list(OrbitCtx *ctx, const void *local_list, uint32_t rank, uint32_t world,
                            void *const *out_buffers, void *const *ctrl_buffers, uint32_t out_capacity,
                            uint32_t header_bytes, uint32_t stride, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!local_list || !out_buffers || !ctrl_buffers) return fail(ctx, ORBIT_E_MISSING, "exchange_list: NULL argument");
    if (world == 0 || world > ORBIT_P2P_MAX_WORLD || rank >= world)
        return fail(ctx, ORBIT_E_INVALID, "exchange_list: rank %u of world %u (max %u)", rank, world,
                    (unsigned)ORBIT_P2P_MAX_WORLD);
    if (header_bytes < 4 || header_bytes % 4u || stride == 0 || stride % 4u)
        return fail(ctx, ORBIT_E_INVALID, "exchange_list: header %u / stride %u must be multiples of 4", header_bytes,
                    stride);
    ExchangeListParams p{};
    p.local_list = (const uint8_t *)local_list;
    for (uint32_t r = 0; r < world; r++) {
        if (!out_buffers[r] || !ctrl_buffers[r]) return fail(ctx, ORBIT_E_MISSING, "exchange_list: buffer %u is NULL", r);
        p.out[r] = (uint8_t *)out_buffers[r];
        p.ctrl[r] = (uint8_t *)ctrl_buffers[r];
    }
    p.rank = rank;
    p.world = world;
    p.out_capacity = out_capacity;
    p.header_bytes = header_bytes;
    p.stride = stride;
    p.status = ctx->status;
    const hipError_t e = launch_exchange_list(p, ctx->num_cus, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch exchange_list");
    return ORBIT_OK;
}

static int32_t compact_segments_locked(OrbitCtx *ctx, const void *segments, uint32_t world, uint32_t segment_capacity,
                                       void *out_list, uint32_t out_capacity, uint32_t header_bytes, uint32_t stride,
                                       void *stream) {
    if (!segments || !out_list) return fail(ctx, ORBIT_E_MISSING, "compact_segments: NULL argument");
    if (world == 0 || world > kMaxGatherWorld)
        return fail(ctx, ORBIT_E_INVALID, "compact_segments: world %u (max %u)", world, kMaxGatherWorld);
    if (header_bytes < 4 || header_bytes % 4u || stride == 0 || stride % 4u)
        return fail(ctx, ORBIT_E_INVALID, "compact_segments: header %u / stride %u must be multiples of 4", header_bytes,
                    stride);
    const hipError_t e = launch_compact_segments((const uint8_t *)segments, world, segment_capacity, (uint8_t *)out_list,
                                                 out_capacity, header_bytes, stride, ctx->num_cus, ctx->status,
                                                 (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(ctx, e, "launch compact_segments");
    return ORBIT_OK;
}

int32_t orbit_compact_segments(OrbitCtx *ctx, const void *segments, uint32_t world, uint32_t segment_capacity,
                               void *out_list, uint32_t out_capacity, uint32_t header_bytes, uint32_t stride,
                               void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    return compact_segments_locked(ctx, segments, world, segment_capacity, out_list, out_capacity, header_bytes, stride,
                                   stream);
}

int32_t orbit_allgather_list(OrbitCtx *ctx, void *nccl_comm, uint32_t rank, uint32_t world, const void *local_list,
                             uint32_t segment_capacity, void *segments, void *out_list, uint32_t out_capacity,
                             uint32_t header_bytes, uint32_t stride, void *stream) {
    if (!ctx) return fail(nullptr, ORBIT_E_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!nccl_comm || !local_list || !segments || !out_list)
        return fail(ctx, ORBIT_E_MISSING, "allgather_list: NULL argument");
    if (world == 0 || world > kMaxGatherWorld || rank >= world)
        return fail(ctx, ORBIT_E_INVALID, "allgather_list: rank %u of world %u (max %u)", rank, world, kMaxGatherWorld);
    if (header_bytes < 4 || header_bytes % 4u || stride == 0 || stride % 4u)
        return fail(ctx, ORBIT_E_INVALID, "allgather_list: header %u / stride %u must be multiples of 4", header_bytes,
                    stride);
    const Rccl &nc = rccl();
    if (!nc.ok) return fail(ctx, ORBIT_E_COMM, "allgather_list: librccl is not loadable in this process");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return hip_fail(ctx, e, "hipSetDevice");
    // ONE collective of a fixed size: every rank's whole segment {count | header | segment_capacity items}.  Nothing is
    // read back, nothing waits: the counts stay on the device, where the compaction reads them.
    const size_t seg_bytes = (size_t)header_bytes + (size_t)stride * segment_capacity;
    const ncclResult_t r = nc.all_gather(local_list, segments, seg_bytes, ncclUint8, (ncclComm_t)nccl_comm, (hipStream_t)stream);
    if (r != ncclSuccess) return fail(ctx, ORBIT_E_COMM, "ncclAllGather(list segments): %s", nc.error_string(r));
    return compact_segments_locked(ctx, segments, world, segment_capacity, out_list, out_capacity, header_bytes, stride,
                                   stream);
}

int32_t orbit_gather_visible(OrbitCtx *ctx, void *nccl_comm, uint32_t rank, uint32_t world,
                             const void *local_draw_buffer, void *out_draw_buffer, uint32_t out_capacity,
                             void *stream) {
    return gather_lists(ctx, nccl_comm, rank, world, local_draw_buffer, out_draw_buffer, out_capacity, stream,
                        ORBIT_DRAW_HEADER, sizeof(OrbitMeshletDrawCommand));
}

} // extern "C"
