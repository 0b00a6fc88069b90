/*
 * orbit_abi_ext.h — entry points of the engine BEYOND the drop-in core of orbit_abi.h.
 *
 * orbit_abi.h holds what a maintainer needs to swap the bodies of the reference's cull passes
 * (src/passes/draw_gen.rs:239-435,456-566, src/passes/cluster.rs:368-591): the context, entity_cull,
 * meshlet_cull, depth_reduce, the three light-cluster stages, the shard range and the RCCL gather.  Everything here is
 * optional on top of that — same conventions, same status codes, same context:
 *   - the mesh-shading path (task-shader records instead of draw commands),
 *   - several views in one call / one launch (orbit_cull_views),
 *   - derived meshlet streams (an MI355X-side re-layout of the static meshlet buffer),
 *   - several pyramids in one launch,
 *   - the late half of a frame with its independent chains side by side (orbit_frame_late),
 *   - the sharded engine: the shard cull, the record list, its device-side exchanges (HIP-IPC stores; RCCL all-gather)
 *     and the expansion of a gathered list,
 *   - the measurement hook of bench.py.
 * A build that only needs the drop-in includes orbit_abi.h alone; liborbit_cull.so exports both sets.
 */
#ifndef ORBIT_ABI_EXT_H
#define ORBIT_ABI_EXT_H

#include "orbit_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The visible list of the sharded engine (no reference counterpart: the reference is single-GPU) — the ONE format a rank
 * sends: the ordered survivor list at RECORD granularity.  Buffer layout: {u32 records; u32 survivors} @0, entries[] @8,
 * one entry per dispatch record, in record order; bit b of `mask` set = meshlet `meshlet_offset + b` of entity
 * `entity_index` is drawn (mask 0: a record without a survivor).  12 B per record: never more than 3/8 of a byte per
 * dispatched meshlet on the xGMI links (28-B commands: 66 MB over one link at N = 2 of BASELINE config 5, slower than
 * not sharding) — and an entry's position is its record's index, so the evaluation launch writes the list itself: no
 * count, no scan, no launch behind it.  (Rounds 2-5 also carried an 8-B {meshlet, entity} item per survivor; retired
 * in round 6 — twice the bytes at config 5 and an emit launch to make it.) */
#define ORBIT_VISIBLE_HEADER 8u
typedef struct OrbitVisibleRecord {
    uint32_t entity_index;
    uint32_t meshlet_offset;
    uint32_t mask; /* should-draw ballot of the record's (up to) 32 meshlets (meshlet_cull.comp:207-213) */
} OrbitVisibleRecord;
ORBIT_STATIC_ASSERT(sizeof(OrbitVisibleRecord) == 12, "visible record is 12 B");

/* Mesh-shading path.  When mesh shading is on, the renderer
 * skips create_meshlet_draw_commands and culls meshlets in the task shader,
 * one workgroup per MeshletDispatch record
 * (shaders/forward/forward_depth_prepass.task:115-257, forward.task, fed by the
 * dispatch buffer of create_meshlet_dispatch_command).  This entry point
 * computes what those workgroups compute — per record the EmitMeshTasksEXT
 * count and the MeshTaskPayload — into `task_records` (dispatch_capacity
 * entries, dispatch order), so that a task shader only has to load its record.
 * Same decision as orbit_meshlet_cull; as in the task shaders, lanes past a
 * record's meshlet_count report `visible` in the pass-2 visibility words.
 * bufs->draw_commands_buffer / draw_capacity are not used (may be NULL / 0). */
int32_t orbit_meshlet_task_cull(OrbitCtx *ctx, const OrbitGpuCullInfo *cull_info,
                                const OrbitMeshletCullBufs *bufs,
                                OrbitMeshTaskRecord *task_records, void *stream);

/* ------------------------------------------------------------------------ */
/* Several views in one call.  A frame of the reference culls the same scene */
/* for the early forward pass and for each of the four shadow cascades        */
/* (forward.rs:286-403, shadow_renderer.rs:391-403) — independent culls,      */
/* each a chain of five short, latency-bound launches on a scene of a few     */
/* thousand entities.  orbit_cull_views puts them into ONE chain: every       */
/* kernel of the path takes all views at once (blockIdx.y = view, parameter   */
/* blocks from a table; the streaming evaluations are grouped by occlusion    */
/* pass and projection type, so the early pass + four cascades are two        */
/* launches), each view on its own scan scratch — 6 launches instead of 25.   */
/* Every view's outputs are exactly those of orbit_entity_cull +              */
/* orbit_meshlet_cull on its buffers.  Meant for scenes whose culls are bound */
/* by launch latency; a 50 M-meshlet view belongs in the single-view calls    */
/* (their kernels keep the parameters in scalar registers).                   */
/* ------------------------------------------------------------------------ */
typedef struct OrbitCullView {
    const OrbitGpuCullInfo *cull_info; /* HOST, 400 B: this view's CullInfo::to_gpu */
    OrbitEntityCullBufs entity;        /* its own dispatch buffer (and bitset / pyramid in the occlusion passes) */
    OrbitMeshletCullBufs meshlet;      /* meshlet_dispatch_buffer = entity.meshlet_dispatch_buffer */
    uint32_t entity_draw_count;
    uint32_t skip_meshlet_stage;       /* != 0: mesh-shading path, only the dispatch records are produced */
} OrbitCullView;
#define ORBIT_MAX_CULL_VIEWS 8
/* The views must not share output buffers; visibility buffers written by two views (occlusion_pass 2) must differ. */
int32_t orbit_cull_views(OrbitCtx *ctx, const OrbitCullView *views, uint32_t count, void *stream);
/* One launch instead of the chain.  A call whose views all have both stages and between 1 and 16 384 entity-draws
 * each — every scene the reference itself can hold: MAX_INSTANCE_COUNT = 100 000 instances, MAX_MESHLET_COUNT = 256 k
 * (src/scene.rs:303, src/assets/mod.rs:202), a cull being one clear and two dispatches there (draw_gen.rs:283-322) —
 * runs as ONE persistent launch per (occlusion pass, projection type) among its views: entity test, record append,
 * meshlet test and ordered command append hand over inside the launch (orbit_amd/csrc/cull_fused.hip), which is what
 * such a cull costs: its dependent memory round trips once, not once per launch.  count = 1 is the single cull
 * (create_draw_commands, draw_gen.rs:239-322).  The outputs are those of the chain bit for bit; the path reads the 32-B
 * Meshlet buffer (a bound stream is not consulted) and can be captured into a graph like every other call.
 * OrbitCaps.cull_path overrides the choice.  Returns how many views this context has culled that way. */
uint64_t orbit_ctx_fused_culls(const OrbitCtx *ctx);

/* ------------------------------------------------------------------------ */
/* Derived meshlet streams — an MI355X-side copy of the static meshlet       */
/* buffer in the layout the cull streams best.                               */
/*                                                                           */
/* GpuAssets::add_mesh writes a mesh's Meshlet records into meshlet_buffer   */
/* once (src/assets/mod.rs:441-445); every frame's meshlet cull then streams */
/* all 32 B of every dispatched meshlet, although the decision reads 22 of   */
/* them (bounding sphere, cone, material index) — 20 and a quarter once the  */
/* material's alpha_mode is known — and only the ~10 % that survive need the */
/* rest (vertex_offset, data_offset, counts) for their command.  A stream    */
/* object keeps derived arrays for a range of the buffer (36.65 B of HBM per */
/* meshlet on top of the buffer itself):                                     */
/*   spheres 16 B, cones 4 B, material indices 2 B — what every meshlet's    */
/*     test reads;                                                           */
/*   alpha classes, 2 bits — the material's alpha_mode, so that the test     */
/*     reads no material index (set_materials below);                        */
/*   command words 12 B — gathered per survivor by                           */
/*     orbit_expand_visible_records;                                         */
/*   counts 2 B + 1 link bit — the command chain: compute_meshlets /         */
/*     add_mesh lay a mesh's meshlets out so that a meshlet's data_offset is */
/*     its predecessor's plus the predecessor's vertex_count +               */
/*     ceil(3 * triangle_count / 4) words, with one vertex_offset per        */
/*     submesh (src/assets/mesh.rs:309-316, assets/mod.rs:413-416); the link */
/*     bit records where that holds, and the emit launch then derives a      */
/*     survivor's offsets from one gathered base per chain and the counts    */
/*     streamed at 2 B per meshlet instead of gathering 12 B per survivor.   */
/*     Meshlets laid out any other way are served by gathers: the chain is   */
/*     an acceleration, never an assumption.                                 */
/* With a stream bound to a context, orbit_meshlet_cull / _task_cull /       */
/* _cull_visible_records / orbit_cull_shard calls whose bufs->meshlet_buffer */
/* is the pointer the stream was last updated from evaluate occlusion passes */
/* 0 and 2 from the arrays and write no survivor payload, and                */
/* orbit_expand_visible_records gathers 12 B instead of touching the         */
/* survivors' Meshlets.  Results are bit-identical                           */
/* to the plain path (every test of tests/test_gpu_parity.py runs all ways). */
/* orbit_cull_views applies the same rule per view.  Pass 1 (list-driven     */
/* gathers) and any call with another meshlet_buffer pointer read the        */
/* Meshlet buffer as before.                                                 */
/*                                                                           */
/*   create   arrays for global meshlet indices [first_meshlet,              */
/*            first_meshlet + capacity), capacity <= 2^27 (device memory:    */
/*            36.65 B x capacity, zero-filled)                               */
/*   update   re-derives [first, first + count) from `meshlet_buffer` (same  */
/*            global indexing as bufs->meshlet_buffer), enqueued on `stream`;*/
/*            call it wherever the renderer writes meshlets (add_mesh) —     */
/*            meshlets changed without an update are culled from stale data  */
/*            (caps.validate_streams / orbit_meshlet_stream_validate find    */
/*            that).  Culls read the HULL of the ranges updated from the     */
/*            same buffer pointer; meshlets inside it that no update reached */
/*            are empty (zero) meshlets.  An update from another pointer     */
/*            starts the range over.                                         */
/*   set_materials  derives every meshlet's alpha class from                 */
/*            material_buffer[material_index].alpha_mode (material_count     */
/*            entries; indices beyond it and modes > 2 are looked up at cull */
/*            time as before) and remembers the pointer: later updates       */
/*            derive the classes of their range from it, and culls whose     */
/*            bufs->material_buffer is that pointer read the classes instead */
/*            of the material indices — provided NO meshlet of the stream    */
/*            needs the look-up; a stream that holds one keeps evaluating    */
/*            through the indices.  Call it wherever the renderer writes     */
/*            materials (add_material, assets/mod.rs:520); NULL forgets.     */
/*            An upload-time call: unless `stream` is being captured into a  */
/*            graph it returns once the derivation has run (it waits for its */
/*            own launch on `stream`), so that which evaluation kernel the   */
/*            next cull takes never depends on timing.  The pointer is       */
/*            dereferenced on the device by every later update: call         */
/*            set_materials(NULL) (or with the new buffer) BEFORE the         */
/*            material buffer is freed or re-created.                        */
/*   validate compares the stream with `meshlet_buffer` (and the classes     */
/*            with `material_buffer`, may be NULL) over the readable range   */
/*            and latches ORBIT_E_STALE in the context's status on any       */
/*            difference.                                                    */
/*   bind     NULL unbinds; one stream can be bound to any number of         */
/*            contexts of its device, and cannot be destroyed while bound    */
/*            (ORBIT_E_INVALID; destroying a context unbinds).  The stream   */
/*            must cover every meshlet the culls dispatch: a meshlet outside */
/*            its range is not read from it (the lane evaluates zeros, the   */
/*            command carries zeros) and ORBIT_E_RANGE is latched            */
/* ------------------------------------------------------------------------ */
typedef struct OrbitMeshletStream OrbitMeshletStream;
int32_t orbit_meshlet_stream_create(OrbitCtx *ctx, uint64_t first_meshlet, uint64_t capacity,
                                    OrbitMeshletStream **out_stream);
int32_t orbit_meshlet_stream_update(OrbitCtx *ctx, OrbitMeshletStream *ms, const void *meshlet_buffer,
                                    uint64_t first, uint64_t count, void *stream);
int32_t orbit_meshlet_stream_set_materials(OrbitCtx *ctx, OrbitMeshletStream *ms, const void *material_buffer,
                                           uint32_t material_count, void *stream);
int32_t orbit_meshlet_stream_validate(OrbitCtx *ctx, OrbitMeshletStream *ms, const void *meshlet_buffer,
                                      const void *material_buffer, void *stream);
/* The entity stage's share of the same idea.  GpuAssets::add_mesh also writes the mesh's 128-B MeshInfo
 * (src/assets/mod.rs:18-28, types.glsl:123-141) once, and every frame's entity cull then fetches that whole line per
 * entity-draw for 28 bytes of it: the bounding sphere, lod_count and the chosen MeshLod.  update_meshes derives a 32-B
 * side entry {sphere, lod_count, mesh_lods[0]} for meshes [first_mesh, first_mesh + count) of `mesh_info_buffer`
 * (32 B of device memory per mesh; the table — at least 2^20 entries, zero-filled — is allocated by the first call);
 * entity culls (orbit_entity_cull[_range], orbit_cull_views, orbit_cull_shard) of contexts the stream is bound to whose
 * bufs->mesh_info_buffer is that pointer read the entry instead of the MeshInfo, and the MeshInfo itself only where a
 * LOD other than 0 is picked or where no update reached the mesh.  Same results bit for bit.  Call it wherever the
 * renderer writes mesh infos; an update from another buffer pointer starts the table over, NULL forgets it;
 * orbit_meshlet_stream_validate (and every entity cull of a context with validate_streams) compares the table with the
 * buffer it was derived from (ORBIT_E_STALE).  orbit_ctx_mesh_side_culls: entity culls that were handed the table. */
int32_t orbit_meshlet_stream_update_meshes(OrbitCtx *ctx, OrbitMeshletStream *ms, const void *mesh_info_buffer,
                                           uint32_t first_mesh, uint32_t count, void *stream);
uint64_t orbit_ctx_mesh_side_culls(const OrbitCtx *ctx);
int32_t orbit_meshlet_stream_destroy(OrbitMeshletStream *ms);
int32_t orbit_ctx_bind_meshlet_stream(OrbitCtx *ctx, OrbitMeshletStream *ms);
/* Calls this context has served from a bound stream so far — meshlet culls of passes 0 and 2 (each such view of
 * orbit_cull_views' launch chain counts) and orbit_expand_visible_records (tests and integration checks: a call whose
 * meshlet_buffer is not the stream's source silently takes the plain path).  Views that orbit_cull_views runs as its
 * ONE launch (up to 16 384 entity-draws, OrbitCaps.cull_path) read the 32-B Meshlet buffer and do not count —
 * orbit_ctx_fused_culls counts those — and neither ORBIT_E_RANGE nor the validate_streams check applies to them
 * (a context created with validate_streams keeps the chain for that reason). */
uint64_t orbit_ctx_meshlet_stream_culls(const OrbitCtx *ctx);
/* ... of which evaluated from the alpha classes (no material index read): culls whose material buffer the classes
 * mirror, of a stream known to hold no meshlet of class 3 — known when set_materials returns (it waits for its
 * derivation unless it is being captured); after an update that derives classes, once that launch has finished
 * (polled, never waited for: culls enqueued before that read the material indices). */
uint64_t orbit_ctx_meshlet_class_culls(const OrbitCtx *ctx);

/* One pyramid of a batch.  `depth_row_pitch` = texels per row of the depth buffer (0 = screen_width: tightly
 * packed); exactly one of `pyramid` (packed chain) and `levels` (HOST array of the mip_levels the pyramid has for
 * this screen size, orbit_depth_pyramid_desc; each entry holds DEVICE pointers) is non-NULL. */
typedef struct OrbitDepthReduceItem {
    const float *depth;
    uint32_t screen_width, screen_height;
    uint32_t depth_row_pitch;
    uint32_t _pad;
    float *pyramid;
    const OrbitDepthPyramidLevel *levels;
} OrbitDepthReduceItem;
ORBIT_STATIC_ASSERT(sizeof(OrbitDepthReduceItem) == 40, "depth-reduce item is 40 B");
#define ORBIT_MAX_PYRAMID_BATCH 8
/* update_multiple_depth_pyramids::<C> (draw_gen.rs:569-628): `count` (<= 8) pyramids — the main view's and the
 * shadow cascades' — in ONE launch pair instead of C x 12 dispatches; pyramids of different sizes may be mixed.
 * Also the entry point for pyramids made of separate per-mip images (`levels`) and pitched depth buffers. */
int32_t orbit_depth_reduce_multi(OrbitCtx *ctx, const OrbitDepthReduceItem *items, uint32_t count, void *stream);

/* ------------------------------------------------------------------------ */
/* The late half of a frame as ONE call with its independent chains side by  */
/* side.  Between "the depth buffer exists" and the forward pass the renderer */
/* records (src/app.rs:1151-1212): the depth pyramid(s) and the late          */
/* (VisibilityWrite) culls of render_depth_prepass (forward.rs:371-403,      */
/* draw_gen.rs:510-566), render_shadows' cascade culls (shadow_renderer.rs:  */
/* 391-403: occlusion_pass 0, no pyramid) and compute_clusters (cluster.rs:  */
/* 368-397).  Three chains that share no output: on scenes of the            */
/* reference's own size each is a handful of dependent, latency-bound        */
/* launches on a device that is 95 % idle during any one of them — issued    */
/* one after the other on one stream they cost their sum.                    */
/* orbit_frame_late runs                                                     */
/*   A  pyramids (orbit_depth_reduce_multi) -> late_views (orbit_cull_views), */
/*   B  cascade_views (orbit_cull_views) and                                  */
/*   C  clusters (orbit_compute_clusters)                                     */
/* side by side: the chain with the most dependent launches stays on         */
/* `stream`, the other two run on two streams of the context, forked behind  */
/* whatever `stream` holds when                                              */
/* the call is made and joined into `stream` before it returns — events only, */
/* no host wait: the call can be captured into a graph, and work enqueued    */
/* on `stream` behind it sees all three chains' outputs.  Outputs are byte   */
/* for byte those of the serial calls.  Every group is optional (count 0 /   */
/* NULL); late + cascade views together at most ORBIT_MAX_CULL_VIEWS (each   */
/* view on its own scan scratch, OrbitCaps.max_views); everything is         */
/* validated before anything is enqueued.  The chains' inputs must be        */
/* complete in `stream` order when the call is made (the depth buffer, last  */
/* frame's visibility words); their output buffers must differ.  The two     */
/* side streams are created by the first call that needs them.               */
/* ------------------------------------------------------------------------ */
typedef struct OrbitClusterFrame { /* orbit_compute_clusters' arguments */
    const OrbitMarkActivePush *push;
    const OrbitClusterCullInfo *info;
    const float *depth;
    const OrbitLightData *lights;
    uint32_t *tile_depth_slice_mask;
    OrbitClusterDepthBounds *depth_bounds;
    void *unique_cluster_buffer;
    void *light_index_buffer;
    uint32_t *cluster_offset_image;
    uint32_t index_capacity;
    uint32_t light_index_capacity;
} OrbitClusterFrame;
ORBIT_STATIC_ASSERT(sizeof(OrbitClusterFrame) == 80, "cluster frame block is 80 B");
typedef struct OrbitFrameLate {
    const OrbitDepthReduceItem *pyramids; /* chain A, first: the late views' pyramids (NULL / 0: none) */
    const OrbitCullView *late_views;      /* chain A, then: the VisibilityWrite culls that read them */
    const OrbitCullView *cascade_views;   /* chain B */
    const OrbitClusterFrame *clusters;    /* chain C (NULL: none) */
    uint32_t pyramid_count, late_view_count, cascade_view_count, _pad;
} OrbitFrameLate;
ORBIT_STATIC_ASSERT(sizeof(OrbitFrameLate) == 48, "frame-late descriptor is 48 B");
int32_t orbit_frame_late(OrbitCtx *ctx, const OrbitFrameLate *frame, void *stream);

/* ------------------------------------------------------------------------ */
/* Measurement hook (bench.py).  While enabled, every orbit_meshlet_cull     */
/* records a HIP event pair on the caller's stream around the op's dominant  */
/* kernel (the streaming meshlet evaluation); profile_read waits for the     */
/* recorded pairs and returns their mean duration.  Never on by default.     */
/* `enable` = n > 0 times every n-th call (an event pair costs a few          */
/* microseconds of stream time: it keeps the next launch from being          */
/* prefetched), 0 switches the hook off.                                     */
/* ------------------------------------------------------------------------ */
int32_t orbit_ctx_profile(OrbitCtx *ctx, int32_t enable);
/* The pairs are created on first use (a hipEventCreate and a first record each: host time inside the region that is
 * being timed); _reserve creates `pairs` of them up front, each recorded once on `stream`. */
int32_t orbit_ctx_profile_reserve(OrbitCtx *ctx, uint32_t pairs, void *stream);
int32_t orbit_ctx_profile_read(OrbitCtx *ctx, float *avg_ms, uint32_t *launches);

/* ------------------------------------------------------------------------ */
/* The sharded engine — ONE product.  Per rank and frame:    */
/*   1. orbit_cull_shard: the rank's entity range through both stages; ends  */
/*      in the record list of its shard (what it sends) and, optionally, the */
/*      28-B MeshletDrawCommandBuffer of its shard (what it draws from);     */
/*   2. the rank-ordered all-gather of the record lists — every rank's list  */
/*      is in canonical order and the shards are contiguous in entity order, */
/*      so the concatenation in rank order IS the single-GPU sequence:       */
/*      orbit_exchange_list (direct xGMI stores into the peers' IPC-mapped   */
/*      buffers, counts and completion signalled on the device; default) or  */
/*      orbit_allgather_list (north_star's RCCL all-gather; the fallback);   */
/*      neither reads a count on the host, both are capturable;              */
/*   3. where a GPU wants the WHOLE scene's commands: orbit_expand_visible_   */
/*      records of the gathered list (needs the meshlet buffer there).       */
/*                                                                           */
/*   orbit_meshlet_cull_visible_records  orbit_meshlet_cull, but              */
/*       `record_buffer` ({records, survivors} @0, OrbitVisibleRecord[] @8,  */
/*       record_capacity entries) replaces bufs->draw_commands_buffer /      */
/*       draw_capacity (not used).  Entry i is dispatch record i with its    */
/*       should-draw ballot, written by the evaluation launch itself (ONE    */
/*       launch for the whole cull; the header by its last workgroup):       */
/*       records = the dispatched records, survivors = the set bits of all   */
/*       of them.  ORBIT_E_CAPACITY is latched if the records do not fit     */
/*       (the header holds the clamped count, the first record_capacity      */
/*       entries are written);                                               */
/*   orbit_expand_visible_records  record list -> MeshletDrawCommandBuffer   */
/*       in list order (= the canonical order), the command words read from  */
/*       `meshlet_buffer` under global indices (or from a bound stream that  */
/*       mirrors it); at most 64 M records per list (the gathered header's   */
/*       second word is not maintained by the exchanges and not read here).  */
/* ------------------------------------------------------------------------ */
int32_t orbit_meshlet_cull_visible_records(OrbitCtx *ctx, const OrbitGpuCullInfo *cull_info,
                                           const OrbitMeshletCullBufs *bufs, void *record_buffer,
                                           uint32_t record_capacity, void *stream);
int32_t orbit_expand_visible_records(OrbitCtx *ctx, const void *record_buffer, const void *meshlet_buffer,
                                     void *draw_commands_buffer, uint32_t draw_capacity, void *stream);
/* Both products of ONE evaluation: the record list (as orbit_meshlet_cull_visible_records: what the rank sends) and the
 * rank's own MeshletDrawCommandBuffer in bufs->draw_commands_buffer / draw_capacity (as orbit_meshlet_cull: what it
 * draws from, "each rank keeps its shard and issues its own indirect draws") — the evaluation writes the
 * list, scan + emit of the same ballots follow; cheaper than the list followed by orbit_expand_visible_records of it. */
int32_t orbit_meshlet_cull_records_and_commands(OrbitCtx *ctx, const OrbitGpuCullInfo *cull_info,
                                                const OrbitMeshletCullBufs *bufs, void *record_buffer,
                                                uint32_t record_capacity, void *stream);
/* A rank's whole cull as ONE call — orbit_entity_cull_range over [draw_first, draw_first + draw_count) followed by
 * orbit_meshlet_cull_visible_records (with_commands == 0) or orbit_meshlet_cull_records_and_commands (!= 0) on the
 * records it appended (meshlet_bufs->meshlet_dispatch_buffer must be entity_bufs->meshlet_dispatch_buffer) — and, for
 * occlusion pass 0 and at most 65 536 entity-draws, ONE launch for the entity test, the record append, the meshlet
 * test and the record list (+ the emit launch when the commands are wanted): a shard's step is mostly its launches'
 * fixed costs, and this is what a 1/8 shard of BASELINE config 5 pays them once for instead of four times.  Other
 * passes and larger ranges run the two calls' launch chain; the outputs are the same bit for bit either way
 * (OrbitCaps.cull_path = 1 forces the chain).  orbit_ctx_shard_culls: calls that took the one launch. */
int32_t orbit_cull_shard(OrbitCtx *ctx, const OrbitGpuCullInfo *cull_info, const OrbitEntityCullBufs *entity_bufs,
                         uint32_t draw_first, uint32_t draw_count, const OrbitMeshletCullBufs *meshlet_bufs,
                         void *record_buffer, uint32_t record_capacity, uint32_t with_commands, void *stream);
uint64_t orbit_ctx_shard_culls(const OrbitCtx *ctx);

/* ------------------------------------------------------------------------ */
/* The exchange by direct stores (one node, one process per GPU, peers       */
/* reachable over xGMI): the bulk data does not go through RCCL at all —     */
/* every rank copies its list straight into every peer's output buffer at    */
/* the rank-ordered position, the offsets computed on the device from counts */
/* that travelled the same way.  Nothing waits for the host.                 */
/* ------------------------------------------------------------------------ */

#define ORBIT_P2P_HANDLE_BYTES 64 /* hipIpcMemHandle_t */
#define ORBIT_P2P_MAX_WORLD 16
#define ORBIT_P2P_CTRL_BYTES 1024 /* a rank's control block of orbit_exchange_list (orbit_p2p_alloc'ed, zero-filled) */

/* Exchange buffers: FINE-GRAINED device memory of this context's GPU that peers may map (hipExtMallocWithFlags,
 * hipDeviceMallocFinegrained: what a peer stores into it over xGMI — list segments, and the control words a waiting
 * kernel of this device polls while it runs — is visible here without a kernel boundary in between; ordinary device
 * memory is coherent between devices only at kernel boundaries).
 * orbit_p2p_alloc returns the pointer and an opaque handle to send to the other
 * processes (any host channel: a file, torch.distributed.all_gather_object);
 * orbit_p2p_open maps a peer's buffer into this process (hipIpcOpenMemHandle;
 * dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0).  The buffers stay caller-visible
 * for the life of the context; *_close / *_free undo the calls. */
int32_t orbit_p2p_alloc(OrbitCtx *ctx, uint64_t bytes, void **out_ptr,
                        uint8_t out_handle[ORBIT_P2P_HANDLE_BYTES]);
int32_t orbit_p2p_free(OrbitCtx *ctx, void *ptr);
int32_t orbit_p2p_open(OrbitCtx *ctx, const uint8_t handle[ORBIT_P2P_HANDLE_BYTES], void **out_peer_ptr);
int32_t orbit_p2p_close(OrbitCtx *ctx, void *peer_ptr);

/* The whole exchange on the device — no collective, no host in the step, capturable into a graph.
 *   local_list   {u32 count @0 | header_bytes | items of `stride` bytes}: a record list (header 8, stride 12); any
 *                header / stride that are multiples of 4 work (a MeshletDrawCommandBuffer: 4 / 28)
 *   out_buffers  HOST array of `world` device pointers: rank r's output buffer as mapped in THIS process
 *                (out_buffers[rank] = this rank's own buffer); same layout as local_list, `out_capacity` items
 * Counts and completion travel like the lists themselves, as stores into the peers' IPC-mapped memory: every rank
 * owns a CONTROL BLOCK (ORBIT_P2P_CTRL_BYTES from orbit_p2p_alloc, which
 * zero-fills it; mapped by every peer with orbit_p2p_open).  One call enqueues two launches on `stream`: a scatter that
 * first stores this rank's count into every peer's block, waits (on the device, bounded: ORBIT_E_TIMEOUT) for the counts
 * of the ranks before it, copies the list to its rank-ordered position in every rank's `out_buffers[r]` and then raises
 * "done" in that rank's block; and a one-workgroup launch that waits until every rank's count and "done" for this
 * exchange have arrived and writes the header {total, 0 ..} of this rank's own buffer.  Work enqueued behind the call
 * reads the complete list.  Exchanges are numbered by a counter in the control block (all ranks count in step), so a
 * captured graph replays correctly.  Every rank must call it the same number of times, with the same `world`; a
 * rank's out buffer must not be rewritten (by the next exchange into the same buffer) while its consumers still read
 * it: alternate two (buffer, control block) pairs, as bench.py does.
 *   ctrl_buffers  HOST array of `world` device pointers: rank r's control block as mapped in THIS process */
int32_t orbit_exchange_list(OrbitCtx *ctx, const void *local_list, uint32_t rank, uint32_t world,
                            void *const *out_buffers, void *const *ctrl_buffers, uint32_t out_capacity,
                            uint32_t header_bytes, uint32_t stride, void *stream);

/* ------------------------------------------------------------------------ */
/* north_star's transport — "an RCCL all-gather of the compacted visible    */
/* list" — with no host in it either.  RCCL send/recv take message sizes as  */
/* host arguments (orbit_gather_visible of orbit_abi.h reads them back); a   */
/* collective of a FIXED size needs no size: every rank contributes its     */
/* whole list buffer as a segment {count | header | segment_capacity items},*/
/* ONE ncclAllGather moves all segments to all ranks, and one launch         */
/* compacts them — the counts are read on the device — into the contiguous  */
/* rank-ordered list {total | header | items}: byte for byte what            */
/* orbit_exchange_list delivers.  Nothing is read back, no stream is         */
/* synchronised, both steps can be captured into a graph.  The price is the */
/* segments' slack on the links (capacity - count items per rank): nothing  */
/* for the record list, which holds an entry per dispatch record and is as  */
/* long as the shard's entity stage made it — the list this is meant for    */
/* (28-B commands or 8-B items sized for the worst case would move mostly    */
/* slack: those keep orbit_gather_visible*).                                 */
/*   local_list        this rank's list; the buffer holds at least           */
/*                     header_bytes + stride * segment_capacity bytes        */
/*   segment_capacity  items per segment, the same on every rank             */
/*   segments          world x (header_bytes + stride * segment_capacity)    */
/*                     bytes of device memory: the collective's receive      */
/*                     buffer (caller-owned like every buffer)               */
/*   out_list          {total | header | out_capacity items}; a total beyond */
/*                     out_capacity latches ORBIT_E_CAPACITY                 */
/* orbit_compact_segments is the second half alone, for callers that issue   */
/* the collective themselves (bench.py: torch.distributed's                  */
/* all_gather_into_tensor, which is ncclAllGather on ROCm).                  */
/* ------------------------------------------------------------------------ */
int32_t orbit_allgather_list(OrbitCtx *ctx, void *nccl_comm, uint32_t rank, uint32_t world, const void *local_list,
                             uint32_t segment_capacity, void *segments, void *out_list, uint32_t out_capacity,
                             uint32_t header_bytes, uint32_t stride, void *stream);
int32_t orbit_compact_segments(OrbitCtx *ctx, const void *segments, uint32_t world, uint32_t segment_capacity,
                               void *out_list, uint32_t out_capacity, uint32_t header_bytes, uint32_t stride,
                               void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ORBIT_ABI_EXT_H */
