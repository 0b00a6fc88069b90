/*
 * orbit_abi.h — C ABI of the MI355X-native culling / light-clustering engine.
 *
 * This header is the drop-in boundary.  The reference has no
 * FFI: its cull passes are Rust functions that record Vulkan compute work
 * (src/passes/draw_gen.rs, src/passes/cluster.rs).  Each entry point below
 * replaces the *body* of one of those functions; the Rust-side `extern "C"`
 * declarations a maintainer adds are listed in INTEGRATION.md and
 * bindings/rust/orbit_hip.rs.
 *
 * This header is the drop-in CORE: the reference's layouts and the 18 functions
 * that stand in for src/passes/draw_gen.rs and src/passes/cluster.rs
 * (SURVEY.md §8b).  What the engine offers beyond that — the mesh-shading
 * path, several views per call, derived meshlet streams, the sharded engine's
 * list formats and device-side exchanges, the measurement hook — is declared
 * in orbit_abi_ext.h, which includes this file.
 *
 * Conventions
 *  - plain C, no HIP / torch types: `void *stream` is a hipStream_t,
 *    every `*_buffer` pointer is a DEVICE pointer unless the comment says HOST.
 *  - all structs are byte-identical to the reference's std430 / repr(C)
 *    layouts (shaders/include/types.glsl); sizes and offsets are asserted.
 *  - every call only ENQUEUES work on `stream` (no hidden device sync, no
 *    allocation); ordering between calls is stream order, exactly like the
 *    render graph's barrier-separated batches (src/graphics/graph.rs:874-1050).
 *  - return value: 0 = ORBIT_OK, < 0 = ORBIT_E_*; text via orbit_last_error().
 *    Conditions only detectable on the device (an append past a caller
 *    buffer's capacity) are latched in the context and read with
 *    orbit_ctx_status().
 */
#ifndef ORBIT_ABI_H
#define ORBIT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#define ORBIT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define ORBIT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

#define ORBIT_ABI_VERSION 6u /* 2: *_CullBufs gained depth_pyramid_levels (per-mip images); 3: meshlet streams carry the
                                command chain and alpha classes, OrbitCaps.validate_streams, ORBIT_E_STALE;
                                4: OrbitCaps.cull_path, orbit_ctx_fused_culls (the one-launch cull of orbit_cull_views);
                                5: this header is the drop-in core, the rest moved to orbit_abi_ext.h; the record list holds an
                                entry per dispatch record (written by the evaluation launch); orbit_allgather_list,
                                orbit_compact_segments;
                                6: retired — the 8-B item list (orbit_meshlet_cull_visible, orbit_expand_visible,
                                orbit_gather_visible_meshlets) and orbit_scatter_list (the exchange signalled by two
                                collectives): the sharded engine is ONE product (orbit_abi_ext.h); OrbitCaps.dispatch_size
                                takes 32, 64 or 128 (pass 0 and the task path); orbit_frame_late */

/* u32::MAX = "resource absent" (src/passes/draw_gen.rs:142-144,
 * shaders/meshlet_cull.comp:127). */
#define ORBIT_NONE 0xFFFFFFFFu
#define ORBIT_MAX_CULL_PLANES 12 /* src/passes/draw_gen.rs:206 */
#define ORBIT_MAX_MESH_LODS 8    /* shaders/include/types.glsl:4 */
/* The only dispatch size for which the reference's visibility-word addressing
 * is self-consistent (src/graphics/device.rs:369-372). */
#define ORBIT_MESHLET_DISPATCH_SIZE 32u /* the default of OrbitCaps.dispatch_size */
#define ORBIT_MAX_LIGHTS_PER_CLUSTER 256u /* light_culling.comp:135 */

enum {
    ORBIT_OK = 0,
    ORBIT_E_INVALID = -1,  /* bad argument (NULL buffer, projection_type > 1, S != 32 ...) */
    ORBIT_E_PLANES = -2,   /* more than 12 cull planes: assert! at draw_gen.rs:247,334,390 */
    ORBIT_E_CAPACITY = -3, /* an append would overflow a caller buffer (latched on device) */
    ORBIT_E_HIP = -4,      /* a HIP runtime call failed */
    ORBIT_E_NO_DEVICE = -5,/* no gfx950 device / kernels not loadable: there is NO CPU fallback */
    ORBIT_E_TIMEOUT = -6,  /* a bounded in-kernel wait gave up (latched on device): the hand-overs inside the one-launch
                              cull / entity stage / chain emit, the device-signalled exchange waiting for a peer */
    ORBIT_E_MISSING = -7,  /* a resource the CullInfo declares present was passed as NULL
                              (.unwrap() at draw_gen.rs:123-133) */
    ORBIT_E_COMM = -8,     /* RCCL is not loadable in this process or one of its calls failed */
    ORBIT_E_RANGE = -9,    /* a culled / expanded meshlet lies outside the bound meshlet stream (latched on device) */
    ORBIT_E_STALE = -10    /* the bound meshlet stream no longer mirrors the meshlet buffer it was derived from
                              (orbit_meshlet_stream_validate, or any cull with caps.validate_streams; latched) */
};

/* ------------------------------------------------------------------------ */
/* Scene / asset input layouts                                               */
/* ------------------------------------------------------------------------ */

/* types.glsl:112-116, src/scene.rs:127-133 */
typedef struct OrbitEntityDraw {
    uint32_t entity_index;
    uint32_t mesh_index;
    uint32_t visibility_offset;
} OrbitEntityDraw;
ORBIT_STATIC_ASSERT(sizeof(OrbitEntityDraw) == 12, "EntityDraw is 12 B");
/* EntityDrawBuffer (types.glsl:118-121): u32 count @0, draws[] @4, stride 12. */
#define ORBIT_ENTITY_DRAW_HEADER 4u

/* types.glsl:75-78, src/scene.rs:120-125; matrices column-major */
typedef struct OrbitEntityData {
    float model_matrix[16];
    float normal_matrix[16];
} OrbitEntityData;
ORBIT_STATIC_ASSERT(sizeof(OrbitEntityData) == 128, "EntityData is 128 B");

typedef struct OrbitMeshLod {
    uint32_t meshlet_offset;
    uint32_t meshlet_count;
} OrbitMeshLod;

/* types.glsl:133-141, src/assets/mod.rs:18-28 */
typedef struct OrbitMeshInfo {
    float bounding_sphere[4];
    float aabb_min[4];
    float aabb_max[4];
    uint32_t vertex_offset;
    uint32_t meshlet_data_offset;
    uint32_t lod_count;
    uint32_t _padding;
    OrbitMeshLod mesh_lods[ORBIT_MAX_MESH_LODS];
} OrbitMeshInfo;
ORBIT_STATIC_ASSERT(sizeof(OrbitMeshInfo) == 128, "MeshInfo is 128 B");
ORBIT_STATIC_ASSERT(offsetof(OrbitMeshInfo, lod_count) == 56, "lod_count @56");
ORBIT_STATIC_ASSERT(offsetof(OrbitMeshInfo, mesh_lods) == 64, "mesh_lods @64");

/* types.glsl:143-152, src/assets/mod.rs:111-122 */
typedef struct OrbitMeshlet {
    float bounding_sphere[4];
    int8_t cone_axis[3];
    int8_t cone_cutoff;
    uint32_t vertex_offset;
    uint32_t data_offset;
    uint16_t material_index;
    uint8_t vertex_count;
    uint8_t triangle_count;
} OrbitMeshlet;
ORBIT_STATIC_ASSERT(sizeof(OrbitMeshlet) == 32, "Meshlet is 32 B");
ORBIT_STATIC_ASSERT(offsetof(OrbitMeshlet, vertex_offset) == 20, "vertex_offset @20");
ORBIT_STATIC_ASSERT(offsetof(OrbitMeshlet, material_index) == 28, "material_index @28");

/* types.glsl:92-110, src/assets/mod.rs:171-191 */
typedef struct OrbitMaterialData {
    float base_color[4];
    float emissive_factor[3];
    float metallic_factor;
    float roughness_factor;
    float occlusion_factor;
    float alpha_cutoff;
    uint32_t base_texture_index;
    uint32_t normal_texture_index;
    uint32_t metallic_roughness_texture_index;
    uint32_t occlusion_texture_index;
    uint32_t emissive_texture_index;
    uint32_t alpha_mode; /* 0 opaque, 1 masked, 2 transparent: assets/mod.rs:124-130 */
    uint32_t _padding[3];
} OrbitMaterialData;
ORBIT_STATIC_ASSERT(sizeof(OrbitMaterialData) == 80, "MaterialData is 80 B");
ORBIT_STATIC_ASSERT(offsetof(OrbitMaterialData, alpha_mode) == 64, "alpha_mode @64");

/* AlphaModeFlags, draw_gen.rs:630-641 */
#define ORBIT_ALPHA_OPAQUE 1u
#define ORBIT_ALPHA_MASKED 2u
#define ORBIT_ALPHA_TRANSPARENT 4u
#define ORBIT_ALPHA_ALL 7u

/* types.glsl:16-27, src/scene.rs:278-291 */
typedef struct OrbitLightData {
    uint32_t light_type; /* 0 sky, 1 directional, 2 point: types.glsl:298-300 */
    uint32_t shadow_data_index;
    uint32_t irradiance_map_index;
    uint32_t prefiltered_map_index;
    float color[3];
    float intensity;
    float position[3];
    float inner_radius;
    float direction[3];
    float outer_radius;
} OrbitLightData;
ORBIT_STATIC_ASSERT(sizeof(OrbitLightData) == 64, "LightData is 64 B");
ORBIT_STATIC_ASSERT(offsetof(OrbitLightData, position) == 32, "position @32");
ORBIT_STATIC_ASSERT(offsetof(OrbitLightData, outer_radius) == 60, "outer_radius @60");
#define ORBIT_LIGHT_TYPE_POINT 2u

/* ------------------------------------------------------------------------ */
/* Cull-pass output layouts                                                  */
/* ------------------------------------------------------------------------ */

/* types.glsl:166-171 */
typedef struct OrbitMeshletDispatch {
    uint32_t entity_index;
    uint32_t meshlet_offset;
    uint32_t meshlet_count;
    uint32_t visibility_offset;
} OrbitMeshletDispatch;
ORBIT_STATIC_ASSERT(sizeof(OrbitMeshletDispatch) == 16, "MeshletDispatch is 16 B");
/* MeshletDispatchBuffer (types.glsl:173-178): u32 workgroup_count_{x,y,z} @0,
 * dispatches[] @12, stride 16.  x = record count, y = z = 1. */
#define ORBIT_DISPATCH_HEADER 12u

/* types.glsl:180-189, src/assets/mod.rs:98-109.  The first 20 B are a
 * VkDrawIndexedIndirectCommand consumed with stride 28 (context.rs:1101-1108). */
typedef struct OrbitMeshletDrawCommand {
    uint32_t cmd_index_count;
    uint32_t cmd_instance_count;
    uint32_t cmd_first_index;
    int32_t cmd_vertex_offset;
    uint32_t cmd_first_instance;
    uint32_t meshlet_vertex_offset;
    uint32_t meshlet_index;
} OrbitMeshletDrawCommand;
ORBIT_STATIC_ASSERT(sizeof(OrbitMeshletDrawCommand) == 28, "MeshletDrawCommand is 28 B");

/* Mesh-shading path: what a task-shader workgroup hands to its mesh shaders —
 * MeshTaskPayload (types.glsl:196-200) and the EmitMeshTasksEXT count
 * (forward_depth_prepass.task:255-257) — one per MeshletDispatch record. */
typedef struct OrbitMeshTaskPayload {
    uint32_t entity_index;
    uint32_t meshlet_offset;
    uint8_t meshlet_indices[32]; /* lanes of the record that are drawn, ascending; 0 past the count */
} OrbitMeshTaskPayload;
typedef struct OrbitMeshTaskRecord {
    uint32_t task_mesh_count;
    OrbitMeshTaskPayload payload;
} OrbitMeshTaskRecord;
ORBIT_STATIC_ASSERT(sizeof(OrbitMeshTaskPayload) == 40, "MeshTaskPayload is 40 B");
ORBIT_STATIC_ASSERT(sizeof(OrbitMeshTaskRecord) == 44, "task record is 44 B");
/* MeshletDrawCommandBuffer (types.glsl:191-194): u32 count @0, draws[] @4. */
#define ORBIT_DRAW_HEADER 4u

/* GpuCullInfo, draw_gen.rs:208-237 == CullInfo, types.glsl:202-228 */
typedef struct OrbitGpuCullInfo {
    float view_matrix[16];         /* @0   */
    float reprojection_matrix[16]; /* @64  always zero, unused */
    float cull_planes[ORBIT_MAX_CULL_PLANES][4]; /* @128 view-space, normalised */
    uint32_t cull_plane_count;     /* @320 */
    uint32_t alpha_mode_flag;      /* @324 */
    uint32_t noskip_alphamode;     /* @328 */
    uint32_t occlusion_pass;       /* @332 0 none / 1 read / 2 write: draw_gen.rs:96-102 */
    uint32_t visibility_buffer;         /* @336 bindless index or ORBIT_NONE */
    uint32_t meshlet_visibility_buffer; /* @340 */
    uint32_t depth_pyramid;             /* @344 */
    uint32_t secondary_depth_pyramid;   /* @348 unused */
    uint32_t projection_type;      /* @352 0 perspective / 1 orthographic */
    float p00_or_width_recipx2;    /* @356 */
    float p11_or_height_recipx2;   /* @360 */
    float z_near;                  /* @364 */
    float z_far;                   /* @368 */
    float lod_base;                /* @372 */
    float lod_step;                /* @376 */
    uint32_t min_mesh_lod;         /* @380 */
    float lod_target_pos_view_space[3]; /* @384 */
    uint32_t max_mesh_lod;         /* @396 */
} OrbitGpuCullInfo;
ORBIT_STATIC_ASSERT(sizeof(OrbitGpuCullInfo) == 400, "GpuCullInfo is 400 B");
ORBIT_STATIC_ASSERT(offsetof(OrbitGpuCullInfo, cull_planes) == 128, "cull_planes @128");
ORBIT_STATIC_ASSERT(offsetof(OrbitGpuCullInfo, cull_plane_count) == 320, "cull_plane_count @320");
ORBIT_STATIC_ASSERT(offsetof(OrbitGpuCullInfo, occlusion_pass) == 332, "occlusion_pass @332");
ORBIT_STATIC_ASSERT(offsetof(OrbitGpuCullInfo, projection_type) == 352, "projection_type @352");
ORBIT_STATIC_ASSERT(offsetof(OrbitGpuCullInfo, lod_target_pos_view_space) == 384, "lod_target @384");
ORBIT_STATIC_ASSERT(offsetof(OrbitGpuCullInfo, max_mesh_lod) == 396, "max_mesh_lod @396");

/* ------------------------------------------------------------------------ */
/* Light-cluster layouts                                                     */
/* ------------------------------------------------------------------------ */

/* types.glsl:251-255: bit patterns of floats, u32-compared */
typedef struct OrbitClusterDepthBounds {
    uint32_t min_depth; /* bits(1 - depth) maximised  (mark_active.comp:33) */
    uint32_t max_depth; /* bits(depth) maximised      (mark_active.comp:34) */
} OrbitClusterDepthBounds;

/* CompactedClusterIndexList (types.glsl:270-276): also indirect args */
#define ORBIT_COMPACT_HEADER 16u /* wg_x, wg_y, wg_z, cluster_count; indices @16 */
/* ClusterLightIndices (types.glsl:246-249): u32 light_count @0, indices @4 */
#define ORBIT_LIGHT_INDEX_HEADER 4u

/* mark_active.comp:8-23 push-constant block, 56 B, with the three bindless
 * indices (depth_buffer, mask, bounds) kept in place. */
typedef struct OrbitMarkActivePush {
    uint32_t cluster_count[3];
    uint32_t tile_size_px;
    uint32_t screen_size[2];
    float z_near;
    float z_far;
    float z_scale;
    float z_bias;
    uint32_t depth_buffer;
    uint32_t depth_buffer_sample_count;
    uint32_t depth_slice_mask_buffer;
    uint32_t depth_bounds_buffer;
} OrbitMarkActivePush;
ORBIT_STATIC_ASSERT(sizeof(OrbitMarkActivePush) == 56, "mark_active push block is 56 B");

/* ClusterCullInfo, cluster.rs:186-207 == ClusterLightCullInfoBuffer,
 * light_culling.comp:8-26 */
typedef struct OrbitClusterCullInfo {
    float world_to_view_matrix[16];  /* @0   */
    float screen_to_view_matrix[16]; /* @64  inverse projection */
    uint32_t cluster_count[3];       /* @128 */
    uint32_t tile_size_px;           /* @140 */
    uint32_t screen_size[2];         /* @144 */
    float z_near;                    /* @152 */
    float z_far;                     /* @156 */
    uint32_t unique_cluster_buffer;  /* @160 bindless indices, unused by HIP */
    uint32_t cluster_offset_image;   /* @164 */
    uint32_t light_index_buffer;     /* @168 */
    uint32_t depth_bounds_buffer;    /* @172 */
    uint32_t global_light_count;     /* @176 */
    uint32_t global_light_list;      /* @180 */
    uint32_t _padding[2];            /* @184 */
} OrbitClusterCullInfo;
ORBIT_STATIC_ASSERT(sizeof(OrbitClusterCullInfo) == 192, "ClusterCullInfo is 192 B");
ORBIT_STATIC_ASSERT(offsetof(OrbitClusterCullInfo, cluster_count) == 128, "cluster_count @128");
ORBIT_STATIC_ASSERT(offsetof(OrbitClusterCullInfo, global_light_count) == 176, "light count @176");

/* GpuClusterInfoBuffer, cluster.rs:322-335 == ClusterBuffer, types.glsl:42-53 */
typedef struct OrbitGpuClusterInfoBuffer {
    uint32_t cluster_count[3];
    uint32_t tile_size_px;
    uint32_t screen_size[2];
    uint32_t z_slice_count;
    float z_scale;
    float z_bias;
    float luminance_cutoff;
    uint32_t light_offset_image;
    uint32_t light_index_list;
    uint32_t tile_depth_slice_mask_buffer;
} OrbitGpuClusterInfoBuffer;
ORBIT_STATIC_ASSERT(sizeof(OrbitGpuClusterInfoBuffer) == 52, "GpuClusterInfoBuffer is 52 B");

/* ------------------------------------------------------------------------ */
/* Context                                                                   */
/* ------------------------------------------------------------------------ */

/* Runtime capacities replace the reference's compile-time constants
 * (draw_gen.rs:15-16, scene.rs:303-304, assets/mod.rs:197-205).  They size the
 * context's private scratch only; caller buffers stay caller-owned.  What the
 * scratch costs in device memory, by its largest terms: 96 B per dispatch record
 * of max_dispatches (survivor payload of the Meshlet-buffer path) + 8 B (ballots,
 * counts); per cluster of max_clusters 256 B (the light indices the count launch
 * keeps) + 36 B; per group of 256 clusters 20 B x max_lights (the coarse candidate
 * lists: index + view-space light) — 0.57 GB at orbit_default_caps, zero-filled
 * once by orbit_ctx_create, which returns when the fill has run. */
typedef struct OrbitCaps {
    uint32_t max_entities;   /* scene.rs:303   MAX_INSTANCE_COUNT   = 100_000   */
    uint32_t max_dispatches; /* draw_gen.rs:16 MAX_MESHLET_DISPATCH = 1_000_000 */
    uint32_t max_draws;      /* draw_gen.rs:15 MAX_DRAW_COUNT       = 1_000_000 */
    uint32_t max_lights;     /* scene.rs:304   MAX_LIGHT_COUNT      = 2_000     */
    uint32_t max_clusters;   /* tiles_x * tiles_y * max(4, z_slices): cluster.rs:53-56 */
    uint32_t dispatch_size;  /* MESHLET_DISPATCH_SIZE, spec constant 0 of both cull shaders, which the reference derives
                                from the device (src/graphics/device.rs:369-372; the shaders' own default is 64,
                                entity_cull.comp:26): 32 (default), 64 or 128.  32 is the one value for which the
                                reference is consistent throughout and serves everything.  With 64 / 128,
                                orbit_entity_cull writes records of up to that many meshlets (entity_cull.comp:210-223)
                                and orbit_meshlet_cull / orbit_cull_views / orbit_frame_late consume them — the command
                                list is the same as with 32, as the reference's binaries' is
                                (tests/golden/spirv_cull_dispatch_sizes.npz) — for OCCLUSION PASS 0.  Refused there, with
                                ORBIT_E_INVALID and the reason: passes 1 / 2 (an entity owns ceil(meshlets / 32)
                                visibility words, src/scene.rs:427, but a record's word offset advances by
                                meshlet_count / S, entity_cull.comp:222, and the ballots are S bits wide), the
                                mesh-shading path (MeshTaskPayload.meshlet_indices has 32 entries whatever S is,
                                types.glsl:196-200) and the sharded engine (one 32-bit ballot per list entry).
                                caps.max_dispatches counts the caller's records; the scratch is sized for S / 32 times
                                as many records of 32. */
    uint32_t max_views;      /* orbit_cull_views: scan scratch for this many views is allocated at context creation
                                (0: for one view; more is then allocated on the first call that needs it — the only
                                allocation an enqueue call can ever make) */
    uint32_t validate_streams; /* debug: != 0 makes every cull / expansion served from a bound meshlet stream first
                                  compare the stream with the meshlet buffer (one extra pass over both, ORBIT_E_STALE
                                  latched on a difference): finds a missing orbit_meshlet_stream_update */
    uint32_t cull_path;        /* orbit_cull_views / orbit_cull_shard: 0 = the library chooses per call (views of at
                                  most 16 384 entity-draws each go through the one-launch cull, larger ones through the
                                  launch chain), 1 = always the launch chain, 2 = the one-launch cull whatever the size —
                                  for calls it can serve: every view with both stages and at least one entity-draw; a
                                  call with a view of 0 entity-draws or skip_meshlet_stage runs the chain under every
                                  setting, as does a context with validate_streams and a bound stream.  Same outputs
                                  either way; tests and A/B timings set it. */
    uint32_t arith_profile;    /* ORBIT_ARITH_*: how the points SPIR-V leaves to a driver's compiler are decided
                                  (DESIGN.md section 2).  0 = canonical: no contraction, the shared software log2 —
                                  the oracle's and the committed vectors' choice.  1 = contracted: dot products, plane
                                  tests, the projected bounds and the cone test fused the way a driver's compiler
                                  contracts them, hardware log2; pinned by tests/golden/spirv_cull_contracted.npz. */
} OrbitCaps;
ORBIT_STATIC_ASSERT(sizeof(OrbitCaps) == 40, "OrbitCaps layout");
#define ORBIT_ARITH_CANONICAL 0u
#define ORBIT_ARITH_CONTRACTED 1u

typedef struct OrbitCtx OrbitCtx;

/* Concurrency contract.  Calls on one context may come from any thread (they are serialised inside the library) and
 * only ENQUEUE work on the caller's stream.  The context owns one set of scratch buffers, so the cull / cluster entry
 * points of ONE context must all be ordered on the device — use one stream per context, or order the streams with
 * events — exactly as the reference's passes are ordered by its render graph.  Work that may overlap (a second view
 * culled on another stream, the per-rank contexts of the sharded engine) takes one context per stream; contexts are
 * independent.  orbit_exchange_list, orbit_allgather_list and orbit_gather_visible touch none of the cull scratch and
 * may run on a side stream of the same context (bench.py does: the exchange of step i overlaps the cull of step i+1). */

uint32_t orbit_abi_version(void);
/* Fills `caps` with the reference's constants listed above. */
void orbit_default_caps(OrbitCaps *caps);
/* Creates a context on HIP device `device_id`.  Fails with ORBIT_E_NO_DEVICE
 * when no GPU is usable — there is no host fallback. */
int32_t orbit_ctx_create(int32_t device_id, const OrbitCaps *caps, OrbitCtx **out_ctx);
int32_t orbit_ctx_destroy(OrbitCtx *ctx);
/* Message for the last failing call on this thread / context (never NULL). */
const char *orbit_last_error(const OrbitCtx *ctx);
/* Device-latched status (ORBIT_OK / ORBIT_E_CAPACITY / ORBIT_E_RANGE / ORBIT_E_STALE / ORBIT_E_TIMEOUT).
 * `sync` != 0 first waits for `stream` to drain.  Clears the latch. */
int32_t orbit_ctx_status(OrbitCtx *ctx, void *stream, int32_t sync);

/* ------------------------------------------------------------------------ */
/* entity_cull  — replaces the clear + ComputePass recorded by              */
/* create_meshlet_dispatch_command (draw_gen.rs:327-380) and the first half */
/* of create_draw_commands (draw_gen.rs:283-307); device code restates      */
/* shaders/entity_cull.comp:104-245.                                        */
/* ------------------------------------------------------------------------ */

/* One level of a depth pyramid that is NOT one packed buffer: the reference's pyramid is a Vulkan image with a view
 * per mip (draw_gen.rs:456-482); exported for HIP it becomes one linear-tiled image (or buffer) per level, each with
 * its own base address and row pitch (vkGetImageSubresourceLayout::rowPitch / 4).  A DEVICE array of mip_levels such
 * entries can stand in for the packed chain wherever a pyramid is read (the *_CullBufs below) or written
 * (orbit_depth_reduce_levels).  INTEGRATION.md "Depth buffer and pyramid" has the Vulkan side. */
typedef struct OrbitDepthPyramidLevel {
    float *texels;       /* DEVICE pointer to the level's first texel */
    uint32_t row_pitch;  /* texels per row, >= max(width >> level, 1) */
    uint32_t _pad;
} OrbitDepthPyramidLevel;
ORBIT_STATIC_ASSERT(sizeof(OrbitDepthPyramidLevel) == 16, "pyramid level entry is 16 B");

/* Raw device pointers in the push-constant order of entity_cull.comp:17-23
 * (== draw_gen.rs:372-376); the three CullInfo-carried resources follow. */
typedef struct OrbitEntityCullBufs {
    const void *entity_draw_buffer;  /* EntityDrawBuffer */
    const void *mesh_info_buffer;    /* OrbitMeshInfo[] */
    void *meshlet_dispatch_buffer;   /* MeshletDispatchBuffer, written */
    const void *entity_buffer;       /* OrbitEntityData[] */
    uint32_t *visibility_buffer;     /* entity bitset; required iff occlusion_pass != 0 */
    const float *depth_pyramid;      /* R32F mip chain (orbit_depth_pyramid_desc); pass 2 */
    uint32_t depth_pyramid_size[2];  /* mip-0 width, height = textureSize(.., 0) */
    uint32_t dispatch_capacity;      /* records the dispatch buffer can hold */
    uint32_t _pad;
    /* NULL: depth_pyramid is the packed chain.  Else a DEVICE array of mip levels (depth_pyramid is ignored). */
    const OrbitDepthPyramidLevel *depth_pyramid_levels;
} OrbitEntityCullBufs;
ORBIT_STATIC_ASSERT(sizeof(OrbitEntityCullBufs) == 72, "EntityCullBufs is 72 B");

/* The library performs the {0,1,1} header clear itself (draw_gen.rs:356-363).
 * `cull_info` is a HOST pointer to the 400-B block CullInfo::to_gpu built.
 * Grid = ceil(entity_draw_count / 256) like draw_gen.rs:377; the in-buffer
 * `count` bounds the work like entity_cull.comp:106. */
int32_t orbit_entity_cull(OrbitCtx *ctx, const OrbitGpuCullInfo *cull_info,
                          const OrbitEntityCullBufs *bufs, uint32_t entity_draw_count,
                          void *stream);
/* Same over the entity-draw range [draw_first, draw_first + draw_count) only:
 * the per-rank shard of the multi-GPU path.  draw_first must
 * be a multiple of 32 so that entity-visibility words stay rank-private. */
int32_t orbit_entity_cull_range(OrbitCtx *ctx, const OrbitGpuCullInfo *cull_info,
                                const OrbitEntityCullBufs *bufs, uint32_t draw_first,
                                uint32_t draw_count, void *stream);

/* ------------------------------------------------------------------------ */
/* meshlet_cull — replaces create_meshlet_draw_commands (draw_gen.rs:382-435)*/
/* and the second half of create_draw_commands; device code restates        */
/* shaders/meshlet_cull.comp:108-255.                                       */
/* ------------------------------------------------------------------------ */

/* push-constant order of meshlet_cull.comp:16-23 (== draw_gen.rs:426-431) */
typedef struct OrbitMeshletCullBufs {
    const void *meshlet_dispatch_buffer; /* read: header x = record count (device side) */
    const void *meshlet_buffer;          /* OrbitMeshlet[] */
    void *draw_commands_buffer;          /* MeshletDrawCommandBuffer, written */
    const void *entity_buffer;           /* OrbitEntityData[] */
    const void *material_buffer;         /* OrbitMaterialData[] */
    uint32_t *meshlet_visibility_buffer; /* required iff cull_info says present */
    const float *depth_pyramid;
    uint32_t depth_pyramid_size[2];
    uint32_t dispatch_capacity; /* upper bound of records in the dispatch buffer, which holds at least
                                   12 + 16 * dispatch_capacity bytes (the library may read any of them, also
                                   records past the count in the header) */
    uint32_t draw_capacity;     /* commands the draw buffer can hold */
    uint32_t material_count;    /* entries in material_buffer (assets/mod.rs: materials arena);
                                   0 = unknown.  Only an optimisation hint: when given (<= 4096)
                                   the alpha modes are staged in LDS instead of gathered. */
    uint32_t _pad;
    /* NULL: depth_pyramid is the packed chain.  Else a DEVICE array of mip levels (depth_pyramid is ignored). */
    const OrbitDepthPyramidLevel *depth_pyramid_levels;
} OrbitMeshletCullBufs;
ORBIT_STATIC_ASSERT(sizeof(OrbitMeshletCullBufs) == 88, "MeshletCullBufs is 88 B");

/* The record count is read on the device from the dispatch header — the
 * equivalent of vkCmdDispatchIndirect (draw_gen.rs:432); no host readback.
 * The `count = 0` clear (draw_gen.rs:411-417) is performed by the library. */
int32_t orbit_meshlet_cull(OrbitCtx *ctx, const OrbitGpuCullInfo *cull_info,
                           const OrbitMeshletCullBufs *bufs, void *stream);

/* ------------------------------------------------------------------------ */
/* depth_reduce — replaces DepthPyramid::{new,resize,update}                 */
/* (draw_gen.rs:456-566); device code restates shaders/depth_reduce.comp and */
/* the ReduceMin sampler (src/graphics/device.rs:1404-1420).                 */
/* ------------------------------------------------------------------------ */

#define ORBIT_MAX_PYRAMID_MIPS 16
typedef struct OrbitDepthPyramidDesc {
    uint32_t width, height; /* mip 0: next_pow2(screen)/2, draw_gen.rs:458 */
    uint32_t mip_levels;    /* math.rs:18-20 */
    uint32_t total_texels;  /* sum over mips; buffer size = 4 * total_texels */
    uint32_t mip_offset[ORBIT_MAX_PYRAMID_MIPS]; /* texel offset of each level */
    uint32_t mip_width[ORBIT_MAX_PYRAMID_MIPS];  /* max(dim >> k, 1), image.rs:533 */
    uint32_t mip_height[ORBIT_MAX_PYRAMID_MIPS];
} OrbitDepthPyramidDesc;

/* Host-only: pyramid geometry for a screen_width x screen_height depth buffer. */
int32_t orbit_depth_pyramid_desc(uint32_t screen_width, uint32_t screen_height,
                                 OrbitDepthPyramidDesc *desc);
/* Same but from the mip-0 size (what the cull kernels get as textureSize). */
int32_t orbit_depth_pyramid_desc_from_mip0(uint32_t mip0_width, uint32_t mip0_height,
                                           OrbitDepthPyramidDesc *desc);
/* All mips in one launch (the reference loops <= 12 dispatches with barriers,
 * draw_gen.rs:538-564).  `depth` is the W x H reverse-Z D32F buffer, row-major. */
int32_t orbit_depth_reduce(OrbitCtx *ctx, const float *depth, uint32_t screen_width,
                           uint32_t screen_height, float *pyramid, void *stream);

/* ------------------------------------------------------------------------ */
/* light_cluster — replaces compute_clusters and its three stages            */
/* (cluster.rs:368-591); device code restates shaders/light_cluster/ (the .comp files). */
/* ------------------------------------------------------------------------ */

/* mark_active_clusters (cluster.rs:399-477).  `depth` is W x H x samples
 * floats, sample index fastest.  Both outputs are fully (re)written, which
 * subsumes the zero-fill of cluster.rs:439-456. */
int32_t orbit_cluster_mark(OrbitCtx *ctx, const OrbitMarkActivePush *push, const float *depth,
                           uint32_t *tile_depth_slice_mask, /* tiles_x * tiles_y words */
                           OrbitClusterDepthBounds *depth_bounds, /* cx*cy*cz entries */
                           void *stream);

/* compact_active_clusters (cluster.rs:479-517).  Output order is ascending
 * linear cluster index; header = {ceil(n/256), 1, 1, n}. */
int32_t orbit_cluster_compact(OrbitCtx *ctx, const uint32_t cluster_count[3],
                              const uint32_t *tile_depth_slice_mask,
                              void *unique_cluster_buffer, uint32_t index_capacity,
                              void *stream);

/* cluster_light_assignment (cluster.rs:519-591).  `info` is a HOST pointer.
 * `cluster_offset_image` is the RG32UI 3-D image as cx*cy*cz (offset,count)
 * pairs in linear-index order; only active clusters are written, like the
 * reference's imageStore (light_culling.comp:150). */
int32_t orbit_cluster_assign(OrbitCtx *ctx, const OrbitClusterCullInfo *info,
                             const void *unique_cluster_buffer,
                             const OrbitClusterDepthBounds *depth_bounds,
                             const OrbitLightData *lights,
                             void *light_index_buffer, uint32_t light_index_capacity,
                             uint32_t *cluster_offset_image, void *stream);

/* compute_clusters (cluster.rs:368-397): the three stages above in the
 * reference's order on one stream — mark_active_clusters, then
 * compact_active_clusters on its mask, then cluster_light_assignment on the
 * compacted list and the depth bounds.  `push` and `info` are HOST pointers
 * and must describe the same cluster grid. */
int32_t orbit_compute_clusters(OrbitCtx *ctx, const OrbitMarkActivePush *push,
                               const OrbitClusterCullInfo *info, const float *depth,
                               const OrbitLightData *lights,
                               uint32_t *tile_depth_slice_mask,
                               OrbitClusterDepthBounds *depth_bounds,
                               void *unique_cluster_buffer, uint32_t index_capacity,
                               void *light_index_buffer, uint32_t light_index_capacity,
                               uint32_t *cluster_offset_image, void *stream);

/* ------------------------------------------------------------------------ */
/* Multi-GPU helpers: one process per GPU, contiguous entity */
/* ranges per rank, one exchange step over RCCL.                             */
/* ------------------------------------------------------------------------ */

/* Contiguous, 32-aligned entity-draw range owned by `rank` of `world`. */
void orbit_shard_range(uint32_t entity_draw_count, uint32_t rank, uint32_t world,
                       uint32_t *begin, uint32_t *end);

/* Rank-ordered all-gather of the ranks' MeshletDrawCommandBuffers: because
 * every rank's list is in canonical order and the shards are contiguous in
 * entity order, the concatenation in rank order IS the single-GPU list
 * (the reference is single-GPU; this is the exchange step of the sharded
 * engine).
 *   nccl_comm         the caller's ncclComm_t (RCCL), `rank` of `world` in it
 *   local_draw_buffer this rank's buffer (count @0, 28-B commands @4)
 *   out_draw_buffer   receives count @0 and all commands @4, rank order
 *   out_capacity      commands `out_draw_buffer` can hold
 * RCCL is resolved from the library already loaded in the process (the one
 * that made `nccl_comm`); there is no link-time dependency.  The counts are
 * all-gathered first and read on the host (message sizes): ONE stream
 * synchronisation per call, the only entry point that has one — SURVEY.md
 * §8b's export for a caller that wants the 28-B commands themselves gathered
 * through its own communicator.  It is NOT the sharded engine's step: that
 * sends 12-B records with no host in it (orbit_abi_ext.h: orbit_cull_shard,
 * orbit_exchange_list / orbit_allgather_list).  The lists
 * travel as grouped send/recv straight into out + prefix[rank] (direct
 * fan-out over the point-to-point xGMI links).  ORBIT_E_CAPACITY if the
 * concatenation does not fit (nothing is written past out_capacity). */
int32_t orbit_gather_visible(OrbitCtx *ctx, void *nccl_comm, uint32_t rank, uint32_t world,
                             const void *local_draw_buffer, void *out_draw_buffer,
                             uint32_t out_capacity, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ORBIT_ABI_H */