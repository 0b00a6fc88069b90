/* orbit_host_c.h — flat C surface of the C++ host mirror (orbit_host.hpp), for the
 * ctypes-driven test-suite.  The functions keep the reference's names
 * (src/passes/draw_gen.rs, src/passes/cluster.rs, src/math.rs, src/camera.rs). */
#ifndef ORBIT_HOST_C_H
#define ORBIT_HOST_C_H
#include "../../include/orbit_abi_ext.h"
#ifdef __cplusplus
extern "C" {
#endif

#define ORBIT_HOST_PANIC (-100) /* the reference would have panicked; text in orbit_host_last_error() */

typedef struct OrbitHostProjection {
    uint32_t kind; /* 0 = Perspective{fov, near_clip}, 1 = Orthographic{half_width, near_clip, far_clip} */
    float fov, half_width, near_clip, far_clip;
} OrbitHostProjection;

/* CullInfo, draw_gen.rs:105-118 (+ OcclusionCullInfo :18-33) */
typedef struct OrbitHostCullInfo {
    float view_matrix[16];
    float view_space_cull_planes[16][4]; /* room for > 12 so that the assert can be exercised */
    uint32_t plane_count;
    OrbitHostProjection projection;
    uint32_t occlusion_kind; /* 0 None, 1 VisibilityRead, 2 VisibilityWrite */
    void *visibility_buffer;
    void *meshlet_visibility_buffer; /* NULL = Option::None */
    float *depth_pyramid;
    uint32_t depth_pyramid_size[2];
    uint32_t noskip_alphamode;
    float aspect_ratio;
    uint32_t alpha_mode_filter;
    uint32_t lod_range[2];
    float lod_base, lod_step;
    float lod_target_pos_view_space[3];
} OrbitHostCullInfo;

typedef struct OrbitHostAssets {
    void *meshlet_buffer, *mesh_info_buffer, *materials_buffer;
    uint32_t material_count;
} OrbitHostAssets;
typedef struct OrbitHostScene {
    uint64_t entity_draw_count;
    void *entity_draw_buffer, *entity_buffer, *light_data_buffer, *meshlet_visibility_buffer;
    uint64_t light_count;
} OrbitHostScene;
typedef struct OrbitHostBuffer {
    void *ptr;
    uint64_t size;
} OrbitHostBuffer;

typedef struct OrbitHostClusterSettings {
    uint32_t px_size_power, screen_resolution[2], z_slice_count;
    float far_plane, luminance_cutoff;
} OrbitHostClusterSettings;
typedef struct OrbitHostClusterDerived {
    uint64_t tile_counts[2], cluster_counts[3], linear_cluster_count, linear_max_allocated_cluster_count;
    uint32_t tile_px_size;
    float z_scale, z_bias;
} OrbitHostClusterDerived;
typedef struct OrbitHostClusterOut {
    OrbitHostBuffer tile_depth_slice_mask, depth_bounds, unique_cluster_buffer, light_offset_image, light_index_list;
    OrbitGpuClusterInfoBuffer info;
} OrbitHostClusterOut;

const char *orbit_host_last_error(void);

/* math.rs / camera.rs (host only) */
void orbit_host_perspective_infinite_reverse_rh(float fov_y, float aspect, float z_near, float out[16]);
void orbit_host_orthographic_rh(float l, float r, float b, float t, float n, float f, float out[16]);
void orbit_host_projection_compute_matrix(const OrbitHostProjection *p, float aspect, float out[16]);
void orbit_host_mat4_inverse(const float in[16], float out[16]);
void orbit_host_frustum_planes_from_matrix(const float m[16], float out[6][4], int32_t normalize);
uint32_t orbit_host_mip_levels_from_size(uint32_t max_size);
int32_t orbit_host_project_sphere_clip_space(const float sphere[4], float znear, float p00, float p11, float aabb[4]);
/* CullInfo::to_gpu */
int32_t orbit_host_cull_info_to_gpu(const OrbitHostCullInfo *ci, OrbitGpuCullInfo *out);
/* ShadowRenderer::render_cascaded_shadow, per-cascade CullInfo (shadow_renderer.rs:466-706).  direction and the
 * camera orientation are unit quaternions (x, y, z, w); the camera must be perspective {fov, near_clip}. */
typedef struct OrbitHostShadowCascadeIn {
    uint32_t shadow_resolution;
    float cascade_split_lambda, max_shadow_distance;
    uint32_t min_mesh_lod, max_mesh_lod;
    float lod_base, lod_step;
    float direction[4];
    float camera_position[3], camera_orientation[4];
    float camera_fov, camera_near_clip, camera_aspect_ratio;
    uint32_t cascade_index, frustum_culling;
} OrbitHostShadowCascadeIn;
int32_t orbit_host_shadow_cascade(const OrbitHostShadowCascadeIn *in, OrbitHostCullInfo *out_cull_info,
                                  float out_light_projection_matrix[16], float *out_world_size);
/* ClusterSettings methods */
void orbit_host_cluster_settings_derive(const OrbitHostClusterSettings *s, float z_near, OrbitHostClusterDerived *out);

/* compute_cluster_aabb, cluster.rs:150-184 (the reference's CPU twin of the cluster volume): out = min xyz, max xyz */
void orbit_host_compute_cluster_aabb(const float inverse_projection[16], const float screen_size[2], float tile_size_px,
                                     const float cluster_count[3], float z_near, float z_far, const float cluster_id[3],
                                     float out_min_max[6]);

/* graphics::Context slice + passes (need a GPU) */
void *orbit_host_context_create(int32_t device, void *stream, const OrbitCaps *caps);
void orbit_host_context_destroy(void *hctx);
/* Context::sync_meshlet_stream: derived meshlet streams of meshlets [first, first + count) re-derived and bound */
int32_t orbit_host_sync_meshlet_stream(void *hctx, void *meshlet_buffer, uint64_t first, uint64_t count,
                                       uint64_t capacity);
/* Context::sync_meshlet_stream_materials: the stream's alpha classes from the materials buffer */
int32_t orbit_host_sync_meshlet_stream_materials(void *hctx, void *materials_buffer, uint32_t material_count);
int32_t orbit_host_create_meshlet_dispatch_command(void *hctx, const char *name, const OrbitHostAssets *assets,
                                                   const OrbitHostScene *scene, const OrbitHostCullInfo *ci,
                                                   OrbitGpuCullInfo *out_cull_info, OrbitHostBuffer *out_dispatch);
int32_t orbit_host_create_meshlet_draw_commands(void *hctx, const char *name, const OrbitHostAssets *assets,
                                                const OrbitHostScene *scene, const OrbitHostCullInfo *ci,
                                                const OrbitHostBuffer *dispatch, OrbitHostBuffer *out_draws);
int32_t orbit_host_create_draw_commands(void *hctx, const char *name, const OrbitHostAssets *assets,
                                        const OrbitHostScene *scene, const OrbitHostCullInfo *ci,
                                        OrbitHostBuffer *out_draws);
/* DepthPyramid::new + update for a named pyramid; returns the image handle */
int32_t orbit_host_depth_pyramid_update(void *hctx, const char *name, const float *depth, uint32_t width,
                                        uint32_t height, float **out_pyramid, uint32_t out_size_mips[3]);
int32_t orbit_host_compute_clusters(void *hctx, const OrbitHostClusterSettings *settings, const float view_matrix[16],
                                    const OrbitHostProjection *projection, float aspect_ratio, const float *depth,
                                    uint32_t depth_size[2], uint32_t samples, const OrbitHostScene *scene,
                                    OrbitHostClusterOut *out);

/* ---- scene side (orbit_scene.hpp): collections::{Arena, FreeListAllocator}, scene::SceneData (host only) ---- */
typedef struct OrbitHostArenaIndex {
    uint32_t generation, slot;
} OrbitHostArenaIndex;
/* Arena<i64>: arena.rs:98-330 */
void *orbit_host_arena_create(void);
void orbit_host_arena_destroy(void *arena);
OrbitHostArenaIndex orbit_host_arena_insert(void *arena, int64_t value);
int32_t orbit_host_arena_get(const void *arena, OrbitHostArenaIndex index, int64_t *out);    /* 1 = Some */
int32_t orbit_host_arena_remove(void *arena, OrbitHostArenaIndex index, int64_t *out);       /* 1 = Some */
int32_t orbit_host_arena_has_index(const void *arena, OrbitHostArenaIndex index);
uint64_t orbit_host_arena_len(const void *arena);
/* occupied entries in iteration order; returns how many were written (<= capacity) */
uint64_t orbit_host_arena_iter(const void *arena, OrbitHostArenaIndex *indices, int64_t *values, uint64_t capacity);
/* FreeListAllocator: freelist_alloc.rs:22-121 */
void *orbit_host_freelist_create(uint64_t size);
void orbit_host_freelist_destroy(void *alloc);
int32_t orbit_host_freelist_allocate(void *alloc, uint64_t size, OrbitHostArenaIndex *out_index,
                                     uint64_t out_range[2]); /* 1 = Some, 0 = None */
void orbit_host_freelist_deallocate(void *alloc, OrbitHostArenaIndex index);
/* blocks in address order: ranges[2*i], ranges[2*i+1], is_free[i]; returns the block count */
uint64_t orbit_host_freelist_blocks(const void *alloc, uint64_t *ranges, int32_t *is_free, uint64_t capacity);

typedef struct OrbitHostLight {
    float color[3], intensity;
    uint32_t kind; /* 0 Sky, 1 Directional, 2 Point (scene.rs:136-141) */
    float param;   /* Directional: angular_size, Point: inner_radius */
    uint32_t irradiance_map_index, prefiltered_map_index; /* Sky */
    uint32_t cast_shadows;
} OrbitHostLight;
typedef struct OrbitHostEntity {
    float position[3], orientation[4], scale[3]; /* Transform, scene.rs:20-24; quaternion (x, y, z, w) */
    int32_t mesh;                                /* MeshHandle slot, -1 = None */
    int32_t has_light;
    OrbitHostLight light;
    const char *name; /* NULL = None */
} OrbitHostEntity;
/* SceneData (scene.rs:358-492); visibility_chunk_count 0 = MESHLET_VISIBILITY_BUFFER_CHUNK_COUNT */
void *orbit_host_scene_create(uint64_t visibility_chunk_count);
void orbit_host_scene_destroy(void *scene);
int64_t orbit_host_scene_add_entity(void *scene, const OrbitHostEntity *entity);
int32_t orbit_host_scene_set_transform(void *scene, uint64_t entity, const float position[3],
                                       const float orientation[4], const float scale[3]);
int32_t orbit_host_scene_update(void *scene, const OrbitMeshInfo *mesh_infos, uint64_t mesh_info_count,
                                float luminance_cutoff, uint64_t frame_index);
/* the caches update_scene filled: pointers stay valid until the next update / destroy */
const void *orbit_host_scene_entity_draws(const void *scene, uint64_t *count);   /* {instance, mesh, vis_offset}[] */
const OrbitEntityData *orbit_host_scene_entity_data(const void *scene, uint64_t *count);
const OrbitLightData *orbit_host_scene_light_data(const void *scene, uint64_t *count);
uint64_t orbit_host_scene_shadow_command_count(const void *scene);

/* ---- asset side (orbit_assets.hpp): mesh -> Meshlet[] + meshlet data, mesh bounds (host only) ---- */
/* assets::mesh::compute_meshlets (mesh.rs:292-338).  Two-call protocol: with out_meshlets == NULL only the counts are
 * returned; then call again with buffers of those sizes.  positions: xyz per vertex. */
int32_t orbit_host_compute_meshlets(const float *positions, uint64_t vertex_count, const uint32_t *indices,
                                    uint64_t index_count, uint32_t material, uint32_t vertex_offset,
                                    uint32_t data_offset_base, OrbitMeshlet *out_meshlets, uint32_t *out_meshlet_data,
                                    uint64_t *meshlet_count, uint64_t *meshlet_data_words);
/* gltf_loader.rs:480-506 */
void orbit_host_compute_mesh_bounds(const float *positions, uint64_t vertex_count, float aabb_min[3], float aabb_max[3],
                                    float bounding_sphere[4]);

/* ---- asset ingestion (orbit_gltf.hpp): glTF 2.0 (.glb / .gltf) -> materials, meshes with their LOD chains, entities ----
 * load_gltf (gltf_loader.rs:511-676) + GpuAssets::add_mesh (assets/mod.rs:325-476).  NULL + orbit_host_last_error()
 * on failure.  The arrays stay valid until orbit_host_gltf_free. */
typedef struct OrbitHostGltfCounts {
    uint64_t meshes, meshlets, meshlet_data_words, materials, vertices, entities;
} OrbitHostGltfCounts;
typedef struct OrbitHostGltfEntity {
    int32_t mesh;                                /* MeshHandle slot, -1 = None */
    float position[3], orientation[4], scale[3]; /* Transform::from_mat4 of the node's world matrix */
} OrbitHostGltfEntity;
void *orbit_host_gltf_load(const char *path);
void orbit_host_gltf_free(void *scene);
void orbit_host_gltf_counts(const void *scene, OrbitHostGltfCounts *out);
const OrbitMeshInfo *orbit_host_gltf_mesh_infos(const void *scene);
const OrbitMeshlet *orbit_host_gltf_meshlets(const void *scene);
const uint32_t *orbit_host_gltf_meshlet_data(const void *scene);
const OrbitMaterialData *orbit_host_gltf_materials(const void *scene);
const float *orbit_host_gltf_vertex_positions(const void *scene);
/* copies the entities (capacity entries at most); returns their number */
uint64_t orbit_host_gltf_entities(const void *scene, OrbitHostGltfEntity *out, uint64_t capacity);
/* the LOD-chain pieces alone: meshopt::simplify's stand-in (vertex clustering); returns the indices written
 * (<= capacity, whole triangles), *needed = the full result's length */
uint64_t orbit_host_simplify_clustered(const float *positions, uint64_t vertex_count, const uint32_t *indices,
                                       uint64_t index_count, uint64_t target_index_count, uint32_t *out,
                                       uint64_t capacity, uint64_t *needed);
/* Transform::from_mat4 (scene.rs:41-48) of a column-major matrix -> position[3], orientation[4] (x, y, z, w), scale[3] */
void orbit_host_transform_from_mat4(const float matrix[16], float position[3], float orientation[4], float scale[3]);
#ifdef __cplusplus
}
#endif
#endif
