/* orbit_host_c.h — flat C surface of the C++ host mirror (orbit_host.hpp), for the
 * ctypes-driven test-suite.  The functions keep the reference's names
 * (src/passes/draw_gen.rs, src/passes/cluster.rs, src/math.rs, src/camera.rs). */
#ifndef ORBIT_HOST_C_H
#define ORBIT_HOST_C_H
#include "../../include/orbit_abi.h"
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

/* graphics::Context slice + passes (need a GPU) */
void *orbit_host_context_create(int32_t device, void *stream, const OrbitCaps *caps);
void orbit_host_context_destroy(void *hctx);
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
#ifdef __cplusplus
}
#endif
#endif
