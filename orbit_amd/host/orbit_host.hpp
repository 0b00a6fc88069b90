// orbit_host.hpp — host-side mirror of the reference's cull-pass API, in C++.
//
// The reference's host code for this path is Rust (src/passes/draw_gen.rs,
// src/passes/cluster.rs, src/math.rs, src/camera.rs); no Rust toolchain exists
// in this image, so the same interface — same names, argument meaning and
// failure behaviour — is restated here above the C ABI (include/orbit_abi.h).
// It contains parameter derivation and call sequencing only; all device work
// goes through liborbit_cull.so.  Reference panics (assert!/unwrap) become
// orbit::Panic exceptions.
#pragma once
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/orbit_abi_ext.h"

namespace orbit {

struct Panic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct Vec3 {
    float x = 0, y = 0, z = 0;
};
struct Vec4 {
    float x = 0, y = 0, z = 0, w = 0;
};

// glam::Mat4 — column-major
struct Mat4 {
    float c[16] = {0};
    static Mat4 identity();
    static Mat4 from_cols_array(const float *a);
    // glam Mat4::perspective_infinite_reverse_rh (src/camera.rs:87)
    static Mat4 perspective_infinite_reverse_rh(float fov_y, float aspect, float z_near);
    // glam Mat4::orthographic_rh (src/camera.rs:91-96)
    static Mat4 orthographic_rh(float left, float right, float bottom, float top, float near, float far);
    Mat4 mul(const Mat4 &o) const;
    Vec4 mul_vec4(const Vec4 &v) const;
    Mat4 transpose() const;
    Mat4 inverse() const;
    Vec4 col(int i) const { return {c[4 * i], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3]}; }
};

struct Quat { // glam::Quat (x, y, z, w)
    float x = 0, y = 0, z = 0, w = 1;
    Quat inverse() const { return {-x, -y, -z, w}; } // unit quaternions: the conjugate
};

namespace math {
uint32_t mip_levels_from_size(uint32_t max_size);                                  // src/math.rs:18-20
float frustum_split(float near, float far, float lambda, float ratio);             // src/math.rs:64-69
void perspective_corners(float fovy, float aspect_ratio, float near, float far, Vec4 out[8]); // :149-168
Mat4 mat4_from_quat(const Quat &q);                                                // glam Mat4::from_quat
Mat4 mat4_from_translation(const Vec3 &t);                                         // glam Mat4::from_translation
Mat4 mat4_from_scale_rotation_translation(const Vec3 &s, const Quat &q, const Vec3 &t);
void frustum_planes_from_matrix(const Mat4 &m, Vec4 planes[6]);                    // src/math.rs:71-84
Vec4 normalize_plane(const Vec4 &p);                                               // src/math.rs:86-89
float largest_scale_from_matrix(const Mat4 &m);                                    // src/math.rs:132-138
bool project_sphere_clip_space(const Vec4 &sphere, float znear, float p00, float p11, Vec4 *aabb); // :170-199
} // namespace math

namespace camera {
// src/camera.rs:69-113
struct Projection {
    enum Kind { Orthographic, Perspective } kind = Perspective;
    float half_width = 0, near_clip = 0, far_clip = 0, fov = 0;
    static Projection perspective(float fov, float near_clip);
    static Projection orthographic(float half_width, float near_clip, float far_clip);
    Mat4 compute_matrix(float aspect_ratio) const;
    float z_near() const { return near_clip; }
    float z_far() const;
};
} // namespace camera

namespace graphics {
// GraphBufferHandle / GraphImageHandle: raw device memory instead of graph handles
struct GraphBufferHandle {
    void *ptr = nullptr;
    size_t size = 0;
    bool valid() const { return ptr != nullptr; }
};
struct GraphImageHandle {
    float *ptr = nullptr;
    uint32_t width = 0, height = 0, mip_levels = 0; // mip-0 size of a pyramid, or the size of a depth buffer
    uint32_t samples = 1;
    bool valid() const { return ptr != nullptr; }
};

// The slice of graphics::Context the passes use (src/graphics/context.rs:1162-1342): transient buffers from a
// name + size cache (context.rs:1275-1316) and ordering = one HIP stream.
class Context {
  public:
    Context(int device, void *stream, const OrbitCaps *caps);
    ~Context();
    Context(const Context &) = delete;
    GraphBufferHandle create_transient(const std::string &name, size_t size);
    // Derived meshlet streams (include/orbit_abi.h): what GpuAssets::add_mesh calls right after
    // queue_write_buffer(&self.meshlet_buffer, ..) (assets/mod.rs:441-445) for the range it wrote.  The stream is
    // created on first use for `capacity` meshlets (the size of meshlet_buffer, MAX_MESHLET_COUNT, assets/mod.rs:202,272-276), re-derived for
    // [first, first + count) on this context's stream and bound; culls of passes 0 and 2 then read it.
    // A buffer that moved (re-created larger: assets grow) has everything uploaded so far re-derived from its new
    // address, not only the new range.
    void sync_meshlet_stream(const GraphBufferHandle &meshlet_buffer, uint64_t first, uint64_t count,
                             uint64_t capacity);
    // what GpuAssets::add_material calls after its upload (assets/mod.rs:520-526): the stream's alpha classes
    void sync_meshlet_stream_materials(const GraphBufferHandle &materials_buffer, uint32_t material_count);
    OrbitCtx *abi() const { return ctx_; }
    void *stream() const { return stream_; }
    void check(int32_t rc) const; // non-zero status -> Panic with orbit_last_error

  private:
    OrbitCtx *ctx_ = nullptr;
    OrbitMeshletStream *meshlet_stream_ = nullptr;
    uint64_t meshlet_stream_capacity_ = 0;
    uint64_t meshlet_stream_populated_ = 0;       // high-water mark of the meshlets uploaded so far
    const void *meshlet_stream_source_ = nullptr; // the buffer they were derived from
    // the material buffer the stream's alpha classes mirror (sync_meshlet_stream_materials): re-issued when the stream
    // is re-created, forgotten (nullptr) before that buffer is freed or re-created
    const void *stream_materials_ = nullptr;
    uint32_t stream_material_count_ = 0;
    void *stream_ = nullptr;
    std::map<std::string, GraphBufferHandle> transients_;
};
} // namespace graphics

namespace passes {

constexpr size_t MAX_DRAW_COUNT = 1000000;             // draw_gen.rs:15
constexpr size_t MAX_MESHLET_DISPATCH_COUNT = 1000000; // draw_gen.rs:16
constexpr size_t MAX_CULL_PLANES = 12;                 // draw_gen.rs:206

struct AlphaModeFlags { // draw_gen.rs:630-641
    uint32_t bits = 0;
    static constexpr uint32_t OPAQUE = 1, MASKED = 2, TRANSPARENT = 4, ALL = 7;
};

// draw_gen.rs:18-103
struct OcclusionCullInfo {
    enum Kind { None, VisibilityRead, VisibilityWrite } kind = None;
    graphics::GraphBufferHandle visibility_buffer;
    graphics::GraphBufferHandle meshlet_visibility_buffer; // Option: invalid handle = None
    graphics::GraphImageHandle depth_pyramid;
    AlphaModeFlags noskip_alphamode;
    float aspect_ratio = 1.0f;
    uint32_t pass_index() const { return kind == None ? 0u : kind == VisibilityRead ? 1u : 2u; }
};

// draw_gen.rs:105-118
struct CullInfo {
    Mat4 view_matrix;
    std::vector<Vec4> view_space_cull_planes;
    camera::Projection projection;
    OcclusionCullInfo occlusion_culling;
    AlphaModeFlags alpha_mode_filter;
    size_t lod_range_start = 0, lod_range_end = 8; // Range<usize>
    float lod_base = 16.0f, lod_step = 2.0f;
    Vec3 lod_target_pos_view_space;
    // CullInfo::to_gpu, draw_gen.rs:121-203.  Present resources get a non-sentinel index.
    OrbitGpuCullInfo to_gpu() const;
};

// assets/mod.rs:207-215 and scene.rs:293-301, reduced to what the cull path binds
struct AssetGraphData {
    graphics::GraphBufferHandle meshlet_buffer, mesh_info_buffer, materials_buffer;
    uint32_t material_count = 0;
};
struct SceneGraphData {
    size_t entity_draw_count = 0;
    graphics::GraphBufferHandle entity_draw_buffer, entity_buffer, light_data_buffer, meshlet_visibility_buffer;
    size_t light_count = 0;
};

// draw_gen.rs:327-380: returns (cull_info data, meshlet_dispatch_buffer)
std::pair<OrbitGpuCullInfo, graphics::GraphBufferHandle> create_meshlet_dispatch_command(
    graphics::Context &context, const std::string &draw_commands_name, const AssetGraphData &assets,
    const SceneGraphData &scene, const CullInfo &cull_info);
// draw_gen.rs:382-435
graphics::GraphBufferHandle create_meshlet_draw_commands(graphics::Context &context,
                                                         const std::string &draw_commands_name,
                                                         const AssetGraphData &assets, const SceneGraphData &scene,
                                                         const CullInfo &cull_info,
                                                         graphics::GraphBufferHandle meshlet_dispatch_buffer);
// draw_gen.rs:239-325
graphics::GraphBufferHandle create_draw_commands(graphics::Context &context, const std::string &draw_commands_name,
                                                 const AssetGraphData &assets, const SceneGraphData &scene,
                                                 const CullInfo &cull_info,
                                                 const graphics::GraphBufferHandle *reuse_buffer);

// draw_gen.rs:451-566
class DepthPyramid {
  public:
    DepthPyramid(graphics::Context &context, const std::string &name, uint32_t width, uint32_t height);
    void resize(graphics::Context &context, uint32_t width, uint32_t height);
    graphics::GraphImageHandle get_current() const { return pyramid; }
    void update(graphics::Context &context, const graphics::GraphImageHandle &depth_buffer);
    graphics::GraphImageHandle pyramid;
    bool usable = false;

  private:
    std::string name_;
};
// draw_gen.rs:569-628
void update_multiple_depth_pyramids(graphics::Context &context, DepthPyramid *const *depth_pyramids,
                                    const graphics::GraphImageHandle *depth_buffers, size_t count);

// scene.rs:20-57
struct Transform {
    Vec3 position;
    Quat orientation;
    Vec3 scale{1.0f, 1.0f, 1.0f};
    Mat4 compute_matrix() const; // Mat4::from_scale_rotation_translation
};

// camera.rs:115-142 with its transform (the cascade builder needs view-to-world)
struct SceneCamera {
    Transform transform;
    camera::Projection projection;
    float aspect_ratio = 1.0f;
    Mat4 compute_matrix() const; // projection * inverse(transform)
};

// shadow_renderer.rs:27,44-73
constexpr size_t MAX_SHADOW_CASCADE_COUNT = 4;
struct ShadowSettings {
    uint32_t shadow_resolution = 2048;
    float cascade_split_lambda = 0.80f;
    float max_shadow_distance = 32.0f;
};
// the slice of app.rs Settings the cascade CullInfo reads (app.rs:370-372)
struct LodSettings {
    size_t min_mesh_lod = 0, max_mesh_lod = 7;
    float lod_base = 16.0f, lod_step = 2.0f;
};
struct ShadowCascade {
    CullInfo cull_info;             // what render_shadow_map hands to the cull passes (:688-706)
    Mat4 light_projection_matrix;   // projection_matrix * light_matrix (:552)
    float shadow_map_world_size = 0; // radius * 2 (:511)
};
// The per-cascade part of ShadowRenderer::render_cascaded_shadow (shadow_renderer.rs:466-706): split distances,
// texel-snapped light-space bounding sphere of the sub-frustum, and the orthographic CullInfo with the six light
// planes plus the camera's planes that face the light (up to 11 planes).
ShadowCascade shadow_cascade(const ShadowSettings &settings, const LodSettings &lod, const Quat &direction,
                             const SceneCamera &camera, size_t cascade_index, bool frustum_culling = true);

// cluster.rs:14-72
struct ClusterSettings {
    uint32_t px_size_power = 3;
    uint32_t screen_resolution[2] = {0, 0};
    uint32_t z_slice_count = 32;
    float far_plane = 200.0f;
    float luminance_cutoff = 0.25f;
    uint32_t tile_px_size() const { return 1u << px_size_power; }
    void set_resolution(uint32_t w, uint32_t h) { screen_resolution[0] = w, screen_resolution[1] = h; }
    void tile_counts(size_t out[2]) const;
    size_t linear_cluster_count() const;
    size_t linear_max_allocated_cluster_count() const;
    void cluster_counts(size_t out[3]) const;
    std::pair<float, float> cluster_grid_info(float near) const; // (z_scale, z_bias)
};

struct Camera { // camera.rs:115-142, with the view matrix given directly
    Mat4 view_matrix;
    camera::Projection projection;
    float aspect_ratio = 1.0f;
    Mat4 compute_projection_matrix() const { return projection.compute_matrix(aspect_ratio); }
    float z_near() const { return projection.z_near(); }
};

// The reference's CPU twin of the cluster volume (cluster.rs:134-184, drawn by debug_cluster_volumes :250-320): the
// view-space AABB of cluster `cluster_id` with the ANALYTIC near/far of its depth slice.  light_culling.comp:62-90
// builds the same box from the MEASURED depth bounds of the slice, which lie inside the analytic ones — so this box
// contains the device's (tests/test_host_cpu.py).
struct Aabb {
    Vec3 min, max;
};
Vec3 screen_to_view(const Mat4 &screen_to_view_matrix, const float screen_size[2], const Vec4 &screen_pos); // :134-141
Aabb compute_cluster_aabb(const Mat4 &inverse_projection, const float screen_size[2], float tile_size_px,
                          const float cluster_count[3], float z_near, float z_far, const float cluster_id[3]);

struct GraphClusterInfo { // cluster.rs:362-366
    graphics::GraphBufferHandle light_offset_image; // RG32UI 3-D image as (offset,count) pairs
    graphics::GraphBufferHandle light_index_list;
    OrbitGpuClusterInfoBuffer info;
    graphics::GraphBufferHandle tile_depth_slice_mask, depth_bounds, unique_cluster_buffer;
};

// cluster.rs:368-397 and its stages :399-591
GraphClusterInfo compute_clusters(graphics::Context &context, const ClusterSettings &settings, const Camera &camera,
                                  const graphics::GraphImageHandle &depth_buffer, const SceneGraphData &scene);
std::pair<graphics::GraphBufferHandle, graphics::GraphBufferHandle> mark_active_clusters(
    graphics::Context &context, const ClusterSettings &settings, const graphics::GraphImageHandle &depth_buffer,
    const Camera &camera);
graphics::GraphBufferHandle compact_active_clusters(graphics::Context &context, const ClusterSettings &settings,
                                                    graphics::GraphBufferHandle active_cluster_mask);
std::pair<graphics::GraphBufferHandle, graphics::GraphBufferHandle> cluster_light_assignment(
    graphics::Context &context, const ClusterSettings &settings, const Camera &camera, const SceneGraphData &scene,
    graphics::GraphBufferHandle unique_cluster_buffer, graphics::GraphBufferHandle depth_bounds_buffer);

} // namespace passes
} // namespace orbit
