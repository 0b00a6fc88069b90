// orbit_host.cpp — see orbit_host.hpp.  Parameter derivation and call
// sequencing of the reference's cull passes above the C ABI; no device code.
#include "orbit_host.hpp"

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstring>

namespace orbit {

// ------------------------------------------------------------------------- Mat4
Mat4 Mat4::identity() {
    Mat4 m;
    m.c[0] = m.c[5] = m.c[10] = m.c[15] = 1.0f;
    return m;
}
Mat4 Mat4::from_cols_array(const float *a) {
    Mat4 m;
    std::memcpy(m.c, a, sizeof(m.c));
    return m;
}
Mat4 Mat4::perspective_infinite_reverse_rh(float fov_y, float aspect, float z_near) {
    const float f = 1.0f / std::tan(0.5f * fov_y);
    Mat4 m;
    m.c[0] = f / aspect;
    m.c[5] = f;
    m.c[11] = -1.0f;   // column 2 = (0, 0, 0, -1)
    m.c[14] = z_near;  // column 3 = (0, 0, z_near, 0)
    return m;
}
Mat4 Mat4::orthographic_rh(float left, float right, float bottom, float top, float near, float far) {
    const float rcp_width = 1.0f / (right - left);
    const float rcp_height = 1.0f / (top - bottom);
    const float r = 1.0f / (near - far);
    Mat4 m;
    m.c[0] = rcp_width + rcp_width;
    m.c[5] = rcp_height + rcp_height;
    m.c[10] = r;
    m.c[12] = -(left + right) * rcp_width;
    m.c[13] = -(top + bottom) * rcp_height;
    m.c[14] = r * near;
    m.c[15] = 1.0f;
    return m;
}
Mat4 Mat4::mul(const Mat4 &o) const {
    Mat4 out;
    for (int j = 0; j < 4; j++)
        for (int r = 0; r < 4; r++)
            out.c[4 * j + r] = ((c[r] * o.c[4 * j] + c[4 + r] * o.c[4 * j + 1]) + c[8 + r] * o.c[4 * j + 2]) +
                               c[12 + r] * o.c[4 * j + 3];
    return out;
}
Vec4 Mat4::mul_vec4(const Vec4 &v) const {
    float o[4];
    for (int r = 0; r < 4; r++) o[r] = ((c[r] * v.x + c[4 + r] * v.y) + c[8 + r] * v.z) + c[12 + r] * v.w;
    return {o[0], o[1], o[2], o[3]};
}
Mat4 Mat4::transpose() const {
    Mat4 t;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) t.c[4 * i + j] = c[4 * j + i];
    return t;
}
Mat4 Mat4::inverse() const {
    // cofactor expansion (what glam::Mat4::inverse computes, scalar form)
    const float *m = c;
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
             m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
             m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
             m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
              m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
             m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
             m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
             m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
              m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
             m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
             m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
              m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
              m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
             m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
             m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
              m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
              m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    const float rdet = 1.0f / det;
    Mat4 out;
    for (int i = 0; i < 16; i++) out.c[i] = inv[i] * rdet;
    return out;
}

// ------------------------------------------------------------------------- math
namespace math {
uint32_t mip_levels_from_size(uint32_t max_size) {
    const uint32_t l = (uint32_t)std::floor(std::log2((float)max_size)) + 1u;
    return l > 1u ? l : 1u;
}
void frustum_planes_from_matrix(const Mat4 &m, Vec4 planes[6]) {
    const Mat4 t = m.transpose();
    const Vec4 c3 = t.col(3);
    for (int i = 0; i < 6; i++) {
        const Vec4 ci = t.col(i / 2);
        const float s = (i & 1) ? -1.0f : 1.0f;
        planes[i] = {c3.x + s * ci.x, c3.y + s * ci.y, c3.z + s * ci.z, c3.w + s * ci.w};
    }
}
Vec4 normalize_plane(const Vec4 &p) {
    const float len = std::sqrt((p.x * p.x + p.y * p.y) + p.z * p.z);
    return {p.x / len, p.y / len, p.z / len, p.w / len};
}
float largest_scale_from_matrix(const Mat4 &m) {
    auto d = [&](int col) {
        const float *v = m.c + 4 * col;
        return (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    };
    return std::sqrt(std::fmax(std::fmax(d(0), d(1)), d(2)));
}
bool project_sphere_clip_space(const Vec4 &s, float znear, float p00, float p11, Vec4 *aabb) {
    const float r = s.w;
    if (s.z < r + znear) return false;
    auto bounds = [&](float c0, float c1, float *mn, float *mx) {
        const float vx = std::sqrt((c0 * c0 + c1 * c1) - r * r), vy = r;
        const float mnx = vx * c0 - vy * c1, mny = vy * c0 + vx * c1;
        const float mxx = vx * c0 + vy * c1, mxy = -vy * c0 + vx * c1;
        *mn = mnx / mny;
        *mx = mxx / mxy;
    };
    float mnx, mxx, mny, mxy;
    bounds(-s.x, -s.z, &mnx, &mxx);
    bounds(-s.y, -s.z, &mny, &mxy);
    *aabb = {mnx * p00, mny * p11, mxx * p00, mxy * p11};
    return true;
}
} // namespace math

// ------------------------------------------------------------------------- camera
namespace math {

float frustum_split(float near, float far, float lambda, float ratio) { // src/math.rs:64-69
    const float uniform = near + (far - near) * ratio;
    const float log = near * std::pow(far / near, ratio);
    return log * lambda + (1.0f - lambda) * uniform;
}

void perspective_corners(float fovy, float aspect_ratio, float near, float far, Vec4 out[8]) { // src/math.rs:149-168
    const float tan_half_h = std::tan(fovy / 2.0f) * aspect_ratio;
    const float tan_half_v = std::tan(fovy / 2.0f);
    const float xn = near * tan_half_h, yn = near * tan_half_v, xf = far * tan_half_h, yf = far * tan_half_v;
    const Vec4 c[8] = {{-xn, -yn, -near, 1.0f}, {xn, -yn, -near, 1.0f}, {xn, yn, -near, 1.0f}, {-xn, yn, -near, 1.0f},
                       {-xf, -yf, -far, 1.0f},  {xf, -yf, -far, 1.0f},  {xf, yf, -far, 1.0f},  {-xf, yf, -far, 1.0f}};
    for (int i = 0; i < 8; i++) out[i] = c[i];
}

Mat4 mat4_from_quat(const Quat &q) { // glam Mat4::from_quat (quat_to_axes)
    const float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    const float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2, yy = q.y * y2, yz = q.y * z2, zz = q.z * z2;
    const float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
    Mat4 m = Mat4::identity();
    m.c[0] = 1.0f - (yy + zz), m.c[1] = xy + wz, m.c[2] = xz - wy;
    m.c[4] = xy - wz, m.c[5] = 1.0f - (xx + zz), m.c[6] = yz + wx;
    m.c[8] = xz + wy, m.c[9] = yz - wx, m.c[10] = 1.0f - (xx + yy);
    return m;
}

Mat4 mat4_from_translation(const Vec3 &t) {
    Mat4 m = Mat4::identity();
    m.c[12] = t.x, m.c[13] = t.y, m.c[14] = t.z;
    return m;
}

Mat4 mat4_from_scale_rotation_translation(const Vec3 &s, const Quat &q, const Vec3 &t) {
    Mat4 m = mat4_from_quat(q);
    for (int r = 0; r < 3; r++) m.c[r] *= s.x, m.c[4 + r] *= s.y, m.c[8 + r] *= s.z;
    m.c[12] = t.x, m.c[13] = t.y, m.c[14] = t.z;
    return m;
}

} // namespace math

namespace camera {
Projection Projection::perspective(float fov, float near_clip) {
    Projection p;
    p.kind = Perspective;
    p.fov = fov;
    p.near_clip = near_clip;
    return p;
}
Projection Projection::orthographic(float half_width, float near_clip, float far_clip) {
    Projection p;
    p.kind = Orthographic;
    p.half_width = half_width;
    p.near_clip = near_clip;
    p.far_clip = far_clip;
    return p;
}
Mat4 Projection::compute_matrix(float aspect_ratio) const {
    if (kind == Perspective) return Mat4::perspective_infinite_reverse_rh(fov, aspect_ratio, near_clip);
    const float half_height = half_width * (1.0f / aspect_ratio);
    // reverse-Z: far and near swapped (camera.rs:91-96)
    return Mat4::orthographic_rh(-half_width, half_width, -half_height, half_height, far_clip, near_clip);
}
float Projection::z_far() const { return kind == Orthographic ? far_clip : INFINITY; }
} // namespace camera

// ------------------------------------------------------------------------- graphics
namespace graphics {
Context::Context(int device, void *stream, const OrbitCaps *caps) : stream_(stream) {
    const int32_t rc = orbit_ctx_create(device, caps, &ctx_);
    if (rc != ORBIT_OK) throw Panic(std::string("orbit_ctx_create: ") + orbit_last_error(nullptr));
}
Context::~Context() {
    for (auto &kv : transients_) (void)hipFree(kv.second.ptr);
    orbit_ctx_bind_meshlet_stream(ctx_, nullptr);
    if (meshlet_stream_) (void)orbit_ctx_bind_meshlet_stream(ctx_, nullptr);
    orbit_meshlet_stream_destroy(meshlet_stream_);
    orbit_ctx_destroy(ctx_);
}
void Context::sync_meshlet_stream(const GraphBufferHandle &meshlet_buffer, uint64_t first, uint64_t count,
                                  uint64_t capacity) {
    if (!meshlet_buffer.valid()) throw Panic("sync_meshlet_stream: no meshlet buffer");
    if (first + count > capacity) throw Panic("sync_meshlet_stream: range beyond the buffer's capacity");
    // A stream mirrors ONE buffer.  When the asset buffer was re-created (larger, or simply at another address) its
    // contents moved with it: everything uploaded so far is re-derived from the new buffer, not only the new range —
    // the library would otherwise (rightly) refuse to read the rest (ORBIT_E_RANGE).
    const bool moved = meshlet_stream_ != nullptr && meshlet_stream_source_ != meshlet_buffer.ptr;
    if (meshlet_stream_ && capacity > meshlet_stream_capacity_) {
        check(orbit_ctx_bind_meshlet_stream(ctx_, nullptr));
        check(orbit_meshlet_stream_destroy(meshlet_stream_));
        meshlet_stream_ = nullptr;
    }
    const bool fresh = meshlet_stream_ == nullptr;
    if (fresh) {
        check(orbit_meshlet_stream_create(ctx_, 0, capacity, &meshlet_stream_));
        meshlet_stream_capacity_ = capacity;
    }
    // a re-created stream remembers no materials: hand it the ones the old one mirrored BEFORE its meshlets are derived,
    // so that the update derives their alpha classes too (otherwise every later cull would read material indices)
    if (fresh && stream_materials_ != nullptr)
        check(orbit_meshlet_stream_set_materials(ctx_, meshlet_stream_, stream_materials_, stream_material_count_, stream_));
    if ((fresh || moved) && meshlet_stream_populated_ > 0)
        check(orbit_meshlet_stream_update(ctx_, meshlet_stream_, meshlet_buffer.ptr, 0, meshlet_stream_populated_, stream_));
    check(orbit_meshlet_stream_update(ctx_, meshlet_stream_, meshlet_buffer.ptr, first, count, stream_));
    meshlet_stream_source_ = meshlet_buffer.ptr;
    if (first + count > meshlet_stream_populated_) meshlet_stream_populated_ = first + count;
    check(orbit_ctx_bind_meshlet_stream(ctx_, meshlet_stream_));
}
void Context::sync_meshlet_stream_materials(const GraphBufferHandle &materials_buffer, uint32_t material_count) {
    if (!meshlet_stream_) throw Panic("sync_meshlet_stream_materials: no meshlet stream yet");
    check(orbit_meshlet_stream_set_materials(ctx_, meshlet_stream_, materials_buffer.ptr, material_count, stream_));
    stream_materials_ = materials_buffer.ptr;
    stream_material_count_ = materials_buffer.ptr ? material_count : 0u;
}
GraphBufferHandle Context::create_transient(const std::string &name, size_t size) {
    auto it = transients_.find(name);
    if (it != transients_.end() && it->second.size == size) return it->second; // name + desc cache
    if (it != transients_.end()) {
        (void)hipFree(it->second.ptr);
        transients_.erase(it);
    }
    GraphBufferHandle h;
    h.size = size;
    if (hipMalloc(&h.ptr, size) != hipSuccess) throw Panic("create_transient(" + name + "): hipMalloc failed");
    transients_[name] = h;
    return h;
}
void Context::check(int32_t rc) const {
    if (rc != ORBIT_OK) throw Panic(std::string("orbit error ") + std::to_string(rc) + ": " + orbit_last_error(ctx_));
}
} // namespace graphics

// ------------------------------------------------------------------------- passes::draw_gen
namespace passes {
using graphics::GraphBufferHandle;
using graphics::GraphImageHandle;

OrbitGpuCullInfo CullInfo::to_gpu() const {
    OrbitGpuCullInfo g;
    std::memset(&g, 0, sizeof(g)); // ..Default::default()
    std::memcpy(g.view_matrix, view_matrix.c, sizeof(g.view_matrix));
    g.cull_plane_count = (uint32_t)view_space_cull_planes.size();
    g.alpha_mode_flag = alpha_mode_filter.bits;
    g.occlusion_pass = occlusion_culling.pass_index();
    const bool has_vis = occlusion_culling.kind != OcclusionCullInfo::None;
    // bindless indices become presence flags; u32::MAX = absent (draw_gen.rs:142-144)
    g.visibility_buffer = has_vis ? 1u : ORBIT_NONE;
    g.meshlet_visibility_buffer = (has_vis && occlusion_culling.meshlet_visibility_buffer.valid()) ? 2u : ORBIT_NONE;
    g.depth_pyramid = occlusion_culling.kind == OcclusionCullInfo::VisibilityWrite ? 3u : ORBIT_NONE;
    g.min_mesh_lod = (uint32_t)lod_range_start;
    g.max_mesh_lod = (uint32_t)lod_range_end - 1u;
    g.lod_base = lod_base;
    g.lod_step = lod_step;
    g.lod_target_pos_view_space[0] = lod_target_pos_view_space.x;
    g.lod_target_pos_view_space[1] = lod_target_pos_view_space.y;
    g.lod_target_pos_view_space[2] = lod_target_pos_view_space.z;
    const size_t n = view_space_cull_planes.size() < MAX_CULL_PLANES ? view_space_cull_planes.size() : MAX_CULL_PLANES;
    for (size_t i = 0; i < n; i++) {
        const Vec4 &p = view_space_cull_planes[i];
        g.cull_planes[i][0] = p.x, g.cull_planes[i][1] = p.y, g.cull_planes[i][2] = p.z, g.cull_planes[i][3] = p.w;
    }
    g.projection_type = projection.kind == camera::Projection::Perspective ? 0u : 1u;
    if (occlusion_culling.kind == OcclusionCullInfo::VisibilityWrite) { // draw_gen.rs:170-199
        g.noskip_alphamode = occlusion_culling.noskip_alphamode.bits;
        const float aspect_ratio = occlusion_culling.aspect_ratio;
        if (projection.kind == camera::Projection::Perspective) {
            const float f = 1.0f / std::tan(0.5f * projection.fov);
            g.p00_or_width_recipx2 = f / aspect_ratio;
            g.p11_or_height_recipx2 = f;
            g.z_near = projection.near_clip;
        } else {
            const float width = projection.half_width * 2.0f;
            const float height = width * (1.0f / aspect_ratio);
            g.p00_or_width_recipx2 = (1.0f / width) * 2.0f;
            g.p11_or_height_recipx2 = (1.0f / height) * 2.0f;
            g.z_near = projection.near_clip;
            g.z_far = projection.far_clip;
        }
    }
    return g;
}

static void assert_planes(const CullInfo &ci) {
    if (ci.view_space_cull_planes.size() > MAX_CULL_PLANES) // assert!, draw_gen.rs:247,334,390
        throw Panic("assertion failed: cull_info.view_space_cull_planes.len() <= MAX_CULL_PLANES");
}

// The two stages' buffer blocks (push-constant order of entity_cull.comp:17-23 / meshlet_cull.comp:16-23, as raw
// device pointers), shared by the single-stage entry points and by create_draw_commands.
static OrbitEntityCullBufs entity_bufs(const AssetGraphData &assets, const SceneGraphData &scene,
                                       const CullInfo &cull_info, const GraphBufferHandle &dispatch) {
    OrbitEntityCullBufs b;
    std::memset(&b, 0, sizeof(b));
    b.entity_draw_buffer = scene.entity_draw_buffer.ptr;
    b.mesh_info_buffer = assets.mesh_info_buffer.ptr;
    b.meshlet_dispatch_buffer = dispatch.ptr;
    b.entity_buffer = scene.entity_buffer.ptr;
    b.visibility_buffer = (uint32_t *)cull_info.occlusion_culling.visibility_buffer.ptr;
    b.depth_pyramid = cull_info.occlusion_culling.depth_pyramid.ptr;
    b.depth_pyramid_size[0] = cull_info.occlusion_culling.depth_pyramid.width;
    b.depth_pyramid_size[1] = cull_info.occlusion_culling.depth_pyramid.height;
    // the buffer is MAX*16 bytes but records start at byte 12: one record less fits
    b.dispatch_capacity = (uint32_t)((dispatch.size - ORBIT_DISPATCH_HEADER) / 16);
    return b;
}

static OrbitMeshletCullBufs meshlet_bufs(const AssetGraphData &assets, const SceneGraphData &scene,
                                         const CullInfo &cull_info, const GraphBufferHandle &dispatch,
                                         const GraphBufferHandle &draws) {
    OrbitMeshletCullBufs b;
    std::memset(&b, 0, sizeof(b));
    b.meshlet_dispatch_buffer = dispatch.ptr;
    b.meshlet_buffer = assets.meshlet_buffer.ptr;
    b.draw_commands_buffer = draws.ptr;
    b.entity_buffer = scene.entity_buffer.ptr;
    b.material_buffer = assets.materials_buffer.ptr;
    b.meshlet_visibility_buffer = (uint32_t *)cull_info.occlusion_culling.meshlet_visibility_buffer.ptr;
    b.depth_pyramid = cull_info.occlusion_culling.depth_pyramid.ptr;
    b.depth_pyramid_size[0] = cull_info.occlusion_culling.depth_pyramid.width;
    b.depth_pyramid_size[1] = cull_info.occlusion_culling.depth_pyramid.height;
    b.dispatch_capacity = (uint32_t)((dispatch.size - ORBIT_DISPATCH_HEADER) / 16);
    b.draw_capacity = (uint32_t)((draws.size - ORBIT_DRAW_HEADER) / sizeof(OrbitMeshletDrawCommand));
    b.material_count = assets.material_count;
    return b;
}

std::pair<OrbitGpuCullInfo, GraphBufferHandle> create_meshlet_dispatch_command(
    graphics::Context &context, const std::string &name, const AssetGraphData &assets, const SceneGraphData &scene,
    const CullInfo &cull_info) {
    assert_planes(cull_info);
    const GraphBufferHandle dispatch =
        context.create_transient(name + "_meshlet_dispatch_buffer", MAX_MESHLET_DISPATCH_COUNT * 16);
    const OrbitGpuCullInfo gpu = cull_info.to_gpu();
    const OrbitEntityCullBufs b = entity_bufs(assets, scene, cull_info, dispatch);
    context.check(orbit_entity_cull(context.abi(), &gpu, &b, (uint32_t)scene.entity_draw_count, context.stream()));
    return {gpu, dispatch};
}

GraphBufferHandle create_meshlet_draw_commands(graphics::Context &context, const std::string &name,
                                               const AssetGraphData &assets, const SceneGraphData &scene,
                                               const CullInfo &cull_info, GraphBufferHandle dispatch) {
    assert_planes(cull_info);
    const GraphBufferHandle draws = context.create_transient(name, MAX_DRAW_COUNT * sizeof(OrbitMeshletDrawCommand));
    const OrbitGpuCullInfo gpu = cull_info.to_gpu();
    const OrbitMeshletCullBufs b = meshlet_bufs(assets, scene, cull_info, dispatch, draws);
    context.check(orbit_meshlet_cull(context.abi(), &gpu, &b, context.stream()));
    return draws;
}

// Both stages (draw_gen.rs:239-322: one clear + entity_cull.comp + meshlet_cull.comp) as ONE call of the library, which
// then runs a scene of the reference's size as one launch (orbit_cull_views, include/orbit_abi.h) and a larger one as
// the launch chain — the same buffers either way.
GraphBufferHandle create_draw_commands(graphics::Context &context, const std::string &name,
                                       const AssetGraphData &assets, const SceneGraphData &scene,
                                       const CullInfo &cull_info, const GraphBufferHandle *reuse_buffer) {
    assert_planes(cull_info);
    (void)reuse_buffer; // the transient cache returns the same buffer for the same name + size
    const GraphBufferHandle dispatch =
        context.create_transient(name + "_meshlet_dispatch_buffer", MAX_MESHLET_DISPATCH_COUNT * 16);
    const GraphBufferHandle draws = context.create_transient(name, MAX_DRAW_COUNT * sizeof(OrbitMeshletDrawCommand));
    const OrbitGpuCullInfo gpu = cull_info.to_gpu();
    OrbitCullView view;
    std::memset(&view, 0, sizeof(view));
    view.cull_info = &gpu;
    view.entity = entity_bufs(assets, scene, cull_info, dispatch);
    view.meshlet = meshlet_bufs(assets, scene, cull_info, dispatch, draws);
    view.entity_draw_count = (uint32_t)scene.entity_draw_count;
    context.check(orbit_cull_views(context.abi(), &view, 1, context.stream()));
    return draws;
}

static uint32_t next_power_of_two(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

DepthPyramid::DepthPyramid(graphics::Context &context, const std::string &name, uint32_t width, uint32_t height)
    : name_(name) {
    resize(context, width, height);
}
void DepthPyramid::resize(graphics::Context &context, uint32_t width, uint32_t height) {
    uint32_t w = next_power_of_two(width) / 2, h = next_power_of_two(height) / 2; // draw_gen.rs:458,485
    if (w == 0) w = 1;
    if (h == 0) h = 1;
    if (pyramid.valid() && pyramid.width == w && pyramid.height == h) return;
    OrbitDepthPyramidDesc d;
    context.check(orbit_depth_pyramid_desc_from_mip0(w, h, &d));
    const GraphBufferHandle buf = context.create_transient(name_, (size_t)d.total_texels * 4);
    pyramid.ptr = (float *)buf.ptr;
    pyramid.width = w;
    pyramid.height = h;
    pyramid.mip_levels = math::mip_levels_from_size(w > h ? w : h);
    usable = false; // draw_gen.rs:491-493
}
void DepthPyramid::update(graphics::Context &context, const GraphImageHandle &depth_buffer) {
    usable = true; // draw_gen.rs:511
    context.check(orbit_depth_reduce(context.abi(), depth_buffer.ptr, depth_buffer.width, depth_buffer.height,
                                     pyramid.ptr, context.stream()));
}
void update_multiple_depth_pyramids(graphics::Context &context, DepthPyramid *const *depth_pyramids,
                                    const GraphImageHandle *depth_buffers, size_t count) {
    // draw_gen.rs:569-628: the reference interleaves the C pyramids' dispatches level by level behind one barrier per
    // level; here all of them are one launch pair (orbit_depth_reduce_multi, batches of <= 8)
    for (size_t first = 0; first < count; first += ORBIT_MAX_PYRAMID_BATCH) {
        OrbitDepthReduceItem items[ORBIT_MAX_PYRAMID_BATCH];
        std::memset(items, 0, sizeof(items));
        const size_t n = count - first < ORBIT_MAX_PYRAMID_BATCH ? count - first : ORBIT_MAX_PYRAMID_BATCH;
        for (size_t i = 0; i < n; i++) {
            depth_pyramids[first + i]->usable = true; // :511
            items[i].depth = depth_buffers[first + i].ptr;
            items[i].screen_width = depth_buffers[first + i].width;
            items[i].screen_height = depth_buffers[first + i].height;
            items[i].pyramid = depth_pyramids[first + i]->pyramid.ptr;
        }
        context.check(orbit_depth_reduce_multi(context.abi(), items, (uint32_t)n, context.stream()));
    }
}

// ------------------------------------------------------------------------- passes::cluster
constexpr size_t MAX_ALLOCATED_DEPTH_SLICES = 4; // cluster.rs:97

void ClusterSettings::tile_counts(size_t out[2]) const {
    for (int i = 0; i < 2; i++) out[i] = (screen_resolution[i] + tile_px_size() - 1) / tile_px_size();
}
size_t ClusterSettings::linear_cluster_count() const {
    size_t t[2];
    tile_counts(t);
    return t[0] * t[1] * z_slice_count;
}
size_t ClusterSettings::linear_max_allocated_cluster_count() const {
    size_t t[2];
    tile_counts(t);
    return t[0] * t[1] * (MAX_ALLOCATED_DEPTH_SLICES > z_slice_count ? MAX_ALLOCATED_DEPTH_SLICES : z_slice_count);
}
void ClusterSettings::cluster_counts(size_t out[3]) const {
    tile_counts(out);
    out[2] = z_slice_count;
}
Vec3 screen_to_view(const Mat4 &m, const float screen_size[2], const Vec4 &screen_pos) { // cluster.rs:134-141
    const float tx = screen_pos.x / screen_size[0], ty = screen_pos.y / screen_size[1];
    const float nx = tx * 2.0f - 1.0f, ny = (1.0f - ty) * 2.0f - 1.0f;
    const Vec4 view = m.mul_vec4(Vec4{nx, ny, screen_pos.z, screen_pos.w});
    return {view.x / view.w, view.y / view.w, view.z / view.w};
}

namespace {
Vec3 line_intersection_to_z_plane(const Vec3 &a, const Vec3 &b, float z_distance) { // cluster.rs:143-148
    const Vec3 normal{0.0f, 0.0f, -1.0f};
    const Vec3 ab{b.x - a.x, b.y - a.y, b.z - a.z};
    const float na = normal.x * a.x + normal.y * a.y + normal.z * a.z;
    const float nab = normal.x * ab.x + normal.y * ab.y + normal.z * ab.z;
    const float t = (z_distance - na) / nab;
    return {a.x + t * ab.x, a.y + t * ab.y, a.z + t * ab.z};
}
} // namespace

Aabb compute_cluster_aabb(const Mat4 &matrix, const float screen_size[2], float tile_size_px,
                          const float cluster_count[3], float z_near, float z_far, const float cluster_id[3]) {
    // cluster.rs:150-184
    const Vec3 eye_pos{0.0f, 0.0f, 0.0f};
    const float min_ss[2] = {cluster_id[0] * tile_size_px, cluster_id[1] * tile_size_px};
    const float max_ss[2] = {std::fmin(min_ss[0] + tile_size_px, screen_size[0]),
                             std::fmin(min_ss[1] + tile_size_px, screen_size[1])};
    const Vec3 min_vs = screen_to_view(matrix, screen_size, Vec4{min_ss[0], min_ss[1], 1.0f, 1.0f});
    const Vec3 max_vs = screen_to_view(matrix, screen_size, Vec4{max_ss[0], max_ss[1], 1.0f, 1.0f});
    const float cluster_near = z_near * std::pow(z_far / z_near, cluster_id[2] / cluster_count[2]);
    const float cluster_far = z_near * std::pow(z_far / z_near, (cluster_id[2] + 1.0f) / cluster_count[2]);
    const Vec3 pts[4] = {line_intersection_to_z_plane(eye_pos, min_vs, cluster_near),
                         line_intersection_to_z_plane(eye_pos, min_vs, cluster_far),
                         line_intersection_to_z_plane(eye_pos, max_vs, cluster_near),
                         line_intersection_to_z_plane(eye_pos, max_vs, cluster_far)};
    Aabb out{pts[0], pts[0]};
    for (int i = 1; i < 4; i++) {
        out.min = {std::fmin(out.min.x, pts[i].x), std::fmin(out.min.y, pts[i].y), std::fmin(out.min.z, pts[i].z)};
        out.max = {std::fmax(out.max.x, pts[i].x), std::fmax(out.max.y, pts[i].y), std::fmax(out.max.z, pts[i].z)};
    }
    return out;
}

std::pair<float, float> ClusterSettings::cluster_grid_info(float near) const {
    const float far = far_plane, num_slices = (float)z_slice_count;
    const float log_f_n = std::log2(far / near);
    return {num_slices / log_f_n, -((num_slices * std::log2(near)) / log_f_n)};
}

std::pair<GraphBufferHandle, GraphBufferHandle> mark_active_clusters(graphics::Context &context,
                                                                     const ClusterSettings &settings,
                                                                     const GraphImageHandle &depth_buffer,
                                                                     const Camera &camera) {
    size_t tc[2], cc[3];
    settings.tile_counts(tc);
    settings.cluster_counts(cc);
    const GraphBufferHandle mask = context.create_transient("tile_depth_slice_mask", tc[0] * tc[1] * 4);
    const GraphBufferHandle bounds =
        context.create_transient("cluster_depth_bounds_buffer", settings.linear_cluster_count() * 8);
    const auto zz = settings.cluster_grid_info(camera.z_near());
    OrbitMarkActivePush pc;
    std::memset(&pc, 0, sizeof(pc));
    for (int i = 0; i < 3; i++) pc.cluster_count[i] = (uint32_t)cc[i];
    pc.tile_size_px = settings.tile_px_size();
    pc.screen_size[0] = depth_buffer.width; // depth_buffer_size, cluster.rs:458-465
    pc.screen_size[1] = depth_buffer.height;
    pc.z_near = camera.z_near();
    pc.z_far = settings.far_plane;
    pc.z_scale = zz.first;
    pc.z_bias = zz.second;
    pc.depth_buffer_sample_count = depth_buffer.samples;
    context.check(orbit_cluster_mark(context.abi(), &pc, depth_buffer.ptr, (uint32_t *)mask.ptr,
                                     (OrbitClusterDepthBounds *)bounds.ptr, context.stream()));
    return {mask, bounds};
}

GraphBufferHandle compact_active_clusters(graphics::Context &context, const ClusterSettings &settings,
                                          GraphBufferHandle active_cluster_mask) {
    const size_t cap = settings.linear_max_allocated_cluster_count();
    const GraphBufferHandle unique = context.create_transient("unique_cluster_buffer", 16 + cap * 4);
    size_t cc[3];
    settings.cluster_counts(cc);
    const uint32_t c3[3] = {(uint32_t)cc[0], (uint32_t)cc[1], (uint32_t)cc[2]};
    context.check(orbit_cluster_compact(context.abi(), c3, (const uint32_t *)active_cluster_mask.ptr, unique.ptr,
                                        (uint32_t)cap, context.stream()));
    return unique;
}

std::pair<GraphBufferHandle, GraphBufferHandle> cluster_light_assignment(graphics::Context &context,
                                                                         const ClusterSettings &settings,
                                                                         const Camera &camera,
                                                                         const SceneGraphData &scene,
                                                                         GraphBufferHandle unique_cluster_buffer,
                                                                         GraphBufferHandle depth_bounds_buffer) {
    size_t cc[3];
    settings.cluster_counts(cc);
    const GraphBufferHandle image = context.create_transient("cluster_offset_image", cc[0] * cc[1] * cc[2] * 8);
    const size_t index_cap = settings.linear_max_allocated_cluster_count() * 32; // cluster.rs:549
    const GraphBufferHandle index_buffer = context.create_transient("light_index_buffer", 4 + index_cap * 4);
    OrbitClusterCullInfo info;
    std::memset(&info, 0, sizeof(info));
    std::memcpy(info.world_to_view_matrix, camera.view_matrix.c, 64);
    const Mat4 inv = camera.compute_projection_matrix().inverse();
    std::memcpy(info.screen_to_view_matrix, inv.c, 64);
    for (int i = 0; i < 3; i++) info.cluster_count[i] = (uint32_t)cc[i];
    info.tile_size_px = settings.tile_px_size();
    info.screen_size[0] = settings.screen_resolution[0];
    info.screen_size[1] = settings.screen_resolution[1];
    info.z_near = camera.z_near();
    info.z_far = settings.far_plane;
    info.global_light_count = (uint32_t)scene.light_count;
    context.check(orbit_cluster_assign(context.abi(), &info, unique_cluster_buffer.ptr,
                                       (const OrbitClusterDepthBounds *)depth_bounds_buffer.ptr,
                                       (const OrbitLightData *)scene.light_data_buffer.ptr, index_buffer.ptr,
                                       (uint32_t)index_cap, (uint32_t *)image.ptr, context.stream()));
    return {image, index_buffer};
}

GraphClusterInfo compute_clusters(graphics::Context &context, const ClusterSettings &settings, const Camera &camera,
                                  const GraphImageHandle &depth_buffer, const SceneGraphData &scene) {
    GraphClusterInfo out;
    auto mb = mark_active_clusters(context, settings, depth_buffer, camera);
    out.tile_depth_slice_mask = mb.first;
    out.depth_bounds = mb.second;
    out.unique_cluster_buffer = compact_active_clusters(context, settings, mb.first);
    auto il = cluster_light_assignment(context, settings, camera, scene, out.unique_cluster_buffer, mb.second);
    out.light_offset_image = il.first;
    out.light_index_list = il.second;
    // GpuClusterInfoBuffer::new, cluster.rs:337-360
    size_t cc[3];
    settings.cluster_counts(cc);
    const auto zz = settings.cluster_grid_info(camera.z_near());
    std::memset(&out.info, 0, sizeof(out.info));
    for (int i = 0; i < 3; i++) out.info.cluster_count[i] = (uint32_t)cc[i];
    out.info.tile_size_px = settings.tile_px_size();
    out.info.screen_size[0] = settings.screen_resolution[0];
    out.info.screen_size[1] = settings.screen_resolution[1];
    out.info.z_slice_count = settings.z_slice_count;
    out.info.z_scale = zz.first;
    out.info.z_bias = zz.second;
    out.info.luminance_cutoff = settings.luminance_cutoff;
    return out;
}

Mat4 Transform::compute_matrix() const { return math::mat4_from_scale_rotation_translation(scale, orientation, position); }

Mat4 SceneCamera::compute_matrix() const { // camera.rs:124-129
    return projection.compute_matrix(aspect_ratio).mul(transform.compute_matrix().inverse());
}

ShadowCascade shadow_cascade(const ShadowSettings &settings, const LodSettings &lod, const Quat &direction,
                             const SceneCamera &camera, size_t cascade_index, bool frustum_culling) {
    if (camera.projection.kind != camera::Projection::Perspective) throw Panic("not yet implemented"); // todo!(), :468
    if (cascade_index >= MAX_SHADOW_CASCADE_COUNT) throw Panic("cascade_index out of range");
    const float fov = camera.projection.fov, near_clip_cam = camera.projection.near_clip;
    const float far_clip_cam = settings.max_shadow_distance;
    const float lambda = settings.cascade_split_lambda;
    const float near_split_ratio = (float)cascade_index / (float)MAX_SHADOW_CASCADE_COUNT;
    const float far_split_ratio = (float)(cascade_index + 1) / (float)MAX_SHADOW_CASCADE_COUNT;
    const float near = math::frustum_split(near_clip_cam, far_clip_cam, lambda, near_split_ratio);
    const float far = math::frustum_split(near_clip_cam, far_clip_cam, lambda, far_split_ratio);

    Mat4 light_matrix = math::mat4_from_quat(direction.inverse()); // :461,479
    const Mat4 view_to_world = camera.transform.compute_matrix();
    const Mat4 view_to_light = light_matrix.mul(view_to_world);

    Vec4 corners[8];
    math::perspective_corners(fov, camera.aspect_ratio, near, far, corners);
    float center[4] = {0, 0, 0, 0}, mn[3], mx[3];
    for (int i = 0; i < 8; i++) { // :483-497
        Vec4 v = view_to_light.mul_vec4(corners[i]);
        v = {v.x / v.w, v.y / v.w, v.z / v.w, v.w / v.w};
        corners[i] = v;
        const float c[3] = {v.x, v.y, v.z};
        for (int k = 0; k < 3; k++) {
            if (i == 0) mn[k] = mx[k] = c[k];
            mn[k] = std::fmin(mn[k], c[k]);
            mx[k] = std::fmax(mx[k], c[k]);
        }
        center[0] += v.x, center[1] += v.y, center[2] += v.z, center[3] += v.w;
    }
    for (float &c : center) c /= 8.0f;
    float radius_sqr = 0.0f;
    for (int i = 0; i < 8; i++) { // :499-506
        const float dx = corners[i].x - center[0], dy = corners[i].y - center[1], dz = corners[i].z - center[2];
        radius_sqr = std::fmax(radius_sqr, dx * dx + dy * dy + dz * dz);
    }
    const float radius = std::sqrt(radius_sqr);

    // forward offset (:513-527): lerp(min, max, (sign + 1) / 2) - radius * sign per component
    const float fs[3] = {view_to_light.c[8], view_to_light.c[9], view_to_light.c[10]}; // z_axis
    float modified[3];
    const float texel = radius * 2.0f / (float)settings.shadow_resolution; // :529
    for (int k = 0; k < 3; k++) {
        const float lo = mn[k] - center[k], hi = mx[k] - center[k];
        const float a = (fs[k] + 1.0f) / 2.0f;
        const float offset = (lo + (hi - lo) * a) - radius * fs[k];
        modified[k] = std::floor((center[k] + offset) / texel) * texel; // :534-535
    }
    light_matrix = math::mat4_from_translation({-modified[0], -modified[1], -modified[2]}).mul(light_matrix); // :543
    const float max_extent = radius, min_extent = -radius;
    const float near_clip = min_extent - 80.0f, far_clip = max_extent; // :545-546

    ShadowCascade out;
    out.shadow_map_world_size = radius * 2.0f;
    out.light_projection_matrix = // :548-555 (reverse z: far and near swapped)
        Mat4::orthographic_rh(min_extent, max_extent, min_extent, max_extent, far_clip, near_clip).mul(light_matrix);

    // culling planes (:623-651)
    Vec4 light_planes[6], cam_planes[6];
    math::frustum_planes_from_matrix(
        Mat4::orthographic_rh(min_extent, max_extent, min_extent, max_extent, near_clip, far_clip), light_planes);
    const Mat4 light_to_world = light_matrix.inverse();
    const Mat4 camera_clip_to_light = camera.compute_matrix().mul(light_to_world);
    math::frustum_planes_from_matrix(camera_clip_to_light, cam_planes);
    CullInfo &ci = out.cull_info;
    if (frustum_culling) { // :685
        for (int i = 0; i < 6; i++) ci.view_space_cull_planes.push_back(math::normalize_plane(light_planes[i]));
        for (int i = 0; i < 5; i++) { // .take(5).filter(dot(plane.xyz, Z) >= 0)
            const Vec4 p = math::normalize_plane(cam_planes[i]);
            if (p.z >= 0.0f) ci.view_space_cull_planes.push_back(p);
        }
    }
    ci.view_matrix = light_matrix;
    ci.projection = camera::Projection::orthographic(radius, near_clip, far_clip); // :687-691
    ci.occlusion_culling = OcclusionCullInfo{};
    ci.alpha_mode_filter.bits = AlphaModeFlags::OPAQUE | AlphaModeFlags::MASKED;
    if (cascade_index <= 1) { // :699-702
        ci.lod_range_start = lod.min_mesh_lod, ci.lod_range_end = lod.max_mesh_lod + 1;
    } else {
        ci.lod_range_start = 2, ci.lod_range_end = lod.max_mesh_lod + 1;
    }
    ci.lod_base = lod.lod_base, ci.lod_step = lod.lod_step;
    const Vec3 &cp = camera.transform.position; // light_matrix.transform_point3(camera position), :705
    const Vec4 t = light_matrix.mul_vec4({cp.x, cp.y, cp.z, 1.0f});
    ci.lod_target_pos_view_space = {t.x, t.y, t.z};
    return out;
}

} // namespace passes
} // namespace orbit
