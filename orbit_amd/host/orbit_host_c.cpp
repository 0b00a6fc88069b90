// orbit_host_c.cpp — flat C exports of the C++ host mirror (see orbit_host_c.h).
#include "orbit_host_c.h"

#include <cstring>
#include <map>
#include <memory>
#include <string>

#include "orbit_host.hpp"
#include "orbit_assets.hpp"
#include "orbit_gltf.hpp"
#include "orbit_scene.hpp"

using namespace orbit;
using namespace orbit::passes;

namespace {
thread_local std::string g_err = "no error";

struct HostCtx {
    std::unique_ptr<graphics::Context> ctx;
    std::map<std::string, std::unique_ptr<DepthPyramid>> pyramids;
};

camera::Projection to_projection(const OrbitHostProjection &p) {
    return p.kind == 0 ? camera::Projection::perspective(p.fov, p.near_clip)
                       : camera::Projection::orthographic(p.half_width, p.near_clip, p.far_clip);
}

CullInfo to_cull_info(const OrbitHostCullInfo &h) {
    CullInfo ci;
    ci.view_matrix = Mat4::from_cols_array(h.view_matrix);
    for (uint32_t i = 0; i < h.plane_count && i < 16; i++)
        ci.view_space_cull_planes.push_back({h.view_space_cull_planes[i][0], h.view_space_cull_planes[i][1],
                                             h.view_space_cull_planes[i][2], h.view_space_cull_planes[i][3]});
    ci.projection = to_projection(h.projection);
    ci.occlusion_culling.kind = (OcclusionCullInfo::Kind)h.occlusion_kind;
    ci.occlusion_culling.visibility_buffer.ptr = h.visibility_buffer;
    ci.occlusion_culling.meshlet_visibility_buffer.ptr = h.meshlet_visibility_buffer;
    ci.occlusion_culling.depth_pyramid.ptr = h.depth_pyramid;
    ci.occlusion_culling.depth_pyramid.width = h.depth_pyramid_size[0];
    ci.occlusion_culling.depth_pyramid.height = h.depth_pyramid_size[1];
    ci.occlusion_culling.noskip_alphamode.bits = h.noskip_alphamode;
    ci.occlusion_culling.aspect_ratio = h.aspect_ratio;
    ci.alpha_mode_filter.bits = h.alpha_mode_filter;
    ci.lod_range_start = h.lod_range[0];
    ci.lod_range_end = h.lod_range[1];
    ci.lod_base = h.lod_base;
    ci.lod_step = h.lod_step;
    ci.lod_target_pos_view_space = {h.lod_target_pos_view_space[0], h.lod_target_pos_view_space[1],
                                    h.lod_target_pos_view_space[2]};
    return ci;
}
AssetGraphData to_assets(const OrbitHostAssets &a) {
    AssetGraphData o;
    o.meshlet_buffer.ptr = a.meshlet_buffer;
    o.mesh_info_buffer.ptr = a.mesh_info_buffer;
    o.materials_buffer.ptr = a.materials_buffer;
    o.material_count = a.material_count;
    return o;
}
SceneGraphData to_scene(const OrbitHostScene &s) {
    SceneGraphData o;
    o.entity_draw_count = s.entity_draw_count;
    o.entity_draw_buffer.ptr = s.entity_draw_buffer;
    o.entity_buffer.ptr = s.entity_buffer;
    o.light_data_buffer.ptr = s.light_data_buffer;
    o.meshlet_visibility_buffer.ptr = s.meshlet_visibility_buffer;
    o.light_count = s.light_count;
    return o;
}
OrbitHostBuffer to_buf(const graphics::GraphBufferHandle &h) { return {h.ptr, (uint64_t)h.size}; }

template <class F>
int32_t guarded(F &&f) {
    try {
        f();
        return ORBIT_OK;
    } catch (const Panic &p) {
        g_err = p.what();
        return ORBIT_HOST_PANIC;
    } catch (const std::exception &e) {
        g_err = e.what();
        return ORBIT_HOST_PANIC;
    }
}
} // namespace

namespace {
collections::Index to_index(OrbitHostArenaIndex i) { return {i.generation, i.slot}; }
OrbitHostArenaIndex from_index(collections::Index i) { return {i.generation, i.slot}; }
Quat to_quat(const float q[4]) {
    Quat o;
    o.x = q[0], o.y = q[1], o.z = q[2], o.w = q[3];
    return o;
}
passes::Transform to_transform(const float position[3], const float orientation[4], const float scale[3]) {
    passes::Transform t;
    t.position = {position[0], position[1], position[2]};
    t.orientation = to_quat(orientation);
    t.scale = {scale[0], scale[1], scale[2]};
    return t;
}
} // namespace

extern "C" {

const char *orbit_host_last_error(void) { return g_err.c_str(); }

void orbit_host_perspective_infinite_reverse_rh(float fov_y, float aspect, float z_near, float out[16]) {
    std::memcpy(out, Mat4::perspective_infinite_reverse_rh(fov_y, aspect, z_near).c, 64);
}
void orbit_host_orthographic_rh(float l, float r, float b, float t, float n, float f, float out[16]) {
    std::memcpy(out, Mat4::orthographic_rh(l, r, b, t, n, f).c, 64);
}
void orbit_host_projection_compute_matrix(const OrbitHostProjection *p, float aspect, float out[16]) {
    std::memcpy(out, to_projection(*p).compute_matrix(aspect).c, 64);
}
void orbit_host_mat4_inverse(const float in[16], float out[16]) {
    std::memcpy(out, Mat4::from_cols_array(in).inverse().c, 64);
}
void orbit_host_frustum_planes_from_matrix(const float m[16], float out[6][4], int32_t normalize) {
    Vec4 pl[6];
    math::frustum_planes_from_matrix(Mat4::from_cols_array(m), pl);
    for (int i = 0; i < 6; i++) {
        const Vec4 p = normalize ? math::normalize_plane(pl[i]) : pl[i];
        out[i][0] = p.x, out[i][1] = p.y, out[i][2] = p.z, out[i][3] = p.w;
    }
}
uint32_t orbit_host_mip_levels_from_size(uint32_t max_size) { return math::mip_levels_from_size(max_size); }
int32_t orbit_host_project_sphere_clip_space(const float s[4], float znear, float p00, float p11, float aabb[4]) {
    Vec4 a;
    if (!math::project_sphere_clip_space({s[0], s[1], s[2], s[3]}, znear, p00, p11, &a)) return 0;
    aabb[0] = a.x, aabb[1] = a.y, aabb[2] = a.z, aabb[3] = a.w;
    return 1;
}
int32_t orbit_host_cull_info_to_gpu(const OrbitHostCullInfo *ci, OrbitGpuCullInfo *out) {
    return guarded([&] { *out = to_cull_info(*ci).to_gpu(); });
}
int32_t orbit_host_shadow_cascade(const OrbitHostShadowCascadeIn *in, OrbitHostCullInfo *h, float out_lpm[16],
                                  float *out_world_size) {
    return guarded([&] {
        ShadowSettings ss;
        ss.shadow_resolution = in->shadow_resolution;
        ss.cascade_split_lambda = in->cascade_split_lambda;
        ss.max_shadow_distance = in->max_shadow_distance;
        LodSettings lod;
        lod.min_mesh_lod = in->min_mesh_lod, lod.max_mesh_lod = in->max_mesh_lod;
        lod.lod_base = in->lod_base, lod.lod_step = in->lod_step;
        SceneCamera cam;
        cam.transform.position = {in->camera_position[0], in->camera_position[1], in->camera_position[2]};
        cam.transform.orientation = {in->camera_orientation[0], in->camera_orientation[1], in->camera_orientation[2],
                                     in->camera_orientation[3]};
        cam.projection = camera::Projection::perspective(in->camera_fov, in->camera_near_clip);
        cam.aspect_ratio = in->camera_aspect_ratio;
        const Quat dir{in->direction[0], in->direction[1], in->direction[2], in->direction[3]};
        const ShadowCascade c = shadow_cascade(ss, lod, dir, cam, in->cascade_index, in->frustum_culling != 0);
        memset(h, 0, sizeof(*h));
        memcpy(h->view_matrix, c.cull_info.view_matrix.c, sizeof(h->view_matrix));
        h->plane_count = (uint32_t)c.cull_info.view_space_cull_planes.size();
        for (uint32_t i = 0; i < h->plane_count && i < 16; i++) {
            const Vec4 &p = c.cull_info.view_space_cull_planes[i];
            h->view_space_cull_planes[i][0] = p.x, h->view_space_cull_planes[i][1] = p.y;
            h->view_space_cull_planes[i][2] = p.z, h->view_space_cull_planes[i][3] = p.w;
        }
        h->projection.kind = 1;
        h->projection.half_width = c.cull_info.projection.half_width;
        h->projection.near_clip = c.cull_info.projection.near_clip;
        h->projection.far_clip = c.cull_info.projection.far_clip;
        h->occlusion_kind = 0;
        h->aspect_ratio = 1.0f;
        h->alpha_mode_filter = c.cull_info.alpha_mode_filter.bits;
        h->lod_range[0] = (uint32_t)c.cull_info.lod_range_start, h->lod_range[1] = (uint32_t)c.cull_info.lod_range_end;
        h->lod_base = c.cull_info.lod_base, h->lod_step = c.cull_info.lod_step;
        h->lod_target_pos_view_space[0] = c.cull_info.lod_target_pos_view_space.x;
        h->lod_target_pos_view_space[1] = c.cull_info.lod_target_pos_view_space.y;
        h->lod_target_pos_view_space[2] = c.cull_info.lod_target_pos_view_space.z;
        memcpy(out_lpm, c.light_projection_matrix.c, 16 * sizeof(float));
        *out_world_size = c.shadow_map_world_size;
    });
}
void orbit_host_cluster_settings_derive(const OrbitHostClusterSettings *s, float z_near, OrbitHostClusterDerived *out) {
    ClusterSettings cs;
    cs.px_size_power = s->px_size_power;
    cs.set_resolution(s->screen_resolution[0], s->screen_resolution[1]);
    cs.z_slice_count = s->z_slice_count;
    cs.far_plane = s->far_plane;
    cs.luminance_cutoff = s->luminance_cutoff;
    size_t t[2], c[3];
    cs.tile_counts(t);
    cs.cluster_counts(c);
    out->tile_counts[0] = t[0], out->tile_counts[1] = t[1];
    for (int i = 0; i < 3; i++) out->cluster_counts[i] = c[i];
    out->linear_cluster_count = cs.linear_cluster_count();
    out->linear_max_allocated_cluster_count = cs.linear_max_allocated_cluster_count();
    out->tile_px_size = cs.tile_px_size();
    const auto zz = cs.cluster_grid_info(z_near);
    out->z_scale = zz.first, out->z_bias = zz.second;
}

void orbit_host_compute_cluster_aabb(const float inverse_projection[16], const float screen_size[2], float tile_size_px,
                                     const float cluster_count[3], float z_near, float z_far, const float cluster_id[3],
                                     float out[6]) {
    const Aabb a = compute_cluster_aabb(Mat4::from_cols_array(inverse_projection), screen_size, tile_size_px,
                                        cluster_count, z_near, z_far, cluster_id);
    out[0] = a.min.x, out[1] = a.min.y, out[2] = a.min.z, out[3] = a.max.x, out[4] = a.max.y, out[5] = a.max.z;
}

void *orbit_host_context_create(int32_t device, void *stream, const OrbitCaps *caps) {
    HostCtx *h = nullptr;
    const int32_t rc = guarded([&] {
        h = new HostCtx();
        h->ctx.reset(new graphics::Context(device, stream, caps));
    });
    if (rc != ORBIT_OK) {
        delete h;
        return nullptr;
    }
    return h;
}
void orbit_host_context_destroy(void *hctx) { delete (HostCtx *)hctx; }

int32_t orbit_host_sync_meshlet_stream(void *hctx, void *meshlet_buffer, uint64_t first, uint64_t count,
                                       uint64_t capacity) {
    return guarded([&] {
        graphics::GraphBufferHandle h;
        h.ptr = meshlet_buffer;
        h.size = (size_t)capacity * sizeof(OrbitMeshlet);
        ((HostCtx *)hctx)->ctx->sync_meshlet_stream(h, first, count, capacity);
    });
}

int32_t orbit_host_sync_meshlet_stream_materials(void *hctx, void *materials_buffer, uint32_t material_count) {
    return guarded([&] {
        graphics::GraphBufferHandle h;
        h.ptr = materials_buffer;
        h.size = (size_t)material_count * sizeof(OrbitMaterialData);
        ((HostCtx *)hctx)->ctx->sync_meshlet_stream_materials(h, material_count);
    });
}

int32_t orbit_host_create_meshlet_dispatch_command(void *hctx, const char *name, const OrbitHostAssets *assets,
                                                   const OrbitHostScene *scene, const OrbitHostCullInfo *ci,
                                                   OrbitGpuCullInfo *out_cull_info, OrbitHostBuffer *out_dispatch) {
    return guarded([&] {
        auto r = create_meshlet_dispatch_command(*((HostCtx *)hctx)->ctx, name, to_assets(*assets), to_scene(*scene),
                                                 to_cull_info(*ci));
        if (out_cull_info) *out_cull_info = r.first;
        *out_dispatch = to_buf(r.second);
    });
}
int32_t orbit_host_create_meshlet_draw_commands(void *hctx, const char *name, const OrbitHostAssets *assets,
                                                const OrbitHostScene *scene, const OrbitHostCullInfo *ci,
                                                const OrbitHostBuffer *dispatch, OrbitHostBuffer *out_draws) {
    return guarded([&] {
        graphics::GraphBufferHandle d;
        d.ptr = dispatch->ptr;
        d.size = dispatch->size;
        *out_draws = to_buf(create_meshlet_draw_commands(*((HostCtx *)hctx)->ctx, name, to_assets(*assets),
                                                         to_scene(*scene), to_cull_info(*ci), d));
    });
}
int32_t orbit_host_create_draw_commands(void *hctx, const char *name, const OrbitHostAssets *assets,
                                        const OrbitHostScene *scene, const OrbitHostCullInfo *ci,
                                        OrbitHostBuffer *out_draws) {
    return guarded([&] {
        *out_draws = to_buf(create_draw_commands(*((HostCtx *)hctx)->ctx, name, to_assets(*assets), to_scene(*scene),
                                                 to_cull_info(*ci), nullptr));
    });
}
int32_t orbit_host_depth_pyramid_update(void *hctx, const char *name, const float *depth, uint32_t width,
                                        uint32_t height, float **out_pyramid, uint32_t out_size_mips[3]) {
    return guarded([&] {
        HostCtx *h = (HostCtx *)hctx;
        auto &p = h->pyramids[name];
        if (!p) p.reset(new DepthPyramid(*h->ctx, name, width, height));
        p->resize(*h->ctx, width, height);
        graphics::GraphImageHandle d;
        d.ptr = const_cast<float *>(depth);
        d.width = width;
        d.height = height;
        p->update(*h->ctx, d);
        const graphics::GraphImageHandle cur = p->get_current();
        *out_pyramid = cur.ptr;
        out_size_mips[0] = cur.width, out_size_mips[1] = cur.height, out_size_mips[2] = cur.mip_levels;
    });
}
int32_t orbit_host_compute_clusters(void *hctx, const OrbitHostClusterSettings *s, const float view_matrix[16],
                                    const OrbitHostProjection *projection, float aspect_ratio, const float *depth,
                                    uint32_t depth_size[2], uint32_t samples, const OrbitHostScene *scene,
                                    OrbitHostClusterOut *out) {
    return guarded([&] {
        ClusterSettings cs;
        cs.px_size_power = s->px_size_power;
        cs.set_resolution(s->screen_resolution[0], s->screen_resolution[1]);
        cs.z_slice_count = s->z_slice_count;
        cs.far_plane = s->far_plane;
        cs.luminance_cutoff = s->luminance_cutoff;
        Camera cam;
        cam.view_matrix = Mat4::from_cols_array(view_matrix);
        cam.projection = to_projection(*projection);
        cam.aspect_ratio = aspect_ratio;
        graphics::GraphImageHandle d;
        d.ptr = const_cast<float *>(depth);
        d.width = depth_size[0];
        d.height = depth_size[1];
        d.samples = samples;
        const GraphClusterInfo r = compute_clusters(*((HostCtx *)hctx)->ctx, cs, cam, d, to_scene(*scene));
        out->tile_depth_slice_mask = to_buf(r.tile_depth_slice_mask);
        out->depth_bounds = to_buf(r.depth_bounds);
        out->unique_cluster_buffer = to_buf(r.unique_cluster_buffer);
        out->light_offset_image = to_buf(r.light_offset_image);
        out->light_index_list = to_buf(r.light_index_list);
        out->info = r.info;
    });
}

// ---------------------------------------------------------------- scene side (orbit_scene.hpp), host only
void *orbit_host_arena_create(void) { return new collections::Arena<int64_t>(); }
void orbit_host_arena_destroy(void *arena) { delete (collections::Arena<int64_t> *)arena; }
OrbitHostArenaIndex orbit_host_arena_insert(void *arena, int64_t value) {
    return from_index(((collections::Arena<int64_t> *)arena)->insert(value));
}
int32_t orbit_host_arena_get(const void *arena, OrbitHostArenaIndex index, int64_t *out) {
    const int64_t *v = ((const collections::Arena<int64_t> *)arena)->get(to_index(index));
    if (v && out) *out = *v;
    return v ? 1 : 0;
}
int32_t orbit_host_arena_remove(void *arena, OrbitHostArenaIndex index, int64_t *out) {
    const std::optional<int64_t> v = ((collections::Arena<int64_t> *)arena)->remove(to_index(index));
    if (v && out) *out = *v;
    return v ? 1 : 0;
}
int32_t orbit_host_arena_has_index(const void *arena, OrbitHostArenaIndex index) {
    return ((const collections::Arena<int64_t> *)arena)->has_index(to_index(index)) ? 1 : 0;
}
uint64_t orbit_host_arena_len(const void *arena) { return ((const collections::Arena<int64_t> *)arena)->len(); }
uint64_t orbit_host_arena_iter(const void *arena, OrbitHostArenaIndex *indices, int64_t *values, uint64_t capacity) {
    uint64_t n = 0;
    ((const collections::Arena<int64_t> *)arena)->for_each([&](collections::Index i, const int64_t &v) {
        if (n < capacity) {
            indices[n] = from_index(i);
            values[n] = v;
            n++;
        }
    });
    return n;
}

void *orbit_host_freelist_create(uint64_t size) { return new collections::FreeListAllocator((size_t)size); }
void orbit_host_freelist_destroy(void *alloc) { delete (collections::FreeListAllocator *)alloc; }
int32_t orbit_host_freelist_allocate(void *alloc, uint64_t size, OrbitHostArenaIndex *out_index, uint64_t out_range[2]) {
    const auto r = ((collections::FreeListAllocator *)alloc)->allocate((size_t)size);
    if (!r) return 0;
    *out_index = from_index(r->first);
    out_range[0] = r->second.start;
    out_range[1] = r->second.end;
    return 1;
}
void orbit_host_freelist_deallocate(void *alloc, OrbitHostArenaIndex index) {
    ((collections::FreeListAllocator *)alloc)->deallocate(to_index(index));
}
uint64_t orbit_host_freelist_blocks(const void *alloc, uint64_t *ranges, int32_t *is_free, uint64_t capacity) {
    const auto blocks = ((const collections::FreeListAllocator *)alloc)->blocks_in_address_order();
    for (uint64_t i = 0; i < blocks.size() && i < capacity; i++) {
        ranges[2 * i] = blocks[i].first.start;
        ranges[2 * i + 1] = blocks[i].first.end;
        is_free[i] = blocks[i].second ? 1 : 0;
    }
    return blocks.size();
}

void *orbit_host_scene_create(uint64_t visibility_chunk_count) {
    return visibility_chunk_count ? new scene::SceneData((size_t)visibility_chunk_count) : new scene::SceneData();
}
void orbit_host_scene_destroy(void *sc) { delete (scene::SceneData *)sc; }
int64_t orbit_host_scene_add_entity(void *sc, const OrbitHostEntity *e) {
    if (!sc || !e) return -1;
    scene::EntityData d;
    d.transform = to_transform(e->position, e->orientation, e->scale);
    if (e->name) d.name = e->name, d.has_name = true;
    if (e->mesh >= 0) d.mesh = (uint32_t)e->mesh;
    if (e->has_light) {
        scene::Light l;
        l.color = {e->light.color[0], e->light.color[1], e->light.color[2]};
        l.intensity = e->light.intensity;
        l.kind = (scene::LightKind)e->light.kind;
        l.param = e->light.param;
        l.irradiance_map_index = e->light.irradiance_map_index;
        l.prefiltered_map_index = e->light.prefiltered_map_index;
        l.cast_shadows = e->light.cast_shadows != 0;
        d.light = l;
    }
    return (int64_t)((scene::SceneData *)sc)->add_entity(std::move(d));
}
int32_t orbit_host_scene_set_transform(void *sc, uint64_t entity, const float position[3], const float orientation[4],
                                       const float scale[3]) {
    return guarded([&] {
        scene::SceneData *s = (scene::SceneData *)sc;
        if (entity >= s->entities.size()) throw Panic("set_transform: entity index out of range");
        s->entities[entity].transform = to_transform(position, orientation, scale);
    });
}
int32_t orbit_host_scene_update(void *sc, const OrbitMeshInfo *mesh_infos, uint64_t mesh_info_count,
                                float luminance_cutoff, uint64_t frame_index) {
    return guarded([&] {
        ((scene::SceneData *)sc)->update_scene(mesh_infos, (size_t)mesh_info_count, luminance_cutoff, (size_t)frame_index);
    });
}
const void *orbit_host_scene_entity_draws(const void *sc, uint64_t *count) {
    const scene::SceneData *s = (const scene::SceneData *)sc;
    *count = s->entity_draw_cache.size();
    return s->entity_draw_cache.data();
}
const OrbitEntityData *orbit_host_scene_entity_data(const void *sc, uint64_t *count) {
    const scene::SceneData *s = (const scene::SceneData *)sc;
    *count = s->entity_data_cache.size();
    return s->entity_data_cache.data();
}
const OrbitLightData *orbit_host_scene_light_data(const void *sc, uint64_t *count) {
    const scene::SceneData *s = (const scene::SceneData *)sc;
    *count = s->light_data_cache.size();
    return s->light_data_cache.data();
}
uint64_t orbit_host_scene_shadow_command_count(const void *sc) {
    return ((const scene::SceneData *)sc)->shadow_commands.size();
}

// ---------------------------------------------------------------- asset side (orbit_assets.hpp), host only
int32_t orbit_host_compute_meshlets(const float *positions, uint64_t vertex_count, const uint32_t *indices,
                                    uint64_t index_count, uint32_t material, uint32_t vertex_offset,
                                    uint32_t data_offset_base, OrbitMeshlet *out_meshlets, uint32_t *out_meshlet_data,
                                    uint64_t *meshlet_count, uint64_t *meshlet_data_words) {
    return guarded([&] {
        std::vector<uint32_t> data(data_offset_base, 0u); // earlier meshes' data: only its length matters here
        std::vector<OrbitMeshlet> meshlets;
        assets::compute_meshlets(positions, (size_t)vertex_count, indices, (size_t)index_count, (uint16_t)material,
                                 vertex_offset, data, meshlets);
        *meshlet_count = meshlets.size();
        *meshlet_data_words = data.size() - data_offset_base;
        if (out_meshlets && out_meshlet_data) {
            std::memcpy(out_meshlets, meshlets.data(), meshlets.size() * sizeof(OrbitMeshlet));
            std::memcpy(out_meshlet_data, data.data() + data_offset_base, (data.size() - data_offset_base) * 4);
        }
    });
}
void orbit_host_compute_mesh_bounds(const float *positions, uint64_t vertex_count, float aabb_min[3], float aabb_max[3],
                                    float bounding_sphere[4]) {
    assets::compute_mesh_bounds(positions, (size_t)vertex_count, aabb_min, aabb_max, bounding_sphere);
}

// ---------------------------------------------------------------- asset ingestion (orbit_gltf.hpp), host only
void *orbit_host_gltf_load(const char *path) {
    gltf_loader::LoadedScene *scene = nullptr;
    const int32_t rc = guarded([&] {
        if (!path) throw Panic("path is NULL");
        auto s = std::make_unique<gltf_loader::LoadedScene>();
        std::string err;
        if (!gltf_loader::load_gltf(path, *s, err)) throw Panic("load_gltf: " + err);
        scene = s.release();
    });
    return rc == 0 ? scene : nullptr;
}
void orbit_host_gltf_free(void *scene) { delete static_cast<gltf_loader::LoadedScene *>(scene); }
void orbit_host_gltf_counts(const void *scene, OrbitHostGltfCounts *out) {
    const auto &s = *static_cast<const gltf_loader::LoadedScene *>(scene);
    out->meshes = s.assets.mesh_infos.size();
    out->meshlets = s.assets.meshlets.size();
    out->meshlet_data_words = s.assets.meshlet_data.size();
    out->materials = s.assets.materials.size();
    out->vertices = s.assets.vertex_positions.size() / 3;
    out->entities = s.entities.size();
}
const OrbitMeshInfo *orbit_host_gltf_mesh_infos(const void *scene) {
    return static_cast<const gltf_loader::LoadedScene *>(scene)->assets.mesh_infos.data();
}
const OrbitMeshlet *orbit_host_gltf_meshlets(const void *scene) {
    return static_cast<const gltf_loader::LoadedScene *>(scene)->assets.meshlets.data();
}
const uint32_t *orbit_host_gltf_meshlet_data(const void *scene) {
    return static_cast<const gltf_loader::LoadedScene *>(scene)->assets.meshlet_data.data();
}
const OrbitMaterialData *orbit_host_gltf_materials(const void *scene) {
    return static_cast<const gltf_loader::LoadedScene *>(scene)->assets.materials.data();
}
const float *orbit_host_gltf_vertex_positions(const void *scene) {
    return static_cast<const gltf_loader::LoadedScene *>(scene)->assets.vertex_positions.data();
}
uint64_t orbit_host_gltf_entities(const void *scene, OrbitHostGltfEntity *out, uint64_t capacity) {
    const auto &s = *static_cast<const gltf_loader::LoadedScene *>(scene);
    for (size_t i = 0; i < s.entities.size() && i < capacity; i++) {
        const auto &e = s.entities[i];
        out[i].mesh = e.mesh;
        const float p[3] = {e.transform.position.x, e.transform.position.y, e.transform.position.z};
        const float q[4] = {e.transform.orientation.x, e.transform.orientation.y, e.transform.orientation.z,
                            e.transform.orientation.w};
        const float sc[3] = {e.transform.scale.x, e.transform.scale.y, e.transform.scale.z};
        std::memcpy(out[i].position, p, 12);
        std::memcpy(out[i].orientation, q, 16);
        std::memcpy(out[i].scale, sc, 12);
    }
    return s.entities.size();
}
uint64_t orbit_host_simplify_clustered(const float *positions, uint64_t vertex_count, const uint32_t *indices,
                                       uint64_t index_count, uint64_t target_index_count, uint32_t *out,
                                       uint64_t capacity, uint64_t *needed) {
    std::vector<uint32_t> r;
    assets::simplify_clustered(positions, (size_t)vertex_count, indices, (size_t)index_count,
                               (size_t)target_index_count, r);
    if (needed) *needed = r.size();
    const size_t n = std::min<size_t>(r.size(), (size_t)capacity) / 3 * 3;
    if (out && n) std::memcpy(out, r.data(), n * 4);
    return n;
}
void orbit_host_transform_from_mat4(const float matrix[16], float position[3], float orientation[4], float scale[3]) {
    const passes::Transform t = gltf_loader::transform_from_mat4(Mat4::from_cols_array(matrix));
    position[0] = t.position.x, position[1] = t.position.y, position[2] = t.position.z;
    orientation[0] = t.orientation.x, orientation[1] = t.orientation.y, orientation[2] = t.orientation.z;
    orientation[3] = t.orientation.w;
    scale[0] = t.scale.x, scale[1] = t.scale.y, scale[2] = t.scale.z;
}

} // extern "C"
