// orbit_scene.cpp — see orbit_scene.hpp.  Reference: src/scene.rs, src/collections/{arena,freelist_alloc}.rs.
#include "orbit_scene.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace orbit {
namespace collections {

FreeListAllocator::FreeListAllocator(size_t size) { // freelist_alloc.rs:27-38
    blocks_.insert(Block{true, BlockRange{0, size}, std::nullopt, std::nullopt});
}

std::optional<std::pair<Index, BlockRange>> FreeListAllocator::allocate(size_t size) {
    // :41-45 — iter().filter(free && fits).min_by_key(size): the FIRST of equally small blocks in slot order
    std::optional<Index> best;
    size_t best_size = 0;
    blocks_.for_each([&](Index i, const Block &b) {
        if (!b.free || b.range.size() < size) return;
        if (!best || b.range.size() < best_size) {
            best = i;
            best_size = b.range.size();
        }
    });
    if (!best) return std::nullopt;
    const Index free_block_index = *best;
    Block *free_block = blocks_.get_mut(free_block_index);
    if (free_block->range.size() == size) { // :47-50
        free_block->free = false;
        return std::make_pair(free_block_index, free_block->range);
    }
    // :52-71 — the allocation takes the front of the free block and is linked in before it
    const size_t start = free_block->range.start;
    const std::optional<Index> prev_index = free_block->prev_index;
    const BlockRange range{start, start + size};
    const Index new_block = blocks_.insert(Block{false, range, prev_index, free_block_index});
    free_block = blocks_.get_mut(free_block_index); // insert may have moved the entries
    free_block->range.start += size;
    free_block->prev_index = new_block;
    // Deviation, on purpose: the reference stops here and leaves prev's next_index pointing at the shrunk free block
    // (:58-69), so a later deallocate(prev) would merge prev with that block ACROSS the new allocation.  The scene only
    // ever allocates (scene.rs:422-431), so the reference never reaches that state and every sequence it can produce
    // gives the same ranges here; the forward link is kept correct so that deallocate is usable at all.
    if (prev_index)
        if (Block *prev = blocks_.get_mut(*prev_index)) prev->next_index = new_block;
    return std::make_pair(new_block, range);
}

void FreeListAllocator::deallocate(Index index) { // :74-104
    if (!blocks_.has_index(index)) return;
    auto free_neighbour = [&](const std::optional<Index> &n) -> std::optional<Index> {
        if (!n) return std::nullopt;
        const Block *b = blocks_.get(*n); // the reference indexes (panics on a dead index); a dead link is skipped here
        return b && b->free ? n : std::nullopt;
    };
    const std::optional<Index> prev_free_index = free_neighbour(blocks_.get(index)->prev_index);
    const std::optional<Index> next_free_index = free_neighbour(blocks_.get(index)->next_index);
    blocks_.get_mut(index)->free = true;
    if (prev_free_index) {
        const Block prev_block = *blocks_.remove(*prev_free_index);
        if (prev_block.prev_index)
            if (Block *pp = blocks_.get_mut(*prev_block.prev_index)) pp->next_index = index;
        Block *b = blocks_.get_mut(index);
        b->prev_index = prev_block.prev_index;
        b->range.start = prev_block.range.start;
    }
    if (next_free_index) {
        const Block next_block = *blocks_.remove(*next_free_index);
        if (next_block.next_index)
            if (Block *nn = blocks_.get_mut(*next_block.next_index)) nn->prev_index = index;
        Block *b = blocks_.get_mut(index);
        b->next_index = next_block.next_index;
        b->range.end = next_block.range.end;
    }
}

std::vector<std::pair<BlockRange, bool>> FreeListAllocator::blocks_in_address_order() const {
    std::vector<std::pair<BlockRange, bool>> out;
    blocks_.for_each([&](Index, const Block &b) { out.emplace_back(b.range, b.free); });
    std::sort(out.begin(), out.end(), [](const auto &a, const auto &b) {
        if (a.first.start != b.first.start) return a.first.start < b.first.start;
        if (a.first.end != b.first.end) return a.first.end < b.first.end; // ranges can coincide after the stale-link merge
        return a.second < b.second;
    });
    return out;
}

} // namespace collections

namespace scene {

float Light::outer_radius(float cutoff) const { return std::sqrt(intensity / cutoff); } // scene.rs:273-275

namespace {
// glam Quat::mul_vec3: v * (w^2 - b.b) + b * (2 v.b) + (b x v) * (2 w)
Vec3 quat_mul_vec3(const Quat &q, const Vec3 &v) {
    const float w = q.w;
    const Vec3 b{q.x, q.y, q.z};
    const float b2 = b.x * b.x + b.y * b.y + b.z * b.z;
    const float vb = (v.x * b.x + v.y * b.y + v.z * b.z) * 2.0f;
    const Vec3 c{b.y * v.z - v.y * b.z, b.z * v.x - v.z * b.x, b.x * v.y - b.y * v.x};
    const float s = w * w - b2, w2 = w * 2.0f;
    return {v.x * s + b.x * vb + c.x * w2, v.y * s + b.y * vb + c.y * w2, v.z * s + b.z * vb + c.z * w2};
}
} // namespace

OrbitEntityData EntityData::entity_gpu_data() const { // scene.rs:75-82
    const Mat4 model = transform.compute_matrix();
    const Mat4 it = model.inverse().transpose();
    Mat4 normal = Mat4::identity(); // Mat4::from_mat3(Mat3::from_mat4(..)): upper 3x3, identity elsewhere
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) normal.c[4 * c + r] = it.c[4 * c + r];
    OrbitEntityData out;
    std::memcpy(out.model_matrix, model.c, sizeof(out.model_matrix));
    std::memcpy(out.normal_matrix, normal.c, sizeof(out.normal_matrix));
    return out;
}

std::optional<OrbitLightData> EntityData::light_gpu_data(float luminance_cutoff) const { // scene.rs:84-118
    if (!light) return std::nullopt;
    OrbitLightData d;
    std::memset(&d, 0, sizeof(d));
    d.color[0] = light->color.x, d.color[1] = light->color.y, d.color[2] = light->color.z;
    d.intensity = light->intensity;
    d.light_type = (uint32_t)light->kind;
    d.shadow_data_index = UINT32_MAX;
    switch (light->kind) {
    case LightKind::Sky:
        d.irradiance_map_index = light->irradiance_map_index;
        d.prefiltered_map_index = light->prefiltered_map_index;
        break;
    case LightKind::Directional: {
        const Vec3 f = quat_mul_vec3(transform.orientation, Vec3{0.0f, 0.0f, -1.0f});
        d.direction[0] = -f.x, d.direction[1] = -f.y, d.direction[2] = -f.z;
        d.inner_radius = light->param;
        break;
    }
    case LightKind::Point:
        d.position[0] = transform.position.x, d.position[1] = transform.position.y, d.position[2] = transform.position.z;
        d.inner_radius = light->param;
        d.outer_radius = light->outer_radius(luminance_cutoff);
        break;
    }
    return d;
}

size_t SceneData::add_entity(EntityData data) {
    const size_t index = entities.size();
    entities.push_back(std::move(data));
    return index;
}

void SceneData::update_scene(const OrbitMeshInfo *mesh_infos, size_t mesh_info_count, float luminance_cutoff,
                             size_t frame_index) {
    entity_draw_cache.clear();
    entity_data_cache.clear();
    light_data_cache.clear();
    shadow_commands.clear(); // shadow_renderer.clear_shadow_commands(), :418
    for (EntityData &entity : entities) {
        if (entity.mesh) { // :420-437
            const uint32_t mesh = *entity.mesh;
            const uint32_t instance_index = (uint32_t)entity_data_cache.size();
            size_t visibility_offset;
            if (entity.visibility_buffer_range) {
                visibility_offset = entity.visibility_buffer_range->range.start;
            } else {
                if (mesh >= mesh_info_count) throw Panic("update_scene: mesh slot out of range (index panic, scene.rs:425)");
                const size_t meshlet_count = mesh_infos[mesh].mesh_lods[0].meshlet_count;
                const auto a = meshlet_visibility_allocator_.allocate((meshlet_count + 31) / 32);
                if (!a) throw Panic("update_scene: meshlet visibility words exhausted (unwrap on None, scene.rs:427)");
                entity.visibility_buffer_range = VisibilityBufferRange{a->first, a->second};
                visibility_offset = a->second.start;
            }
            entity_data_cache.push_back(entity.entity_gpu_data());
            entity_draw_cache.push_back(GpuEntityDraw{instance_index, mesh, (uint32_t)visibility_offset});
        }
        std::optional<OrbitLightData> light_data = entity.light_gpu_data(luminance_cutoff); // :449-468
        if (entity.light && light_data) {
            entity.light->light_index = light_data_cache.size();
            if (entity.light->cast_shadows && entity.light->kind == LightKind::Directional) {
                const size_t shadow_index = shadow_commands.size();
                shadow_commands.push_back(
                    ShadowCommand{entity.has_name ? entity.name : std::string("unnamed_light"), entity.transform.orientation});
                light_data->shadow_data_index = (uint32_t)(shadow_index + MAX_SHADOW_COMMANDS * frame_index);
            }
            light_data_cache.push_back(*light_data);
        }
    }
}

std::vector<uint8_t> SceneData::entity_draw_buffer_bytes() const {
    static_assert(sizeof(GpuEntityDraw) == 12, "GpuEntityDraw is 12 B");
    std::vector<uint8_t> out(4 + entity_draw_cache.size() * sizeof(GpuEntityDraw));
    const uint32_t n = (uint32_t)entity_draw_cache.size();
    std::memcpy(out.data(), &n, 4);
    if (n) std::memcpy(out.data() + 4, entity_draw_cache.data(), entity_draw_cache.size() * sizeof(GpuEntityDraw));
    return out;
}

} // namespace scene
} // namespace orbit
