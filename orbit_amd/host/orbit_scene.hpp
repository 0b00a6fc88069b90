// orbit_scene.hpp — host-side mirror of the producers of the cull path's input buffers.
//
// SURVEY.md §8 row a10 and the scene half of §8f-3: the reference builds EntityDrawBuffer, EntityData[] and
// LightData[] in SceneData::update_scene (src/scene.rs:404-492) and hands every entity its meshlet-visibility words
// from a best-fit free-list allocator over a generational arena (src/collections/freelist_alloc.rs,
// src/collections/arena.rs).  Same names, same results (word offsets, slot reuse order, byte layouts), restated in
// C++ because this image has no Rust toolchain.  Host code only: nothing here touches the device.
#pragma once
#include <cstdint>
#include <optional>
#include <string>
#include <utility>
#include <vector>

#include "orbit_host.hpp"

namespace orbit {
namespace collections {

// arena.rs:70-96: generation starts at 1 and skips 0 on wrap; slot is the entry's position.
struct Index {
    uint32_t generation = 1;
    uint32_t slot = 0;
    bool operator==(const Index &o) const { return generation == o.generation && slot == o.slot; }
    bool operator!=(const Index &o) const { return !(*this == o); }
};

// arena.rs:98-330.  Removed slots form a LIFO free list; insert reuses the most recently freed slot and bumps its
// generation; iteration visits occupied slots in slot order.
template <typename T>
class Arena {
  public:
    size_t len() const { return len_; }
    bool is_empty() const { return len_ == 0; }

    Index insert(T val) {
        len_++;
        if (first_free_) {
            const uint32_t slot = *first_free_;
            Entry &e = entries_[slot];
            first_free_ = e.next_free;
            e.generation = next_generation(e.generation);
            e.val = std::move(val);
            e.next_free.reset();
            return {e.generation, slot};
        }
        entries_.push_back(Entry{1u, std::move(val), std::nullopt});
        return {1u, (uint32_t)entries_.size() - 1u};
    }

    const T *get(Index i) const {
        if (i.slot >= entries_.size()) return nullptr;
        const Entry &e = entries_[i.slot];
        return e.val && e.generation == i.generation ? &*e.val : nullptr;
    }
    T *get_mut(Index i) { return const_cast<T *>(static_cast<const Arena *>(this)->get(i)); }
    bool has_index(Index i) const { return get(i) != nullptr; }

    std::optional<T> remove(Index i) {
        if (i.slot >= entries_.size()) return std::nullopt;
        Entry &e = entries_[i.slot];
        if (e.generation != i.generation || !e.val) return std::nullopt;
        std::optional<T> out = std::move(e.val);
        e.val.reset();
        e.next_free = first_free_;
        first_free_ = i.slot;
        len_--;
        return out;
    }

    void clear() {
        entries_.clear();
        first_free_.reset();
        len_ = 0;
    }

    // occupied entries in slot order (arena.rs Iter)
    template <typename F>
    void for_each(F &&f) const {
        for (uint32_t s = 0; s < entries_.size(); s++)
            if (entries_[s].val) f(Index{entries_[s].generation, s}, *entries_[s].val);
    }

  private:
    struct Entry {
        uint32_t generation;
        std::optional<T> val;
        std::optional<uint32_t> next_free;
    };
    static uint32_t next_generation(uint32_t g) { return g == UINT32_MAX ? 1u : g + 1u; } // arena.rs:11-13
    std::vector<Entry> entries_;
    std::optional<uint32_t> first_free_;
    size_t len_ = 0;
};

// freelist_alloc.rs:3-13
struct BlockRange {
    size_t start = 0, end = 0;
    size_t size() const { return end - start; }
};

// freelist_alloc.rs:22-121: best fit (smallest free block that holds `size`; among equals the first in arena slot
// order), split at the front of the chosen block, coalescing of both neighbours on deallocate.  One deviation: allocate
// also repairs the predecessor's forward link, which the reference leaves stale (see orbit_scene.cpp) — identical
// results on every allocate-only sequence, which is all the reference's scene ever issues.
class FreeListAllocator {
  public:
    explicit FreeListAllocator(size_t size);
    std::optional<std::pair<Index, BlockRange>> allocate(size_t size);
    void deallocate(Index index);
    size_t block_count() const { return blocks_.len(); }
    // (range, free) of every block in address order — for tests and debugging
    std::vector<std::pair<BlockRange, bool>> blocks_in_address_order() const;

  private:
    struct Block {
        bool free;
        BlockRange range;
        std::optional<Index> prev_index, next_index;
    };
    Arena<Block> blocks_;
};

} // namespace collections

namespace scene {

constexpr size_t MAX_INSTANCE_COUNT = 100000; // scene.rs:303
constexpr size_t MAX_LIGHT_COUNT = 2000;      // scene.rs:304
// scene.rs:350-352: "avg. 256 meshlet per instance, 1 bit per meshlet, 4 byte per chunk"
constexpr size_t MESHLET_VISIBILITY_BUFFER_CHUNK_COUNT = MAX_INSTANCE_COUNT * 256 / 32;
constexpr size_t MAX_SHADOW_COMMANDS = 256;   // shadow_renderer.rs:204

enum class LightKind : uint32_t { Sky = 0, Directional = 1, Point = 2 }; // scene.rs:136-141

// scene.rs:152-238.  `param` is angular_size (Directional) or inner_radius (Point); the two map indices are the
// Sky light's bindless descriptors.
struct Light {
    Vec3 color{1.0f, 1.0f, 1.0f};
    float intensity = 1.0f;
    LightKind kind = LightKind::Point;
    float param = 0.6f;
    uint32_t irradiance_map_index = 0, prefiltered_map_index = 0;
    bool cast_shadows = false;
    std::optional<size_t> light_index; // _light_index, scene.rs:225
    float outer_radius(float cutoff) const; // scene.rs:273-275
};

struct VisibilityBufferRange { // scene.rs:60-63
    collections::Index alloc_index;
    collections::BlockRange range;
};

struct EntityData { // scene.rs:66-72
    std::string name;
    bool has_name = false;
    passes::Transform transform;
    std::optional<uint32_t> mesh; // MeshHandle::slot()
    std::optional<Light> light;
    std::optional<VisibilityBufferRange> visibility_buffer_range;
    OrbitEntityData entity_gpu_data() const;                                  // scene.rs:75-82
    std::optional<OrbitLightData> light_gpu_data(float luminance_cutoff) const; // scene.rs:84-118
};

struct GpuEntityDraw { // scene.rs:127-133, types.glsl:112-116
    uint32_t instance_index, mesh_index, visibility_offset;
};

struct ShadowCommand { // shadow_renderer.rs:180-190, Directional only
    std::string name;
    Quat orientation;
};

// scene.rs:358-492 without the Vulkan buffers: the caches ARE what queue_write_buffer uploads.
class SceneData {
  public:
    SceneData() : meshlet_visibility_allocator_(MESHLET_VISIBILITY_BUFFER_CHUNK_COUNT) {}
    explicit SceneData(size_t visibility_chunk_count) : meshlet_visibility_allocator_(visibility_chunk_count) {}

    std::vector<EntityData> entities;
    std::vector<GpuEntityDraw> entity_draw_cache;
    std::vector<OrbitEntityData> entity_data_cache;
    std::vector<OrbitLightData> light_data_cache;
    std::vector<ShadowCommand> shadow_commands; // what update_scene hands to ShadowRenderer::add_shadow

    size_t add_entity(EntityData data); // scene.rs:398-402
    // scene.rs:404-492.  `mesh_infos` = assets.shared_stuff.mesh_infos; an entity whose visibility words cannot be
    // allocated panics like the reference's unwrap (:427).
    void update_scene(const OrbitMeshInfo *mesh_infos, size_t mesh_info_count, float luminance_cutoff,
                      size_t frame_index);
    // the bytes written at offset 0 of entity_draw_buffer (:470-481): u32 count, then the draws
    std::vector<uint8_t> entity_draw_buffer_bytes() const;
    collections::FreeListAllocator &meshlet_visibility_allocator() { return meshlet_visibility_allocator_; }

  private:
    collections::FreeListAllocator meshlet_visibility_allocator_;
};

} // namespace scene
} // namespace orbit
