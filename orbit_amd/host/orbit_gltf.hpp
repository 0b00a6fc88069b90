// orbit_gltf.hpp — host-side mirror of the asset ingestion in front of the cull path:
//
//   load_gltf            src/gltf_loader.rs:511-676   .glb / .gltf (+ external .bin) -> materials, meshes, entities
//   load_gltf_mesh       src/gltf_loader.rs:423-509   one MeshData per glTF mesh, one submesh per primitive
//   GpuAssets::add_mesh  src/assets/mod.rs:325-476    the LOD chain: up to MAX_MESH_LODS levels, target index count
//                                                     x0.8 per submesh per level, every level meshletized, the
//                                                     (meshlet_offset, meshlet_count) pairs into MeshInfo::mesh_lods
//   Transform::from_mat4 src/scene.rs:41-48           node matrix -> scale / rotation / translation (glam restated)
//
// Third-party pieces of the reference that are NOT in its tree (crates gltf 1.x, meshopt 0.2.0) are replaced by own
// code, declared as such:
//   * the glTF container and JSON are read by a small parser in this file's .cpp (only what the loader touches:
//     buffers, bufferViews, accessors, meshes/primitives POSITION + indices, materials' alpha mode and factors,
//     nodes, scenes);
//   * meshopt::simplify (quadric edge collapse) is replaced by `simplify_clustered` — vertex clustering on a uniform
//     grid (Rossignac-Borrel), grid resolution searched so that the result meets the target index count; like
//     meshopt_simplify it only re-indexes existing vertices.  LOD contents therefore differ from the renderer's; the
//     chain's structure (levels, targets, finish rule, offsets) follows the reference statement by statement.
//   * optimize_mesh (vertex cache / fetch optimisation, gltf_loader.rs:487-493) does not change what the cull path
//     sees except through the meshlet partition, which is already this project's own (orbit_assets.hpp): skipped.
// "parity unpinned": nothing in the reference pins these results; tests/test_gltf_cpu.py checks them against the
// geometry (every LOD's meshlets index valid vertices, bounds contain them, counts fall along the chain) and
// tests/test_gltf_gpu.py runs the product on the result against the oracle.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "orbit_assets.hpp"
#include "orbit_host.hpp"

namespace orbit {
namespace assets {

constexpr size_t MAX_MESH_LODS = ORBIT_MAX_MESH_LODS; // assets/mod.rs:16

struct SubmeshData { // mesh.rs:160-171
    size_t vertex_offset = 0, vertex_count = 0, index_offset = 0, index_count = 0;
    uint16_t material = 0;
};

// mesh.rs:142-215 with positions only (the cull path reads nothing else of a vertex)
struct MeshData {
    std::vector<float> positions; // xyz per vertex
    std::vector<uint32_t> indices;
    std::vector<SubmeshData> submeshes;
    float aabb_min[3] = {0, 0, 0}, aabb_max[3] = {0, 0, 0};
    float bounding_sphere[4] = {0, 0, 0, 0};
    size_t add_submesh(const float *positions, size_t vertex_count, const uint32_t *indices, size_t index_count,
                       uint16_t material);                                  // mesh.rs:173-190
    void compute_bounds();                                                  // mesh.rs:192-215
};

// Replacement of meshopt::simplify (see the header comment): appends at most target_index_count indices (whole
// triangles over existing vertices) to `out`; returns the relative error proxy (grid cell size / mesh extent).
float simplify_clustered(const float *positions, size_t vertex_count, const uint32_t *indices, size_t index_count,
                         size_t target_index_count, std::vector<uint32_t> &out);

// The buffers GpuAssets owns, as the cull path sees them (assets/mod.rs:197-323): arenas become growing vectors.
struct GpuAssets {
    std::vector<float> vertex_positions;    // vertex_buffer, xyz only
    std::vector<uint32_t> meshlet_data;     // meshlet_data_buffer
    std::vector<OrbitMeshlet> meshlets;     // meshlet_buffer
    std::vector<OrbitMeshInfo> mesh_infos;  // mesh_info_buffer
    std::vector<OrbitMaterialData> materials;
    uint32_t add_material(const OrbitMaterialData &m); // assets/mod.rs:478-520 (slot = index)
    uint32_t add_mesh(const MeshData &mesh);           // assets/mod.rs:325-476; returns the mesh slot
};

} // namespace assets

namespace gltf_loader {

struct LoadedEntity { // scene.add_entity(EntityData { transform, mesh, .. }), gltf_loader.rs:656-662
    std::string name;
    passes::Transform transform;
    int32_t mesh = -1; // MeshHandle slot or -1
};

struct LoadedScene {
    assets::GpuAssets assets;
    std::vector<LoadedEntity> entities; // in add_gltf_node order (depth first, parents before children)
};

// Transform::from_mat4 (scene.rs:41-48) = glam Mat4::to_scale_rotation_translation
passes::Transform transform_from_mat4(const Mat4 &m);

// load_gltf (gltf_loader.rs:511-676).  Returns false with `error` set on malformed input (the reference unwraps).
bool load_gltf(const std::string &path, LoadedScene &out, std::string &error);

} // namespace gltf_loader
} // namespace orbit
