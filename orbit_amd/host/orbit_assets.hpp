// orbit_assets.hpp — host-side mirror of the producers of Meshlet[] and MeshInfo.
//
// The reference turns a triangle mesh into the 32-byte Meshlet records the cull path streams in
// assets::mesh::compute_meshlets (src/assets/mesh.rs:292-338) with two calls into meshoptimizer (crate meshopt 0.2.0,
// Cargo.toml:40 — third party, not in the reference tree):
//   * meshopt::compute_meshlet_bounds — restated here from meshoptimizer's published algorithm
//     (meshopt_computeClusterBounds: Ritter bounding sphere of the corners, bounding sphere of the unit normals as
//     the cone axis, cutoff sqrt(1 - mindp^2), snorm8 quantisation with the cutoff rounded up by the axis error);
//   * meshopt::build_meshlets — a heuristic partition; ANY partition within the vertex / triangle limits is a valid
//     input to the cull path, so this file carries its own (greedy scan with vertex reuse), not a copy of
//     meshoptimizer's.  Meshlet contents therefore differ from the renderer's for the same mesh; the layouts, limits
//     and the bounds of each meshlet follow the reference exactly.
// The mesh-level bounds are gltf_loader.rs:480-506 (AABB, sphere = AABB centre + farthest vertex).
// "parity unpinned": nothing in the reference pins these results; tests/test_assets_cpu.py checks them against the
// geometry itself (containment, cone soundness) instead.
#pragma once
#include <cstdint>
#include <vector>

#include "orbit_host.hpp"

namespace orbit {
namespace assets {

constexpr size_t MAX_MESHLET_VERTICES = 64;  // mesh.rs:8
constexpr size_t MAX_MESHLET_TRIANGLES = 64; // mesh.rs:9

// one meshlet before packing: indices into the mesh's vertices + local 8-bit triangle corners
struct RawMeshlet {
    std::vector<uint32_t> vertices;
    std::vector<uint8_t> triangles; // 3 per triangle, indices into `vertices`
};

// meshopt::Bounds
struct MeshletBounds {
    float center[3] = {0, 0, 0};
    float radius = 0;
    float cone_apex[3] = {0, 0, 0};
    float cone_axis[3] = {0, 0, 0};
    float cone_cutoff = 0;
    int8_t cone_axis_s8[3] = {0, 0, 0};
    int8_t cone_cutoff_s8 = 0;
};

// positions: xyz per vertex; indices: 3 per triangle
std::vector<RawMeshlet> build_meshlets(const uint32_t *indices, size_t index_count, size_t vertex_count,
                                       size_t max_vertices = MAX_MESHLET_VERTICES,
                                       size_t max_triangles = MAX_MESHLET_TRIANGLES);
MeshletBounds compute_meshlet_bounds(const RawMeshlet &meshlet, const float *positions);

// assets::mesh::compute_meshlets (mesh.rs:292-338): appends to meshlet_data (vertex indices, then the u8 triangle
// corners padded to a whole u32) and to meshlets.
void compute_meshlets(const float *positions, size_t vertex_count, const uint32_t *indices, size_t index_count,
                      uint16_t material, uint32_t vertex_offset, std::vector<uint32_t> &meshlet_data,
                      std::vector<OrbitMeshlet> &meshlets);

// gltf_loader.rs:480-506: aabb over the positions, sphere centred on the aabb centre through the farthest vertex
void compute_mesh_bounds(const float *positions, size_t vertex_count, float aabb_min[3], float aabb_max[3],
                         float bounding_sphere[4]);

} // namespace assets
} // namespace orbit
