// orbit_assets.cpp — see orbit_assets.hpp.
//
// DERIVED WORK NOTICE.  compute_bounding_sphere() and compute_meshlet_bounds() below follow meshoptimizer's
// computeBoundingSphere / meshopt_computeClusterBounds (clusterizer.cpp; the crate the reference pins is
// meshopt 0.2.0, Cargo.toml:37) closely enough — same algorithm, thresholds and several identifiers — to count
// as derived from it.  meshoptimizer is distributed under the MIT License; its notice is reproduced here and in
// THIRD_PARTY.md at the repo root:
//
//   MIT License
//   Copyright (c) 2016-2024 Arseny Kapoulkine
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated
//   documentation files (the "Software"), to deal in the Software without restriction, including without
//   limitation the rights to use, copy, modify, merge, publish, distribute, sublicense, and/or sell copies of the
//   Software, and to permit persons to whom the Software is furnished to do so, subject to the following
//   conditions: The above copyright notice and this permission notice shall be included in all copies or
//   substantial portions of the Software.
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED
//   TO THE WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT. IN NO EVENT SHALL
//   THE AUTHORS OR COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF
//   CONTRACT, TORT OR OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER
//   DEALINGS IN THE SOFTWARE.
//
// The greedy meshlet partition in this file is this project's own and is not derived from meshoptimizer.
#include "orbit_assets.hpp"

#include <cmath>
#include <cstring>

namespace orbit {
namespace assets {

namespace {

// meshoptimizer computeBoundingSphere: Ritter's sphere seeded with the most distant pair among the six axis extremes.
void compute_bounding_sphere(float result[4], const float (*points)[3], size_t count) {
    size_t pmin[3] = {0, 0, 0}, pmax[3] = {0, 0, 0};
    for (size_t i = 0; i < count; i++)
        for (int axis = 0; axis < 3; axis++) {
            pmin[axis] = points[i][axis] < points[pmin[axis]][axis] ? i : pmin[axis];
            pmax[axis] = points[i][axis] > points[pmax[axis]][axis] ? i : pmax[axis];
        }
    float paxisd2 = 0;
    int paxis = 0;
    for (int axis = 0; axis < 3; axis++) {
        const float *p1 = points[pmin[axis]], *p2 = points[pmax[axis]];
        const float d2 = (p2[0] - p1[0]) * (p2[0] - p1[0]) + (p2[1] - p1[1]) * (p2[1] - p1[1]) +
                         (p2[2] - p1[2]) * (p2[2] - p1[2]);
        if (d2 > paxisd2) {
            paxisd2 = d2;
            paxis = axis;
        }
    }
    const float *p1 = points[pmin[paxis]], *p2 = points[pmax[paxis]];
    float center[3] = {(p1[0] + p2[0]) / 2, (p1[1] + p2[1]) / 2, (p1[2] + p2[2]) / 2};
    float radius = std::sqrt(paxisd2) / 2;
    for (size_t i = 0; i < count; i++) {
        const float *p = points[i];
        const float d2 = (p[0] - center[0]) * (p[0] - center[0]) + (p[1] - center[1]) * (p[1] - center[1]) +
                         (p[2] - center[2]) * (p[2] - center[2]);
        if (d2 > radius * radius) {
            const float d = std::sqrt(d2);
            const float k = 0.5f + (radius / d) / 2;
            center[0] = center[0] * k + p[0] * (1 - k);
            center[1] = center[1] * k + p[1] * (1 - k);
            center[2] = center[2] * k + p[2] * (1 - k);
            radius = (radius + d) / 2;
        }
    }
    result[0] = center[0], result[1] = center[1], result[2] = center[2], result[3] = radius;
}

int quantize_snorm8(float v) { // meshopt_quantizeSnorm(v, 8)
    const float scale = 127.0f;
    const float round = v >= 0 ? 0.5f : -0.5f;
    v = v >= -1 ? v : -1;
    v = v <= +1 ? v : +1;
    return (int)(v * scale + round);
}

} // namespace

std::vector<RawMeshlet> build_meshlets(const uint32_t *indices, size_t index_count, size_t vertex_count,
                                       size_t max_vertices, size_t max_triangles) {
    if (max_vertices < 3 || max_vertices > 255 || max_triangles < 1) throw Panic("build_meshlets: limits out of range");
    if (index_count % 3 != 0) throw Panic("build_meshlets: index count is not a multiple of 3");
    std::vector<RawMeshlet> out;
    std::vector<uint8_t> local(vertex_count, 0xFF); // vertex -> its slot in the open meshlet, 0xFF = not in it
    RawMeshlet cur;
    auto close = [&] {
        for (uint32_t v : cur.vertices) local[v] = 0xFF;
        out.push_back(std::move(cur));
        cur = RawMeshlet();
    };
    for (size_t t = 0; t < index_count; t += 3) {
        const uint32_t v[3] = {indices[t], indices[t + 1], indices[t + 2]};
        size_t fresh = 0;
        for (int k = 0; k < 3; k++) {
            if (v[k] >= vertex_count) throw Panic("build_meshlets: index out of range");
            bool seen = local[v[k]] != 0xFF;
            for (int j = 0; j < k; j++) seen = seen || v[j] == v[k];
            fresh += seen ? 0 : 1;
        }
        if (cur.vertices.size() + fresh > max_vertices || cur.triangles.size() / 3 + 1 > max_triangles) close();
        for (int k = 0; k < 3; k++) {
            if (local[v[k]] == 0xFF) {
                local[v[k]] = (uint8_t)cur.vertices.size();
                cur.vertices.push_back(v[k]);
            }
            cur.triangles.push_back(local[v[k]]);
        }
    }
    if (!cur.triangles.empty()) close();
    return out;
}

MeshletBounds compute_meshlet_bounds(const RawMeshlet &meshlet, const float *positions) {
    // meshopt_computeClusterBounds
    const size_t tri_count = meshlet.triangles.size() / 3;
    std::vector<float> normals_store(3 * (tri_count ? tri_count : 1)), corners_store(9 * (tri_count ? tri_count : 1));
    float(*normals)[3] = reinterpret_cast<float(*)[3]>(normals_store.data());
    float(*corners)[3] = reinterpret_cast<float(*)[3]>(corners_store.data());
    size_t triangles = 0;
    for (size_t i = 0; i < tri_count; i++) {
        const float *p0 = positions + 3 * (size_t)meshlet.vertices[meshlet.triangles[3 * i + 0]];
        const float *p1 = positions + 3 * (size_t)meshlet.vertices[meshlet.triangles[3 * i + 1]];
        const float *p2 = positions + 3 * (size_t)meshlet.vertices[meshlet.triangles[3 * i + 2]];
        const float p10[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
        const float p20[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
        const float nx = p10[1] * p20[2] - p10[2] * p20[1];
        const float ny = p10[2] * p20[0] - p10[0] * p20[2];
        const float nz = p10[0] * p20[1] - p10[1] * p20[0];
        const float area = std::sqrt(nx * nx + ny * ny + nz * nz);
        if (area == 0.f) continue; // no valid normal for a degenerate triangle
        normals[triangles][0] = nx / area, normals[triangles][1] = ny / area, normals[triangles][2] = nz / area;
        std::memcpy(corners[3 * triangles + 0], p0, 12);
        std::memcpy(corners[3 * triangles + 1], p1, 12);
        std::memcpy(corners[3 * triangles + 2], p2, 12);
        triangles++;
    }
    MeshletBounds bounds;
    if (triangles == 0) return bounds; // degenerate cluster: no valid triangles
    float psphere[4];
    compute_bounding_sphere(psphere, corners, triangles * 3);
    const float center[3] = {psphere[0], psphere[1], psphere[2]};
    // the normals as points: the centre of their bounding sphere is the cone axis
    float nsphere[4];
    compute_bounding_sphere(nsphere, normals, triangles);
    float axis[3] = {nsphere[0], nsphere[1], nsphere[2]};
    const float axislength = std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
    const float invaxislength = axislength == 0.f ? 0.f : 1.f / axislength;
    axis[0] *= invaxislength, axis[1] *= invaxislength, axis[2] *= invaxislength;
    float mindp = 1.f; // cos(half angle) of the tightest cone around the normals
    for (size_t i = 0; i < triangles; i++) {
        const float dp = normals[i][0] * axis[0] + normals[i][1] * axis[1] + normals[i][2] * axis[2];
        mindp = dp < mindp ? dp : mindp;
    }
    bounds.center[0] = center[0], bounds.center[1] = center[1], bounds.center[2] = center[2];
    bounds.radius = psphere[3];
    if (mindp <= 0.1f) { // normal cone wider than ~168 degrees: no useful cone, trivial accept
        bounds.cone_cutoff = 1;
        bounds.cone_cutoff_s8 = 127;
        return bounds;
    }
    float maxt = 0;
    for (size_t i = 0; i < triangles; i++) { // the point on centre - t * axis behind every triangle's plane
        const float *corner = corners[3 * i];
        const float cx = center[0] - corner[0], cy = center[1] - corner[1], cz = center[2] - corner[2];
        const float dc = cx * normals[i][0] + cy * normals[i][1] + cz * normals[i][2];
        const float dn = axis[0] * normals[i][0] + axis[1] * normals[i][1] + axis[2] * normals[i][2];
        const float t = dc / dn;
        maxt = t > maxt ? t : maxt;
    }
    for (int k = 0; k < 3; k++) {
        bounds.cone_apex[k] = center[k] - axis[k] * maxt;
        bounds.cone_axis[k] = axis[k];
    }
    bounds.cone_cutoff = std::sqrt(1 - mindp * mindp);
    float err = 0;
    for (int k = 0; k < 3; k++) {
        bounds.cone_axis_s8[k] = (int8_t)quantize_snorm8(axis[k]);
        err += std::fabs(bounds.cone_axis_s8[k] / 127.f - axis[k]);
    }
    const int cutoff_s8 = (int)(127 * (bounds.cone_cutoff + err) + 1); // rounded up: the 8-bit test stays conservative
    bounds.cone_cutoff_s8 = cutoff_s8 > 127 ? 127 : (int8_t)cutoff_s8;
    return bounds;
}

void compute_meshlets(const float *positions, size_t vertex_count, const uint32_t *indices, size_t index_count,
                      uint16_t material, uint32_t vertex_offset, std::vector<uint32_t> &meshlet_data,
                      std::vector<OrbitMeshlet> &meshlets) {
    const std::vector<RawMeshlet> raw = build_meshlets(indices, index_count, vertex_count);
    for (const RawMeshlet &m : raw) {
        const size_t data_offset = meshlet_data.size(); // mesh.rs:311-312
        meshlet_data.insert(meshlet_data.end(), m.vertices.begin(), m.vertices.end());
        const size_t triangle_offset = meshlet_data.size() * 4; // :314, in bytes
        meshlet_data.resize(meshlet_data.size() + (m.triangles.size() + 3) / 4, 0u);
        std::memcpy(reinterpret_cast<uint8_t *>(meshlet_data.data()) + triangle_offset, m.triangles.data(),
                    m.triangles.size());
        const MeshletBounds b = compute_meshlet_bounds(m, positions);
        OrbitMeshlet out;
        std::memset(&out, 0, sizeof(out));
        out.bounding_sphere[0] = b.center[0], out.bounding_sphere[1] = b.center[1], out.bounding_sphere[2] = b.center[2];
        out.bounding_sphere[3] = b.radius;
        out.cone_axis[0] = b.cone_axis_s8[0], out.cone_axis[1] = b.cone_axis_s8[1], out.cone_axis[2] = b.cone_axis_s8[2];
        out.cone_cutoff = b.cone_cutoff_s8;
        out.vertex_offset = vertex_offset;
        if (data_offset > 0xFFFFFFFFull) throw Panic("compute_meshlets: data_offset does not fit u32 (try_into().unwrap(), mesh.rs:332)");
        out.data_offset = (uint32_t)data_offset;
        out.material_index = material;
        out.vertex_count = (uint8_t)m.vertices.size();
        out.triangle_count = (uint8_t)(m.triangles.size() / 3);
        meshlets.push_back(out);
    }
}

void compute_mesh_bounds(const float *positions, size_t vertex_count, float aabb_min[3], float aabb_max[3],
                         float bounding_sphere[4]) {
    for (int k = 0; k < 3; k++) aabb_min[k] = INFINITY, aabb_max[k] = -INFINITY;
    for (size_t i = 0; i < vertex_count; i++)
        for (int k = 0; k < 3; k++) {
            aabb_min[k] = std::fmin(aabb_min[k], positions[3 * i + k]);
            aabb_max[k] = std::fmax(aabb_max[k], positions[3 * i + k]);
        }
    const float c[3] = {(aabb_min[0] + aabb_max[0]) * 0.5f, (aabb_min[1] + aabb_max[1]) * 0.5f,
                        (aabb_min[2] + aabb_max[2]) * 0.5f};
    float r2 = 0.0f;
    for (size_t i = 0; i < vertex_count; i++) {
        const float dx = c[0] - positions[3 * i], dy = c[1] - positions[3 * i + 1], dz = c[2] - positions[3 * i + 2];
        r2 = std::fmax(r2, dx * dx + dy * dy + dz * dz); // distance_squared, gltf_loader.rs:503-504
    }
    bounding_sphere[0] = c[0], bounding_sphere[1] = c[1], bounding_sphere[2] = c[2], bounding_sphere[3] = std::sqrt(r2);
}

} // namespace assets
} // namespace orbit
