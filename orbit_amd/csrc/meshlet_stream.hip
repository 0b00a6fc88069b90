// meshlet_stream.hip — the derived meshlet streams (orbit_meshlet_stream_*; kernels.h MeshletStreamView): the static
// 32-B Meshlet records (shaders/include/types.glsl:143-152 of the reference repo) re-laid out once, at upload, into
// what every evaluation reads and what only a survivor's command needs; their alpha classes; the validation launch.
#include "kernels.h"

namespace orbit {

namespace {

// orbit_meshlet_stream_update: Meshlet records -> the derived streams (thread = meshlet; 32 B read, 38 B written).
// A wave owns 64 consecutive, 64-aligned global meshlet indices, i.e. two whole words of link bits: bits of meshlets
// outside the updated range keep their value.  Link bit of meshlet i (kernels.h): i continues i - 1's chain.  It is
// re-derived for i in [first, first + count] — the meshlet behind the range chains from the range's last one; its own
// offsets (and those of the meshlet in front of the range) come from the stream's `cmd` copy, not from the buffer,
// of which only [first, first + count) may be read.
struct StreamBuildArrays {
    uint4 *sphere;
    uint32_t *cone;
    uint16_t *mat;
    uint32_t *cmd;
    uint16_t *cnt;
    uint32_t *link, *cls0, *cls1;
    uint2 *base32;
    uint64_t s_first, s_end; // the stream's range
    const OrbitMaterialData *materials; // alpha classes come from here (nullptr: class 3 everywhere)
    uint32_t material_count;
    uint32_t *other; // set to 1 when a meshlet of class 3 is derived while `materials` is set (abi.hip: such a stream
                     // evaluates through its material indices)
};

// alpha class of a meshlet (kernels.h MeshletStreamView): its material's alpha_mode where the two bit planes can hold
// it, 3 = "look the material up" otherwise
__device__ __forceinline__ uint32_t alpha_class(const StreamBuildArrays &A, uint32_t material_index) {
    if (A.materials == nullptr || material_index >= A.material_count) return 3u;
    const uint32_t mode = A.materials[material_index].alpha_mode;
    return mode < 3u ? mode : 3u;
}

__global__ __launch_bounds__(256) void meshlet_stream_build_kernel(const OrbitMeshlet *meshlets, uint64_t first,
                                                                   uint64_t count, const StreamBuildArrays A) {
    const int lane = threadIdx.x & 63;
    const uint64_t lo = first & ~63ull, end = first + count;
    for (uint64_t base = lo + ((uint64_t)blockIdx.x * 256u + (threadIdx.x & ~63u)); base <= end;
         base += (uint64_t)gridDim.x * 256u) {
        const uint64_t m = base + (uint64_t)lane;
        const bool in = m >= first && m < end;
        uint32_t vtx = 0, data = 0, cls = 3u;
        if (in) {
            const uint4 *src = reinterpret_cast<const uint4 *>(meshlets + m);
            const uint4 a = src[0], b = src[1];
            A.sphere[m] = a;
            A.cone[m] = b.x;
            A.mat[m] = (uint16_t)(b.w & 0xFFFFu);
            A.cmd[m * 3u] = b.y;
            A.cmd[m * 3u + 1u] = b.z;
            A.cmd[m * 3u + 2u] = b.w;
            A.cnt[m] = (uint16_t)(b.w >> 16);
            if ((m & 31u) == 0u) A.base32[m >> 5] = make_uint2(b.y, b.z); // {vertex_offset, data_offset}
            vtx = b.y, data = b.z;
            cls = alpha_class(A, b.w & 0xFFFFu);
            if (cls == 3u && A.materials != nullptr) *A.other = 1u;
        } else if (m == end && m < A.s_end) { // derived earlier (or never: then the bit is as arbitrary as the rest)
            vtx = A.cmd[m * 3u], data = A.cmd[m * 3u + 1u];
        }
        const bool in_stream = m >= A.s_first && m < A.s_end;
        bool bit = false;
        if (in_stream) bit = ((A.link[m >> 5] >> (m & 31u)) & 1u) != 0; // kept unless re-derived below
        if (in_stream && m >= first && m <= end) {
            bit = false;
            if (m > A.s_first) {
                uint32_t pv, pd, p7;
                if (m > first) { // predecessor inside the updated range: from the buffer
                    const uint4 pb = reinterpret_cast<const uint4 *>(meshlets + (m - 1u))[1];
                    pv = pb.y, pd = pb.z, p7 = pb.w;
                } else { // the meshlet in front of the range: the stream's own copy
                    pv = A.cmd[(m - 1u) * 3u], pd = A.cmd[(m - 1u) * 3u + 1u], p7 = A.cmd[(m - 1u) * 3u + 2u];
                }
                bit = vtx == pv && data == pd + meshlet_data_words((p7 >> 16) & 0xFFu, p7 >> 24);
            }
        }
        // the class planes: meshlets outside the updated range keep their bits
        bool p0 = (cls & 1u) != 0u, p1 = (cls & 2u) != 0u;
        if (!in && in_stream) {
            p0 = ((A.cls0[m >> 5] >> (m & 31u)) & 1u) != 0;
            p1 = ((A.cls1[m >> 5] >> (m & 31u)) & 1u) != 0;
        }
        const uint64_t mask = __ballot(bit), m0 = __ballot(p0), m1 = __ballot(p1);
        // both words exist whenever any of their meshlets is in the stream (orbit_meshlet_stream_create)
        if (lane == 0 && base + 32u > A.s_first && base < A.s_end) {
            A.link[base >> 5] = (uint32_t)mask;
            A.cls0[base >> 5] = (uint32_t)m0;
            A.cls1[base >> 5] = (uint32_t)m1;
        }
        if (lane == 32 && base + 64u > A.s_first && base + 32u < A.s_end) {
            A.link[(base >> 5) + 1u] = (uint32_t)(mask >> 32);
            A.cls0[(base >> 5) + 1u] = (uint32_t)(m0 >> 32);
            A.cls1[(base >> 5) + 1u] = (uint32_t)(m1 >> 32);
        }
    }
}

// orbit_meshlet_stream_set_materials: the class planes of the whole stream from its material indices
__global__ __launch_bounds__(256) void meshlet_stream_classes_kernel(const StreamBuildArrays A) {
    const int lane = threadIdx.x & 63;
    for (uint64_t base = (A.s_first & ~63ull) + ((uint64_t)blockIdx.x * 256u + (threadIdx.x & ~63u)); base < A.s_end;
         base += (uint64_t)gridDim.x * 256u) {
        const uint64_t m = base + (uint64_t)lane;
        const bool in = m >= A.s_first && m < A.s_end;
        const uint32_t cls = in ? alpha_class(A, (uint32_t)A.mat[m]) : 3u;
        if (in && cls == 3u && A.materials != nullptr) *A.other = 1u;
        const uint64_t m0 = __ballot((cls & 1u) != 0u), m1 = __ballot((cls & 2u) != 0u);
        if (lane == 0 && base + 32u > A.s_first) {
            A.cls0[base >> 5] = (uint32_t)m0;
            A.cls1[base >> 5] = (uint32_t)m1;
        }
        if (lane == 32 && base + 32u < A.s_end) {
            A.cls0[(base >> 5) + 1u] = (uint32_t)(m0 >> 32);
            A.cls1[(base >> 5) + 1u] = (uint32_t)(m1 >> 32);
        }
    }
}

} // namespace

static StreamBuildArrays stream_build_arrays(const MeshletStreamView &ms, const OrbitMaterialData *materials,
                                             uint32_t material_count, uint32_t *other = nullptr) {
    return StreamBuildArrays{const_cast<uint4 *>(ms.sphere),   const_cast<uint32_t *>(ms.cone),
                             const_cast<uint16_t *>(ms.mat),   const_cast<uint32_t *>(ms.cmd),
                             const_cast<uint16_t *>(ms.cnt),   const_cast<uint32_t *>(ms.link),
                             const_cast<uint32_t *>(ms.cls0),  const_cast<uint32_t *>(ms.cls1),
                             const_cast<uint2 *>(ms.base32),
                             (uint64_t)ms.first,               (uint64_t)ms.first + ms.count,
                             materials,                        material_count,
                             other};
}

hipError_t launch_meshlet_stream_build(const OrbitMeshlet *meshlets, uint64_t first, uint64_t count,
                                       const MeshletStreamView &ms, const OrbitMaterialData *materials,
                                       uint32_t material_count, uint32_t *other, hipStream_t s) {
    if (count == 0) return hipSuccess;
    const uint64_t need = (count + 64u + 255u) / 256u + 1u;
    const uint32_t blocks = (uint32_t)(need < 8192u ? need : 8192u);
    hipLaunchKernelGGL(meshlet_stream_build_kernel, dim3(blocks), dim3(256), 0, s, meshlets, first, count,
                       stream_build_arrays(ms, materials, material_count, other));
    return hipGetLastError();
}

// orbit_meshlet_stream_validate: thread = meshlet of the stream's range; the Meshlet as the arrays hold it against the
// Meshlet in the buffer, the alpha class against the material's alpha_mode.
__global__ __launch_bounds__(256) void meshlet_stream_validate_kernel(const OrbitMeshlet *meshlets, const StreamBuildArrays A,
                                                                      int32_t *status) {
    for (uint64_t m = A.s_first + (uint64_t)blockIdx.x * 256u + threadIdx.x; m < A.s_end; m += (uint64_t)gridDim.x * 256u) {
        const uint4 *src = reinterpret_cast<const uint4 *>(meshlets + m);
        const uint4 a = src[0], b = src[1], sa = A.sphere[m];
        bool same = a.x == sa.x && a.y == sa.y && a.z == sa.z && a.w == sa.w && b.x == A.cone[m] &&
                    (b.w & 0xFFFFu) == (uint32_t)A.mat[m] && b.y == A.cmd[m * 3u] && b.z == A.cmd[m * 3u + 1u] &&
                    b.w == A.cmd[m * 3u + 2u] && (b.w >> 16) == (uint32_t)A.cnt[m];
        if (A.materials) {
            const uint32_t cls = ((A.cls0[m >> 5] >> (m & 31u)) & 1u) | (((A.cls1[m >> 5] >> (m & 31u)) & 1u) << 1);
            same = same && cls == alpha_class(A, b.w & 0xFFFFu);
        }
        if ((m & 31u) == 0u) {
            const uint2 b32 = A.base32[m >> 5];
            same = same && b32.x == b.y && b32.y == b.z;
        }
        if (m > A.s_first) { // the link bit against the two meshlets it relates
            const uint4 pb = reinterpret_cast<const uint4 *>(meshlets + (m - 1u))[1];
            const bool link = b.y == pb.y && b.z == pb.z + meshlet_data_words((pb.w >> 16) & 0xFFu, pb.w >> 24);
            same = same && link == (((A.link[m >> 5] >> (m & 31u)) & 1u) != 0u);
        }
        if (!same) latch_status(status, ORBIT_E_STALE);
    }
}

hipError_t launch_meshlet_stream_validate(const OrbitMeshlet *meshlets, const MeshletStreamView &ms,
                                          const OrbitMaterialData *materials, uint32_t material_count, int32_t *status,
                                          hipStream_t s) {
    if (ms.count == 0) return hipSuccess;
    const uint64_t need = ((uint64_t)ms.count + 255u) / 256u;
    const uint32_t blocks = (uint32_t)(need < 8192u ? need : 8192u);
    hipLaunchKernelGGL(meshlet_stream_validate_kernel, dim3(blocks), dim3(256), 0, s, meshlets,
                       stream_build_arrays(ms, materials, material_count), status);
    return hipGetLastError();
}

hipError_t launch_meshlet_stream_classes(const MeshletStreamView &ms, const OrbitMaterialData *materials,
                                         uint32_t material_count, uint32_t *other, hipStream_t s) {
    const uint64_t need = ((uint64_t)ms.count + 64u + 255u) / 256u + 1u;
    const uint32_t blocks = (uint32_t)(need < 8192u ? need : 8192u);
    hipLaunchKernelGGL(meshlet_stream_classes_kernel, dim3(blocks), dim3(256), 0, s,
                       stream_build_arrays(ms, materials, material_count, other));
    return hipGetLastError();
}

namespace {

// orbit_meshlet_stream_update_meshes: thread = mesh
__global__ __launch_bounds__(256) void mesh_side_build_kernel(const OrbitMeshInfo *mesh_infos, uint32_t first, uint32_t count,
                                                             MeshSide *table) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const OrbitMeshInfo &m = mesh_infos[first + i];
    MeshSide s;
    s.sphere[0] = m.bounding_sphere[0], s.sphere[1] = m.bounding_sphere[1], s.sphere[2] = m.bounding_sphere[2], s.sphere[3] = m.bounding_sphere[3];
    s.lod_count = m.lod_count;
    s.lod0_offset = m.mesh_lods[0].meshlet_offset;
    s.lod0_count = m.mesh_lods[0].meshlet_count;
    s.valid = 1u;
    table[first + i] = s;
}

__global__ __launch_bounds__(256) void mesh_side_validate_kernel(const OrbitMeshInfo *mesh_infos, uint32_t first, uint32_t count,
                                                                const MeshSide *table, int32_t *status) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const OrbitMeshInfo &m = mesh_infos[first + i];
    const MeshSide s = table[first + i];
    if (s.valid == 0u) return; // never derived: the entity stage reads the MeshInfo itself
    bool same = s.lod_count == m.lod_count && s.lod0_offset == m.mesh_lods[0].meshlet_offset && s.lod0_count == m.mesh_lods[0].meshlet_count;
    for (int k = 0; k < 4; k++) same = same && __float_as_uint(s.sphere[k]) == __float_as_uint(m.bounding_sphere[k]);
    if (!same) latch_status(status, ORBIT_E_STALE);
}

} // namespace

hipError_t launch_mesh_side_build(const OrbitMeshInfo *mesh_infos, uint32_t first, uint32_t count, MeshSide *table, hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(mesh_side_build_kernel, dim3((count + 255u) / 256u), dim3(256), 0, s, mesh_infos, first, count, table);
    return hipGetLastError();
}

hipError_t launch_mesh_side_validate(const OrbitMeshInfo *mesh_infos, uint32_t first, uint32_t count, const MeshSide *table,
                                     int32_t *status, hipStream_t s) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(mesh_side_validate_kernel, dim3((count + 255u) / 256u), dim3(256), 0, s, mesh_infos, first, count, table, status);
    return hipGetLastError();
}

} // namespace orbit
