"""ctypes binding of the asset side of ``liborbit_host.so`` (``orbit_amd/host/orbit_assets.hpp``): a triangle mesh ->
the 32-byte Meshlet records and the packed meshlet data of ``assets::mesh::compute_meshlets``
(src/assets/mesh.rs:292-338), with meshoptimizer's cluster-bounds algorithm restated, and the mesh-level bounds of
``gltf_loader.rs:480-506``.  Python adds nothing; host only."""
import ctypes as C

import numpy as np

from . import layouts as L
from .passes import _check, lib

MAX_MESHLET_VERTICES = 64   # mesh.rs:8
MAX_MESHLET_TRIANGLES = 64  # mesh.rs:9


def compute_meshlets(positions, indices, material=0, vertex_offset=0, data_offset_base=0):
    """-> (meshlets: np[layouts.MESHLET], meshlet_data: np.uint32).  `data_offset_base` = words of meshlet data already
    in the scene-wide buffer (the meshlets' data_offset continue from there)."""
    pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
    h = lib()
    n_m, n_w = C.c_uint64(), C.c_uint64()
    args = (pos.ctypes.data_as(C.c_void_p), C.c_uint64(len(pos)), idx.ctypes.data_as(C.c_void_p), C.c_uint64(len(idx)),
            C.c_uint32(material), C.c_uint32(vertex_offset), C.c_uint32(data_offset_base))
    _check(h.orbit_host_compute_meshlets(*args, None, None, C.byref(n_m), C.byref(n_w)))
    meshlets = np.zeros(n_m.value, dtype=L.MESHLET)
    data = np.zeros(max(n_w.value, 1), dtype=np.uint32)
    _check(h.orbit_host_compute_meshlets(*args, meshlets.ctypes.data_as(C.c_void_p), data.ctypes.data_as(C.c_void_p),
                                         C.byref(n_m), C.byref(n_w)))
    return meshlets, data[:n_w.value]


def compute_mesh_bounds(positions):
    """-> (aabb_min[3], aabb_max[3], bounding_sphere[4]) as in gltf_loader.rs:480-506."""
    pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
    mn, mx, sp = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 4)()
    lib().orbit_host_compute_mesh_bounds(pos.ctypes.data_as(C.c_void_p), C.c_uint64(len(pos)), mn, mx, sp)
    return np.array(mn, np.float32), np.array(mx, np.float32), np.array(sp, np.float32)


def meshlet_triangles(meshlet, meshlet_data):
    """The global vertex indices (n_tri, 3) of one meshlet, decoded the way the mesh / vertex shaders read them:
    vertices at data_offset, u8 corners from byte (data_offset + vertex_count) * 4 — the cmd_first_index of the draw
    command the cull path emits (meshlet_cull.comp:216-230)."""
    d0, nv, nt = int(meshlet["data_offset"]), int(meshlet["vertex_count"]), int(meshlet["triangle_count"])
    verts = meshlet_data[d0:d0 + nv]
    corners = meshlet_data.view(np.uint8)[(d0 + nv) * 4:(d0 + nv) * 4 + 3 * nt].reshape(nt, 3)
    return verts[corners]
