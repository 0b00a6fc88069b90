"""ctypes binding of the asset ingestion in ``liborbit_host.so`` (``orbit_amd/host/orbit_gltf.hpp``): glTF 2.0
(.glb / .gltf) -> materials, meshes with their LOD chains (``GpuAssets::add_mesh``, src/assets/mod.rs:325-476) and
entities (``load_gltf``, src/gltf_loader.rs:511-676), then through ``SceneData::update_scene`` into the six buffers the
cull path reads.  Python adds nothing; host only."""
import ctypes as C

import numpy as np

from . import layouts as L
from . import scene as host_scene
from .passes import Panic, lib


class Counts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("meshes", "meshlets", "meshlet_data_words", "materials", "vertices", "entities")]


ENTITY = np.dtype([("mesh", "<i4"), ("position", "<f4", (3,)), ("orientation", "<f4", (4,)), ("scale", "<f4", (3,))])


def _array(ptr, count, dtype):
    if count == 0:
        return np.zeros(0, dtype=dtype)
    n = count * np.dtype(dtype).itemsize
    return np.frombuffer((C.c_uint8 * n).from_address(ptr), dtype=dtype).copy()


def load(path):
    """-> dict(mesh_infos, meshlets, meshlet_data, materials, vertex_positions, entities): numpy copies."""
    h = lib()
    h.orbit_host_gltf_load.restype = C.c_void_p
    for f in ("mesh_infos", "meshlets", "meshlet_data", "materials", "vertex_positions"):
        getattr(h, f"orbit_host_gltf_{f}").restype = C.c_void_p
        getattr(h, f"orbit_host_gltf_{f}").argtypes = [C.c_void_p]
    h.orbit_host_gltf_entities.restype = C.c_uint64
    h.orbit_host_gltf_entities.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    h.orbit_host_gltf_counts.argtypes = [C.c_void_p, C.c_void_p]
    h.orbit_host_gltf_free.argtypes = [C.c_void_p]
    s = h.orbit_host_gltf_load(str(path).encode())
    if not s:
        raise Panic(h.orbit_host_last_error().decode())
    try:
        c = Counts()
        h.orbit_host_gltf_counts(s, C.byref(c))
        ents = np.zeros(c.entities, dtype=ENTITY)
        h.orbit_host_gltf_entities(s, ents.ctypes.data_as(C.c_void_p), c.entities)
        return dict(mesh_infos=_array(h.orbit_host_gltf_mesh_infos(s), c.meshes, L.MESH_INFO),
                    meshlets=_array(h.orbit_host_gltf_meshlets(s), c.meshlets, L.MESHLET),
                    meshlet_data=_array(h.orbit_host_gltf_meshlet_data(s), c.meshlet_data_words, np.uint32),
                    materials=_array(h.orbit_host_gltf_materials(s), c.materials, L.MATERIAL),
                    vertex_positions=_array(h.orbit_host_gltf_vertex_positions(s), c.vertices * 3, np.float32).reshape(-1, 3),
                    entities=ents)
    finally:
        h.orbit_host_gltf_free(s)


def simplify_clustered(positions, indices, target_index_count):
    """The stand-in of meshopt::simplify in the LOD chain (vertex clustering; re-indexes existing vertices)."""
    pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
    out = np.zeros(max(len(idx), 3), dtype=np.uint32)
    h = lib()
    h.orbit_host_simplify_clustered.restype = C.c_uint64
    need = C.c_uint64()
    n = h.orbit_host_simplify_clustered(pos.ctypes.data_as(C.c_void_p), C.c_uint64(len(pos)),
                                        idx.ctypes.data_as(C.c_void_p), C.c_uint64(len(idx)),
                                        C.c_uint64(int(target_index_count)), out.ctypes.data_as(C.c_void_p),
                                        C.c_uint64(len(out)), C.byref(need))
    assert need.value == n
    return out[:n].copy()


def transform_from_mat4(matrix_cols):
    """Transform::from_mat4 (scene.rs:41-48): column-major float[16] -> (position[3], orientation xyzw[4], scale[3])."""
    m = (C.c_float * 16)(*np.asarray(matrix_cols, dtype=np.float32).reshape(16))
    p, q, s = (C.c_float * 3)(), (C.c_float * 4)(), (C.c_float * 3)()
    lib().orbit_host_transform_from_mat4(m, p, q, s)
    return np.array(p, np.float32), np.array(q, np.float32), np.array(s, np.float32)


def to_scene_buffers(loaded, visibility_chunk_count=0):
    """The loaded document through SceneData (scene.rs:398-492): every node becomes an entity, `update_scene` builds
    EntityDrawBuffer / EntityData[] for the ones with a mesh.  -> dict of the cull path's input buffers
    (the sections of an .orbitscene file) + `scene` (the host SceneData, for later transform updates)."""
    sd = host_scene.SceneData(visibility_chunk_count)
    for e in loaded["entities"]:
        sd.add_entity(position=tuple(e["position"]), orientation=tuple(e["orientation"]), scale=tuple(e["scale"]),
                      mesh=None if e["mesh"] < 0 else int(e["mesh"]))
    sd.update_scene(loaded["mesh_infos"])
    return dict(entity_draws=sd.entity_draw_buffer_bytes(), mesh_infos=loaded["mesh_infos"],
                entities=sd.entity_data_cache(), meshlets=loaded["meshlets"], materials=loaded["materials"],
                meshlet_data=loaded["meshlet_data"], vertex_positions=loaded["vertex_positions"], scene=sd)
