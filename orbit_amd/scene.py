"""ctypes binding of the scene side of ``liborbit_host.so`` (``orbit_amd/host/orbit_scene.hpp``): the producers of the
cull path's input buffers — ``SceneData::update_scene`` (src/scene.rs:404-492), the meshlet-visibility word allocator
(src/collections/freelist_alloc.rs) and the generational arena under it (src/collections/arena.rs).

Python adds nothing: every method forwards to the C++ function of the same name.  Host only (no device work).
"""
import ctypes as C

import numpy as np

from . import layouts as L
from .passes import _check, lib

SKY, DIRECTIONAL, POINT = 0, 1, 2  # LightKind, scene.rs:136-141


class ArenaIndex(C.Structure):
    _fields_ = [("generation", C.c_uint32), ("slot", C.c_uint32)]

    def key(self):
        return (self.generation, self.slot)


class HostLight(C.Structure):
    _fields_ = [("color", C.c_float * 3), ("intensity", C.c_float), ("kind", C.c_uint32), ("param", C.c_float),
                ("irradiance_map_index", C.c_uint32), ("prefiltered_map_index", C.c_uint32),
                ("cast_shadows", C.c_uint32)]


class HostEntity(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("orientation", C.c_float * 4), ("scale", C.c_float * 3),
                ("mesh", C.c_int32), ("has_light", C.c_int32), ("light", HostLight), ("name", C.c_char_p)]


_bound = False


def _h():
    global _bound
    h = lib()
    if not _bound:
        h.orbit_host_arena_create.restype = C.c_void_p
        h.orbit_host_arena_destroy.argtypes = [C.c_void_p]
        h.orbit_host_arena_insert.restype = ArenaIndex
        h.orbit_host_arena_insert.argtypes = [C.c_void_p, C.c_int64]
        for f in (h.orbit_host_arena_get, h.orbit_host_arena_remove):
            f.argtypes = [C.c_void_p, ArenaIndex, C.POINTER(C.c_int64)]
        h.orbit_host_arena_has_index.argtypes = [C.c_void_p, ArenaIndex]
        h.orbit_host_arena_len.restype = C.c_uint64
        h.orbit_host_arena_len.argtypes = [C.c_void_p]
        h.orbit_host_arena_iter.restype = C.c_uint64
        h.orbit_host_arena_iter.argtypes = [C.c_void_p, C.POINTER(ArenaIndex), C.POINTER(C.c_int64), C.c_uint64]
        h.orbit_host_freelist_create.restype = C.c_void_p
        h.orbit_host_freelist_create.argtypes = [C.c_uint64]
        h.orbit_host_freelist_destroy.argtypes = [C.c_void_p]
        h.orbit_host_freelist_allocate.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(ArenaIndex), C.POINTER(C.c_uint64)]
        h.orbit_host_freelist_deallocate.argtypes = [C.c_void_p, ArenaIndex]
        h.orbit_host_freelist_blocks.restype = C.c_uint64
        h.orbit_host_freelist_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.c_uint64]
        h.orbit_host_scene_create.restype = C.c_void_p
        h.orbit_host_scene_create.argtypes = [C.c_uint64]
        h.orbit_host_scene_destroy.argtypes = [C.c_void_p]
        h.orbit_host_scene_add_entity.restype = C.c_int64
        h.orbit_host_scene_add_entity.argtypes = [C.c_void_p, C.POINTER(HostEntity)]
        h.orbit_host_scene_set_transform.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                                     C.POINTER(C.c_float)]
        h.orbit_host_scene_update.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_uint64]
        for f in (h.orbit_host_scene_entity_draws, h.orbit_host_scene_entity_data, h.orbit_host_scene_light_data):
            f.restype = C.c_void_p
            f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        h.orbit_host_scene_shadow_command_count.restype = C.c_uint64
        h.orbit_host_scene_shadow_command_count.argtypes = [C.c_void_p]
        _bound = True
    return h


class Arena:
    """collections::Arena<i64> (arena.rs:98-330)."""

    def __init__(self):
        self._p = C.c_void_p(_h().orbit_host_arena_create())

    def __del__(self):
        if getattr(self, "_p", None):
            _h().orbit_host_arena_destroy(self._p)
            self._p = None

    def insert(self, value):
        return _h().orbit_host_arena_insert(self._p, value)

    def get(self, index):
        out = C.c_int64()
        return out.value if _h().orbit_host_arena_get(self._p, index, C.byref(out)) else None

    def remove(self, index):
        out = C.c_int64()
        return out.value if _h().orbit_host_arena_remove(self._p, index, C.byref(out)) else None

    def has_index(self, index):
        return bool(_h().orbit_host_arena_has_index(self._p, index))

    def __len__(self):
        return int(_h().orbit_host_arena_len(self._p))

    def iter(self):
        n = len(self)
        idx = (ArenaIndex * max(n, 1))()
        val = (C.c_int64 * max(n, 1))()
        m = _h().orbit_host_arena_iter(self._p, idx, val, n)
        return [(idx[i].key(), val[i]) for i in range(m)]


class FreeListAllocator:
    """collections::FreeListAllocator (freelist_alloc.rs:22-121)."""

    def __init__(self, size):
        self._p = C.c_void_p(_h().orbit_host_freelist_create(size))

    def __del__(self):
        if getattr(self, "_p", None):
            _h().orbit_host_freelist_destroy(self._p)
            self._p = None

    def allocate(self, size):
        """-> (index, (start, end)) or None."""
        idx = ArenaIndex()
        rng = (C.c_uint64 * 2)()
        if not _h().orbit_host_freelist_allocate(self._p, size, C.byref(idx), rng):
            return None
        return idx, (int(rng[0]), int(rng[1]))

    def deallocate(self, index):
        _h().orbit_host_freelist_deallocate(self._p, index)

    def blocks(self):
        """[(start, end, free)] in address order."""
        cap = 16
        while True:
            r = (C.c_uint64 * (2 * cap))()
            f = (C.c_int32 * cap)()
            n = int(_h().orbit_host_freelist_blocks(self._p, r, f, cap))
            if n <= cap:
                return [(int(r[2 * i]), int(r[2 * i + 1]), bool(f[i])) for i in range(n)]
            cap = n


class SceneData:
    """scene::SceneData (scene.rs:358-492): entities in, the three upload caches out."""

    def __init__(self, visibility_chunk_count=0):
        self._p = C.c_void_p(_h().orbit_host_scene_create(visibility_chunk_count))

    def __del__(self):
        if getattr(self, "_p", None):
            _h().orbit_host_scene_destroy(self._p)
            self._p = None

    def add_entity(self, position=(0, 0, 0), orientation=(0, 0, 0, 1), scale=(1, 1, 1), mesh=None, light=None,
                   name=None):
        """light: dict(kind=POINT, color=(1,1,1), intensity=1, param=0.6, cast_shadows=False, ...) or None."""
        e = HostEntity()
        e.position[:] = position
        e.orientation[:] = orientation
        e.scale[:] = scale
        e.mesh = -1 if mesh is None else int(mesh)
        e.has_light = 0 if light is None else 1
        if light is not None:
            e.light.color[:] = light.get("color", (1.0, 1.0, 1.0))
            e.light.intensity = light.get("intensity", 1.0)
            e.light.kind = light.get("kind", POINT)
            e.light.param = light.get("param", 0.6)
            e.light.irradiance_map_index = light.get("irradiance_map_index", 0)
            e.light.prefiltered_map_index = light.get("prefiltered_map_index", 0)
            e.light.cast_shadows = 1 if light.get("cast_shadows", False) else 0
        e.name = None if name is None else name.encode()
        return int(_h().orbit_host_scene_add_entity(self._p, C.byref(e)))

    def set_transform(self, entity, position, orientation=(0, 0, 0, 1), scale=(1, 1, 1)):
        _check(_h().orbit_host_scene_set_transform(self._p, entity, (C.c_float * 3)(*position),
                                                   (C.c_float * 4)(*orientation), (C.c_float * 3)(*scale)))

    def update_scene(self, mesh_infos, luminance_cutoff=0.25, frame_index=0):
        """mesh_infos: numpy array of layouts.MESH_INFO.  Fills the caches (read them with the methods below)."""
        mi = np.ascontiguousarray(mesh_infos, dtype=L.MESH_INFO)
        _check(_h().orbit_host_scene_update(self._p, mi.ctypes.data_as(C.c_void_p), len(mi), luminance_cutoff,
                                            frame_index))

    def _cache(self, fn, dtype):
        n = C.c_uint64()
        p = fn(self._p, C.byref(n))
        if not n.value:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_uint8 * (n.value * dtype.itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dtype).copy()

    def entity_draw_cache(self):
        return self._cache(_h().orbit_host_scene_entity_draws, L.ENTITY_DRAW)

    def entity_data_cache(self):
        return self._cache(_h().orbit_host_scene_entity_data, L.ENTITY_DATA)

    def light_data_cache(self):
        return self._cache(_h().orbit_host_scene_light_data, L.LIGHT)

    def shadow_command_count(self):
        return int(_h().orbit_host_scene_shadow_command_count(self._p))

    def entity_draw_buffer_bytes(self):
        """u32 count + EntityDraw[] — the bytes update_scene writes to entity_draw_buffer (scene.rs:470-481)."""
        return L.entity_draw_buffer(self.entity_draw_cache())
