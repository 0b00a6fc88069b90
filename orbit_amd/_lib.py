"""ctypes binding of ``liborbit_cull.so`` (the C ABI of ``include/orbit_abi.h``).

There is exactly one compute path: the HIP library.  If it is missing or no
gfx950 device is usable, loading / context creation raises — nothing here falls
back to a host implementation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborbit_cull.so")

OK, E_INVALID, E_PLANES, E_CAPACITY, E_HIP, E_NO_DEVICE, E_TIMEOUT, E_MISSING, E_COMM, E_RANGE, E_STALE = (
    0, -1, -2, -3, -4, -5, -6, -7, -8, -9, -10)
ERROR_NAMES = {
    E_INVALID: "ORBIT_E_INVALID", E_PLANES: "ORBIT_E_PLANES", E_CAPACITY: "ORBIT_E_CAPACITY", E_HIP: "ORBIT_E_HIP",
    E_NO_DEVICE: "ORBIT_E_NO_DEVICE", E_TIMEOUT: "ORBIT_E_TIMEOUT", E_MISSING: "ORBIT_E_MISSING",
    E_COMM: "ORBIT_E_COMM", E_RANGE: "ORBIT_E_RANGE", E_STALE: "ORBIT_E_STALE",
}

MAX_PYRAMID_MIPS = 16


class OrbitError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code


class Caps(C.Structure):
    _fields_ = [("max_entities", C.c_uint32), ("max_dispatches", C.c_uint32), ("max_draws", C.c_uint32),
                ("max_lights", C.c_uint32), ("max_clusters", C.c_uint32), ("dispatch_size", C.c_uint32),
                ("max_views", C.c_uint32), ("validate_streams", C.c_uint32), ("cull_path", C.c_uint32),
                ("arith_profile", C.c_uint32)]


class EntityCullBufs(C.Structure):
    _fields_ = [("entity_draw_buffer", C.c_void_p), ("mesh_info_buffer", C.c_void_p),
                ("meshlet_dispatch_buffer", C.c_void_p), ("entity_buffer", C.c_void_p),
                ("visibility_buffer", C.c_void_p), ("depth_pyramid", C.c_void_p),
                ("depth_pyramid_size", C.c_uint32 * 2), ("dispatch_capacity", C.c_uint32), ("_pad", C.c_uint32),
                ("depth_pyramid_levels", C.c_void_p)]


class MeshletCullBufs(C.Structure):
    _fields_ = [("meshlet_dispatch_buffer", C.c_void_p), ("meshlet_buffer", C.c_void_p),
                ("draw_commands_buffer", C.c_void_p), ("entity_buffer", C.c_void_p), ("material_buffer", C.c_void_p),
                ("meshlet_visibility_buffer", C.c_void_p), ("depth_pyramid", C.c_void_p),
                ("depth_pyramid_size", C.c_uint32 * 2), ("dispatch_capacity", C.c_uint32),
                ("draw_capacity", C.c_uint32), ("material_count", C.c_uint32), ("_pad", C.c_uint32),
                ("depth_pyramid_levels", C.c_void_p)]


class CullView(C.Structure):  # OrbitCullView
    _fields_ = [("cull_info", C.c_void_p), ("entity", EntityCullBufs), ("meshlet", MeshletCullBufs),
                ("entity_draw_count", C.c_uint32), ("skip_meshlet_stage", C.c_uint32)]


class DepthPyramidLevel(C.Structure):  # OrbitDepthPyramidLevel: one mip of a pyramid made of separate images
    _fields_ = [("texels", C.c_void_p), ("row_pitch", C.c_uint32), ("_pad", C.c_uint32)]


class DepthReduceItem(C.Structure):  # OrbitDepthReduceItem
    _fields_ = [("depth", C.c_void_p), ("screen_width", C.c_uint32), ("screen_height", C.c_uint32),
                ("depth_row_pitch", C.c_uint32), ("_pad", C.c_uint32), ("pyramid", C.c_void_p),
                ("levels", C.POINTER(DepthPyramidLevel))]


class DepthPyramidDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("mip_levels", C.c_uint32),
                ("total_texels", C.c_uint32), ("mip_offset", C.c_uint32 * MAX_PYRAMID_MIPS),
                ("mip_width", C.c_uint32 * MAX_PYRAMID_MIPS), ("mip_height", C.c_uint32 * MAX_PYRAMID_MIPS)]


class ClusterFrame(C.Structure):  # OrbitClusterFrame
    _fields_ = [("push", C.c_void_p), ("info", C.c_void_p), ("depth", C.c_void_p), ("lights", C.c_void_p),
                ("tile_depth_slice_mask", C.c_void_p), ("depth_bounds", C.c_void_p), ("unique_cluster_buffer", C.c_void_p),
                ("light_index_buffer", C.c_void_p), ("cluster_offset_image", C.c_void_p), ("index_capacity", C.c_uint32),
                ("light_index_capacity", C.c_uint32)]


class FrameLate(C.Structure):  # OrbitFrameLate
    _fields_ = [("pyramids", C.POINTER(DepthReduceItem)), ("late_views", C.POINTER(CullView)),
                ("cascade_views", C.POINTER(CullView)), ("clusters", C.POINTER(ClusterFrame)),
                ("pyramid_count", C.c_uint32), ("late_view_count", C.c_uint32), ("cascade_view_count", C.c_uint32),
                ("_pad", C.c_uint32)]


# every symbol include/orbit_abi.h and orbit_abi_ext.h declare: (restype, argtypes)
SYMBOLS = {
    "orbit_abi_version": (C.c_uint32, []),
    "orbit_meshlet_cull_records_and_commands": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(MeshletCullBufs),
                                                            C.c_void_p, C.c_uint32, C.c_void_p]),
    "orbit_exchange_list": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "orbit_ctx_fused_culls": (C.c_uint64, [C.c_void_p]),
    "orbit_frame_late": (C.c_int32, [C.c_void_p, C.POINTER(FrameLate), C.c_void_p]),
    "orbit_default_caps": (None, [C.POINTER(Caps)]),
    "orbit_ctx_create": (C.c_int32, [C.c_int32, C.POINTER(Caps), C.POINTER(C.c_void_p)]),
    "orbit_ctx_destroy": (C.c_int32, [C.c_void_p]),
    "orbit_last_error": (C.c_char_p, [C.c_void_p]),
    "orbit_ctx_status": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32]),
    "orbit_entity_cull": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(EntityCullBufs), C.c_uint32, C.c_void_p]),
    "orbit_entity_cull_range": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(EntityCullBufs), C.c_uint32,
                                            C.c_uint32, C.c_void_p]),
    "orbit_meshlet_cull": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(MeshletCullBufs), C.c_void_p]),
    "orbit_meshlet_task_cull": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(MeshletCullBufs), C.c_void_p, C.c_void_p]),
    "orbit_depth_pyramid_desc": (C.c_int32, [C.c_uint32, C.c_uint32, C.POINTER(DepthPyramidDesc)]),
    "orbit_cull_views": (C.c_int32, [C.c_void_p, C.POINTER(CullView), C.c_uint32, C.c_void_p]),
    "orbit_depth_reduce_multi": (C.c_int32, [C.c_void_p, C.POINTER(DepthReduceItem), C.c_uint32, C.c_void_p]),
    "orbit_depth_pyramid_desc_from_mip0": (C.c_int32, [C.c_uint32, C.c_uint32, C.POINTER(DepthPyramidDesc)]),
    "orbit_depth_reduce": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "orbit_cluster_mark": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orbit_cluster_compact": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint32 * 3), C.c_void_p, C.c_void_p, C.c_uint32,
                                          C.c_void_p]),
    "orbit_cluster_assign": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_uint32, C.c_void_p, C.c_void_p]),
    "orbit_compute_clusters": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                           C.c_void_p]),
    "orbit_ctx_profile": (C.c_int32, [C.c_void_p, C.c_int32]),
    "orbit_ctx_profile_reserve": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "orbit_ctx_profile_read": (C.c_int32, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "orbit_shard_range": (None, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "orbit_gather_visible": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                         C.c_uint32, C.c_void_p]),
    "orbit_meshlet_cull_visible_records": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(MeshletCullBufs), C.c_void_p,
                                                       C.c_uint32, C.c_void_p]),
    "orbit_expand_visible_records": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                                 C.c_void_p]),
    "orbit_p2p_alloc": (C.c_int32, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.c_void_p]),
    "orbit_p2p_free": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "orbit_p2p_open": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "orbit_p2p_close": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "orbit_meshlet_stream_create": (C.c_int32, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]),
    "orbit_meshlet_stream_update": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                                C.c_void_p]),
    "orbit_meshlet_stream_set_materials": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "orbit_meshlet_stream_validate": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orbit_meshlet_stream_destroy": (C.c_int32, [C.c_void_p]),
    "orbit_meshlet_stream_update_meshes": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "orbit_ctx_mesh_side_culls": (C.c_uint64, [C.c_void_p]),
    "orbit_ctx_bind_meshlet_stream": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "orbit_ctx_meshlet_stream_culls": (C.c_uint64, [C.c_void_p]),
    "orbit_ctx_meshlet_class_culls": (C.c_uint64, [C.c_void_p]),
    "orbit_cull_shard": (C.c_int32, [C.c_void_p, C.c_void_p, C.POINTER(EntityCullBufs), C.c_uint32, C.c_uint32,
                                     C.POINTER(MeshletCullBufs), C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "orbit_ctx_shard_culls": (C.c_uint64, [C.c_void_p]),
    "orbit_allgather_list": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                         C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "orbit_compact_segments": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                           C.c_uint32, C.c_uint32, C.c_void_p]),
}

_lib = None


def load_variant(path):
    """A second build of the library under another path (A/B tools: tools/ab_libs.py); not cached."""
    import torch  # noqa: F401  (same HIP runtime as the main copy)

    lib = C.CDLL(path)
    for name, (restype, argtypes) in SYMBOLS.items():
        if hasattr(lib, name):  # an older build may lack the newest entry points
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
    return lib


def load():
    """Loads the shared library once; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} is missing: build it with `make -C orbit_amd/csrc` or "
                      "`python -c 'import __graft_entry__ as g; g.build()'` (there is no fallback path)")
    try:
        # torch ships its own libamdhip64; if it is going to be used in this process it has to be the copy the
        # library binds to as well (two HIP runtimes in one process: the second one finds no device)
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, ctx=None):
    if rc != OK:
        msg = load().orbit_last_error(ctx)
        raise OrbitError(rc, msg.decode() if msg else "")
