"""ctypes binding of the CPU oracle (``oracle/orbit_oracle.c``).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg — never by anything under ``orbit_amd/``.
All arguments are host numpy arrays; outputs are fresh numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborbit_oracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("orbit_oracle.c", "orbit_oracle_mt.c")]
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)


NATIVE_LIB_PATH = os.path.join(_HERE, "liborbit_oracle_native.so")
NATIVE_CFLAGS = "-O3 -march=native -ffp-contract=off -fno-fast-math -fopenmp"  # oracle/Makefile NATIVE_CFLAGS
_native = None


def lib_native():
    """The same sources built -O3 -march=native ON THIS HOST (cpu_baseline of bench.py; SURVEY.md §8d flags)."""
    global _native
    if _native is None:
        subprocess.run(["make", "-C", _HERE, "-s", "-B", "native"], check=True)  # always for this host's CPU
        _native = C.CDLL(NATIVE_LIB_PATH)
    return _native


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orbit_log2f.restype = C.c_float
        _lib.orbit_log2f.argtypes = [C.c_float]
        _lib.oracle_hiz_sample.restype = C.c_float
        _lib.oracle_hiz_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float]
        for n in ("oracle_entity_cull", "oracle_meshlet_cull", "oracle_cluster_compact", "oracle_cluster_assign",
                  "oracle_sphere_frustum_cull", "oracle_mip_levels_from_size"):
            getattr(_lib, n).restype = C.c_uint32
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _bytes(a):
    return np.ascontiguousarray(a).view(np.uint8).reshape(-1)


class PyramidDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("mip_levels", C.c_uint32),
                ("total_texels", C.c_uint32), ("mip_offset", C.c_uint32 * 16), ("mip_width", C.c_uint32 * 16),
                ("mip_height", C.c_uint32 * 16)]


def log2f(x):
    return float(lib().orbit_log2f(C.c_float(x)))


def depth_pyramid_desc(sw, sh):
    d = PyramidDesc()
    lib().oracle_depth_pyramid_desc(C.c_uint32(sw), C.c_uint32(sh), C.byref(d))
    return d


def frustum_planes_from_matrix(m):
    m = np.ascontiguousarray(m, dtype=np.float32).reshape(16)
    out = np.zeros((6, 4), dtype=np.float32)
    lib().oracle_frustum_planes_from_matrix(_p(m), _p(out))
    return out


def depth_reduce(depth, sw, sh):
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    d = depth_pyramid_desc(sw, sh)
    pyr = np.zeros(d.total_texels, dtype=np.float32)
    lib().oracle_depth_reduce(_p(depth), C.c_uint32(sw), C.c_uint32(sh), _p(pyr))
    return pyr, d


def hiz_sample(pyr, w0, h0, u, v, lod):
    pyr = np.ascontiguousarray(pyr, dtype=np.float32)
    return float(lib().oracle_hiz_sample(_p(pyr), w0, h0, C.c_float(u), C.c_float(v), C.c_float(lod)))


class dispatch_size:
    """`with oracle.dispatch_size(64): ...` — MESHLET_DISPATCH_SIZE of the cull shaders for the calls inside (pass 0 of
    entity_cull / meshlet_cull only; 32 outside)."""

    def __init__(self, s):
        self.s = int(s)

    def __enter__(self):
        lib().oracle_set_dispatch_size(C.c_uint32(self.s))
        return self

    def __exit__(self, *exc):
        lib().oracle_set_dispatch_size(C.c_uint32(32))
        return False


class arith_profile:
    """`with oracle.arith_profile(1): ...` — the cull shaders' Dot / matrix products / Length as fma chains
    (OrbitCaps.arith_profile = ORBIT_ARITH_CONTRACTED) for the calls inside; canonical (0) outside."""

    def __init__(self, profile):
        self.profile = int(profile)

    def __enter__(self):
        lib().oracle_set_arith_profile(C.c_uint32(self.profile))
        return self

    def __exit__(self, *exc):
        lib().oracle_set_arith_profile(C.c_uint32(0))
        return False


def entity_cull(cull_info, entity_draw_buffer, entity_draw_count, mesh_infos, entities, dispatch_capacity,
                visibility=None, pyramid=None, pyramid_size=(0, 0), draw_first=None):
    """Returns (dispatch_buffer bytes, visibility words (copy, updated in pass 2), dropped).
    draw_first=None: whole-buffer call (reference grid); else the exact shard [draw_first, +count)."""
    ci = _bytes(cull_info)
    assert ci.nbytes == 400
    edb = _bytes(entity_draw_buffer)
    mi, en = _bytes(mesh_infos), _bytes(entities)
    out = np.zeros(12 + 16 * dispatch_capacity, dtype=np.uint8)
    vis = None if visibility is None else np.array(visibility, dtype=np.uint32, copy=True)
    pyr = None if pyramid is None else np.ascontiguousarray(pyramid, dtype=np.float32)
    exact = 0 if draw_first is None else 1
    dropped = lib().oracle_entity_cull(_p(ci), _p(edb), C.c_uint32(draw_first or 0), C.c_uint32(entity_draw_count),
                                       C.c_uint32(exact), _p(mi),
                                       _p(en), _p(out), C.c_uint32(dispatch_capacity), _p(vis), _p(pyr),
                                       C.c_uint32(pyramid_size[0]), C.c_uint32(pyramid_size[1]))
    return out, vis, int(dropped)


def meshlet_cull(cull_info, dispatch_buffer, meshlets, draw_capacity, entities, materials, meshlet_visibility=None,
                 pyramid=None, pyramid_size=(0, 0)):
    """Returns (draw_buffer bytes, meshlet visibility words (copy), dropped)."""
    ci = _bytes(cull_info)
    assert ci.nbytes == 400
    db, ml, en, mt = _bytes(dispatch_buffer), _bytes(meshlets), _bytes(entities), _bytes(materials)
    out = np.zeros(4 + 28 * draw_capacity, dtype=np.uint8)
    vis = None if meshlet_visibility is None else np.array(meshlet_visibility, dtype=np.uint32, copy=True)
    pyr = None if pyramid is None else np.ascontiguousarray(pyramid, dtype=np.float32)
    dropped = lib().oracle_meshlet_cull(_p(ci), _p(db), _p(ml), _p(out), C.c_uint32(draw_capacity), _p(en), _p(mt),
                                        _p(vis), _p(pyr), C.c_uint32(pyramid_size[0]), C.c_uint32(pyramid_size[1]))
    return out, vis, int(dropped)


def meshlet_task_cull(cull_info, dispatch_buffer, meshlets, entities, materials, meshlet_visibility=None,
                      pyramid=None, pyramid_size=(0, 0)):
    """Task-shader path (forward_depth_prepass.task): returns (44-B task records as a structured array,
    meshlet visibility words (copy))."""
    from orbit_amd import layouts as L

    ci = _bytes(cull_info)
    assert ci.nbytes == 400
    db, ml, en, mt = _bytes(dispatch_buffer), _bytes(meshlets), _bytes(entities), _bytes(materials)
    nrec = int(db[:4].view("<u4")[0])
    out = np.zeros(nrec, dtype=L.MESH_TASK_RECORD)
    vis = None if meshlet_visibility is None else np.array(meshlet_visibility, dtype=np.uint32, copy=True)
    pyr = None if pyramid is None else np.ascontiguousarray(pyramid, dtype=np.float32)
    lib().oracle_meshlet_task_cull(_p(ci), _p(db), _p(ml), _p(en), _p(mt), _p(vis), _p(pyr),
                                   C.c_uint32(pyramid_size[0]), C.c_uint32(pyramid_size[1]), _p(out))
    return out, vis


def max_threads():
    return int(lib().oracle_mt_max_threads())


def cull_frame_mt(cull_info, entity_draw_buffer, entity_draw_count, mesh_infos, entities, meshlets, materials,
                  dispatch_capacity, draw_capacity, threads, visibility=None, meshlet_visibility=None, pyramid=None,
                  pyramid_size=(0, 0), native=False, out=None):
    """entity_cull + meshlet_cull of one frame on `threads` OpenMP threads (orbit_oracle_mt.c: the sequential oracle
    on contiguous shards, concatenated).  Returns (dispatch bytes, draw bytes, entity vis, meshlet vis, dropped[2])."""
    ci = _bytes(cull_info)
    assert ci.nbytes == 400
    edb, mi, en = _bytes(entity_draw_buffer), _bytes(mesh_infos), _bytes(entities)
    ml, mt = _bytes(meshlets), _bytes(materials)
    if out is None:  # `out` = (dispatch, draw) buffers of a previous call, reused by timing loops
        disp = np.zeros(12 + 16 * dispatch_capacity, dtype=np.uint8)
        draw = np.zeros(4 + 28 * draw_capacity, dtype=np.uint8)
    else:
        disp, draw = out
    ev = None if visibility is None else np.array(visibility, dtype=np.uint32, copy=True)
    mv = None if meshlet_visibility is None else np.array(meshlet_visibility, dtype=np.uint32, copy=True)
    pyr = None if pyramid is None else np.ascontiguousarray(pyramid, dtype=np.float32)
    dropped = (C.c_uint32 * 2)()
    f = (lib_native() if native else lib()).oracle_cull_frame_mt
    f.restype = C.c_int32
    rc = f(_p(ci), _p(edb), C.c_uint32(entity_draw_count), _p(mi), _p(en), _p(ml), _p(mt), _p(disp),
           C.c_uint32(dispatch_capacity), _p(draw), C.c_uint32(draw_capacity), _p(ev), _p(mv), _p(pyr),
           C.c_uint32(pyramid_size[0]), C.c_uint32(pyramid_size[1]), C.c_int(threads), dropped)
    if rc != 0:
        raise MemoryError("oracle_cull_frame_mt: allocation failed")
    return disp, draw, ev, mv, (int(dropped[0]), int(dropped[1]))


def cluster_grid_info(near, far, z_slices):
    zs, zb = C.c_float(), C.c_float()
    lib().oracle_cluster_grid_info(C.c_float(near), C.c_float(far), C.c_uint32(z_slices), C.byref(zs), C.byref(zb))
    return zs.value, zb.value


def cluster_mark(push, depth):
    pc = _bytes(push)
    assert pc.nbytes == 56
    cc = pc[:12].view("<u4")
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    masks = np.zeros(int(cc[0]) * int(cc[1]), dtype=np.uint32)
    bounds = np.zeros((int(cc[0]) * int(cc[1]) * int(cc[2]), 2), dtype=np.uint32)
    lib().oracle_cluster_mark(_p(pc), _p(depth), _p(masks), _p(bounds))
    return masks, bounds


def cluster_compact(cluster_count, masks, index_capacity):
    cc = np.asarray(cluster_count, dtype=np.uint32)
    masks = np.ascontiguousarray(masks, dtype=np.uint32)
    out = np.zeros(16 + 4 * index_capacity, dtype=np.uint8)
    dropped = lib().oracle_cluster_compact(_p(cc), _p(masks), _p(out), C.c_uint32(index_capacity))
    return out, int(dropped)


def project_sphere(sphere, p00, p11):
    """project_sphere (entity_cull.comp:83-102) of a view-space sphere with z already negated -> uv-space aabb[4]."""
    s4 = np.ascontiguousarray(sphere, dtype=np.float32)
    out = np.zeros(4, dtype=np.float32)
    lib().oracle_project_sphere(_p(s4), C.c_float(p00), C.c_float(p11), _p(out))
    return out


def cluster_aabb(info, bounds, cluster_index):
    ib = _bytes(info)
    bounds = np.ascontiguousarray(bounds, dtype=np.uint32)
    mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().oracle_cluster_aabb(_p(ib), _p(bounds), C.c_uint32(cluster_index), _p(mn), _p(mx))
    return mn, mx


def cluster_assign(info, unique, bounds, lights, light_index_capacity, total_clusters, offset_image=None):
    ib = _bytes(info)
    assert ib.nbytes == 192
    un, lt = _bytes(unique), _bytes(lights)
    bounds = np.ascontiguousarray(bounds, dtype=np.uint32)
    out = np.zeros(4 + 4 * light_index_capacity, dtype=np.uint8)
    img = np.zeros((total_clusters, 2), dtype=np.uint32) if offset_image is None else np.array(
        offset_image, dtype=np.uint32, copy=True)
    dropped = lib().oracle_cluster_assign(_p(ib), _p(un), _p(bounds), _p(lt), _p(out),
                                          C.c_uint32(light_index_capacity), _p(img))
    return out, img, int(dropped)


def sphere_frustum_cull(view, planes, spheres):
    view = np.ascontiguousarray(view, dtype=np.float32).reshape(16)
    planes = np.ascontiguousarray(planes, dtype=np.float32).reshape(-1, 4)
    spheres = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 4)
    n = len(spheres)
    bits = np.zeros((n + 31) // 32, dtype=np.uint32)
    cnt = lib().oracle_sphere_frustum_cull(_p(view), _p(planes), C.c_uint32(len(planes)), _p(spheres), C.c_uint32(n),
                                           _p(bits))
    return bits, int(cnt)
