"""ctypes binding of ``liborbit_host.so`` — the C++ host-side mirror of the reference's cull-pass
API (``src/passes/draw_gen.rs``, ``src/passes/cluster.rs``, ``src/math.rs``, ``src/camera.rs``).

Python adds nothing here: every function forwards to the C++ function of the same name
(``orbit_amd/host/orbit_host.hpp``), which derives parameters and calls the C ABI.
"""
import ctypes as C
import os

import numpy as np

from . import _lib, layouts as L

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborbit_host.so")
HOST_PANIC = -100
_host = None


class Panic(RuntimeError):
    """The reference would have panicked here (assert!/unwrap)."""


class Projection(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("fov", C.c_float), ("half_width", C.c_float), ("near_clip", C.c_float),
                ("far_clip", C.c_float)]

    @staticmethod
    def Perspective(fov, near_clip):  # camera.rs:76-79
        return Projection(0, fov, 0.0, near_clip, 0.0)

    @staticmethod
    def Orthographic(half_width, near_clip, far_clip):  # camera.rs:71-75
        return Projection(1, 0.0, half_width, near_clip, far_clip)


class HostCullInfo(C.Structure):
    _fields_ = [("view_matrix", C.c_float * 16), ("view_space_cull_planes", (C.c_float * 4) * 16),
                ("plane_count", C.c_uint32), ("projection", Projection), ("occlusion_kind", C.c_uint32),
                ("visibility_buffer", C.c_void_p), ("meshlet_visibility_buffer", C.c_void_p),
                ("depth_pyramid", C.c_void_p), ("depth_pyramid_size", C.c_uint32 * 2),
                ("noskip_alphamode", C.c_uint32), ("aspect_ratio", C.c_float), ("alpha_mode_filter", C.c_uint32),
                ("lod_range", C.c_uint32 * 2), ("lod_base", C.c_float), ("lod_step", C.c_float),
                ("lod_target_pos_view_space", C.c_float * 3)]


class HostShadowCascadeIn(C.Structure):
    _fields_ = [("shadow_resolution", C.c_uint32), ("cascade_split_lambda", C.c_float),
                ("max_shadow_distance", C.c_float), ("min_mesh_lod", C.c_uint32), ("max_mesh_lod", C.c_uint32),
                ("lod_base", C.c_float), ("lod_step", C.c_float), ("direction", C.c_float * 4),
                ("camera_position", C.c_float * 3), ("camera_orientation", C.c_float * 4), ("camera_fov", C.c_float),
                ("camera_near_clip", C.c_float), ("camera_aspect_ratio", C.c_float), ("cascade_index", C.c_uint32),
                ("frustum_culling", C.c_uint32)]


class HostAssets(C.Structure):
    _fields_ = [("meshlet_buffer", C.c_void_p), ("mesh_info_buffer", C.c_void_p), ("materials_buffer", C.c_void_p),
                ("material_count", C.c_uint32)]


class HostScene(C.Structure):
    _fields_ = [("entity_draw_count", C.c_uint64), ("entity_draw_buffer", C.c_void_p), ("entity_buffer", C.c_void_p),
                ("light_data_buffer", C.c_void_p), ("meshlet_visibility_buffer", C.c_void_p),
                ("light_count", C.c_uint64)]


class HostBuffer(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("size", C.c_uint64)]


class HostClusterSettings(C.Structure):
    _fields_ = [("px_size_power", C.c_uint32), ("screen_resolution", C.c_uint32 * 2), ("z_slice_count", C.c_uint32),
                ("far_plane", C.c_float), ("luminance_cutoff", C.c_float)]


class HostClusterDerived(C.Structure):
    _fields_ = [("tile_counts", C.c_uint64 * 2), ("cluster_counts", C.c_uint64 * 3),
                ("linear_cluster_count", C.c_uint64), ("linear_max_allocated_cluster_count", C.c_uint64),
                ("tile_px_size", C.c_uint32), ("z_scale", C.c_float), ("z_bias", C.c_float)]


class GpuClusterInfoBuffer(C.Structure):
    _fields_ = [("cluster_count", C.c_uint32 * 3), ("tile_size_px", C.c_uint32), ("screen_size", C.c_uint32 * 2),
                ("z_slice_count", C.c_uint32), ("z_scale", C.c_float), ("z_bias", C.c_float),
                ("luminance_cutoff", C.c_float), ("light_offset_image", C.c_uint32), ("light_index_list", C.c_uint32),
                ("tile_depth_slice_mask_buffer", C.c_uint32)]


class HostClusterOut(C.Structure):
    _fields_ = [("tile_depth_slice_mask", HostBuffer), ("depth_bounds", HostBuffer),
                ("unique_cluster_buffer", HostBuffer), ("light_offset_image", HostBuffer),
                ("light_index_list", HostBuffer), ("info", GpuClusterInfoBuffer)]


def lib():
    global _host
    if _host is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} is missing: build it with `make -C orbit_amd/host`")
        _lib.load()  # dependency (resolved through rpath as well)
        h = C.CDLL(LIB_PATH)
        h.orbit_host_last_error.restype = C.c_char_p
        h.orbit_host_mip_levels_from_size.restype = C.c_uint32
        h.orbit_host_context_create.restype = C.c_void_p
        h.orbit_host_context_create.argtypes = [C.c_int32, C.c_void_p, C.c_void_p]
        h.orbit_host_context_destroy.argtypes = [C.c_void_p]
        h.orbit_host_sync_meshlet_stream.restype = C.c_int32
        h.orbit_host_sync_meshlet_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        h.orbit_host_sync_meshlet_stream_materials.restype = C.c_int32
        h.orbit_host_sync_meshlet_stream_materials.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        _host = h
    return _host


def _check(rc):
    if rc == HOST_PANIC:
        raise Panic(lib().orbit_host_last_error().decode())
    if rc != 0:
        raise _lib.OrbitError(rc, lib().orbit_host_last_error().decode())


def _mat(a):
    return (C.c_float * 16)(*np.asarray(a, dtype=np.float32).reshape(16))


# ----------------------------------------------------------------------------- math.rs / camera.rs
def perspective_infinite_reverse_rh(fov_y, aspect, z_near):
    out = (C.c_float * 16)()
    lib().orbit_host_perspective_infinite_reverse_rh(C.c_float(fov_y), C.c_float(aspect), C.c_float(z_near), out)
    return np.array(out, dtype=np.float32)


def orthographic_rh(left, right, bottom, top, near, far):
    out = (C.c_float * 16)()
    lib().orbit_host_orthographic_rh(*[C.c_float(v) for v in (left, right, bottom, top, near, far)], out)
    return np.array(out, dtype=np.float32)


def projection_compute_matrix(projection, aspect):
    out = (C.c_float * 16)()
    lib().orbit_host_projection_compute_matrix(C.byref(projection), C.c_float(aspect), out)
    return np.array(out, dtype=np.float32)


def mat4_inverse(m):
    out = (C.c_float * 16)()
    lib().orbit_host_mat4_inverse(_mat(m), out)
    return np.array(out, dtype=np.float32)


def frustum_planes_from_matrix(m, normalize=True):
    out = ((C.c_float * 4) * 6)()
    lib().orbit_host_frustum_planes_from_matrix(_mat(m), out, 1 if normalize else 0)
    return np.array(out, dtype=np.float32).reshape(6, 4)


def mip_levels_from_size(n):
    return int(lib().orbit_host_mip_levels_from_size(C.c_uint32(n)))


def project_sphere_clip_space(sphere, znear, p00, p11):
    out = (C.c_float * 4)()
    ok = lib().orbit_host_project_sphere_clip_space((C.c_float * 4)(*sphere), C.c_float(znear), C.c_float(p00),
                                                    C.c_float(p11), out)
    return np.array(out, dtype=np.float32) if ok else None


# ----------------------------------------------------------------------------- CullInfo (draw_gen.rs:105-203)
OCCLUSION_NONE, VISIBILITY_READ, VISIBILITY_WRITE = 0, 1, 2


def _dptr(t):
    return None if t is None else C.c_void_p(t if isinstance(t, int) else t.data_ptr())


def CullInfo(view_matrix, view_space_cull_planes, projection, occlusion_kind=OCCLUSION_NONE, visibility_buffer=None,
             meshlet_visibility_buffer=None, depth_pyramid=None, depth_pyramid_size=(0, 0), noskip_alphamode=0,
             aspect_ratio=1.0, alpha_mode_filter=L.ALPHA_OPAQUE | L.ALPHA_MASKED, lod_range=(0, 8), lod_base=16.0,
             lod_step=2.0, lod_target_pos_view_space=(0.0, 0.0, 0.0)):
    ci = HostCullInfo()
    ci.view_matrix = _mat(view_matrix)
    planes = np.asarray(view_space_cull_planes, dtype=np.float32).reshape(-1, 4)
    ci.plane_count = len(planes)
    for i, p in enumerate(planes[:16]):
        ci.view_space_cull_planes[i] = (C.c_float * 4)(*p)
    ci.projection = projection
    ci.occlusion_kind = occlusion_kind
    ci.visibility_buffer = _dptr(visibility_buffer)
    ci.meshlet_visibility_buffer = _dptr(meshlet_visibility_buffer)
    ci.depth_pyramid = _dptr(depth_pyramid)
    ci.depth_pyramid_size[0], ci.depth_pyramid_size[1] = depth_pyramid_size
    ci.noskip_alphamode = noskip_alphamode
    ci.aspect_ratio = aspect_ratio
    ci.alpha_mode_filter = alpha_mode_filter
    ci.lod_range[0], ci.lod_range[1] = lod_range
    ci.lod_base, ci.lod_step = lod_base, lod_step
    ci.lod_target_pos_view_space = (C.c_float * 3)(*lod_target_pos_view_space)
    return ci


def shadow_cascade(direction, camera_position, camera_orientation, camera_fov, camera_near_clip, camera_aspect_ratio,
                   cascade_index, shadow_resolution=2048, cascade_split_lambda=0.80, max_shadow_distance=32.0,
                   min_mesh_lod=0, max_mesh_lod=7, lod_base=16.0, lod_step=2.0, frustum_culling=True):
    """ShadowRenderer::render_cascaded_shadow for one cascade (shadow_renderer.rs:466-706): returns
    (HostCullInfo, light_projection_matrix[16], shadow_map_world_size).  Quaternions are (x, y, z, w)."""
    a = HostShadowCascadeIn(shadow_resolution, cascade_split_lambda, max_shadow_distance, min_mesh_lod, max_mesh_lod,
                            lod_base, lod_step, (C.c_float * 4)(*direction), (C.c_float * 3)(*camera_position),
                            (C.c_float * 4)(*camera_orientation), camera_fov, camera_near_clip, camera_aspect_ratio,
                            cascade_index, 1 if frustum_culling else 0)
    ci = HostCullInfo()
    lpm = (C.c_float * 16)()
    ws = C.c_float()
    _check(lib().orbit_host_shadow_cascade(C.byref(a), C.byref(ci), lpm, C.byref(ws)))
    return ci, np.array(lpm, dtype=np.float32), float(ws.value)


def cull_info_to_gpu(ci):
    """CullInfo::to_gpu -> the 400-byte GpuCullInfo as a numpy record."""
    out = np.zeros((), dtype=L.GPU_CULL_INFO)
    _check(lib().orbit_host_cull_info_to_gpu(C.byref(ci), out.ctypes.data_as(C.c_void_p)))
    return out


def ClusterSettings(px_size_power=3, screen_resolution=(0, 0), z_slice_count=32, far_plane=200.0,
                    luminance_cutoff=0.25):  # cluster.rs:23-33 defaults
    s = HostClusterSettings()
    s.px_size_power = px_size_power
    s.screen_resolution[0], s.screen_resolution[1] = screen_resolution
    s.z_slice_count, s.far_plane, s.luminance_cutoff = z_slice_count, far_plane, luminance_cutoff
    return s


def cluster_settings_derive(settings, z_near):
    d = HostClusterDerived()
    lib().orbit_host_cluster_settings_derive(C.byref(settings), C.c_float(z_near), C.byref(d))
    return d


def compute_cluster_aabb(inverse_projection, screen_size, tile_size_px, cluster_count, z_near, z_far, cluster_id):
    """compute_cluster_aabb (cluster.rs:150-184), the reference's CPU twin of the cluster volume -> (min[3], max[3])."""
    out = (C.c_float * 6)()
    lib().orbit_host_compute_cluster_aabb(_mat(inverse_projection), (C.c_float * 2)(*[float(v) for v in screen_size]),
                                          C.c_float(tile_size_px), (C.c_float * 3)(*[float(v) for v in cluster_count]),
                                          C.c_float(z_near), C.c_float(z_far),
                                          (C.c_float * 3)(*[float(v) for v in cluster_id]), out)
    a = np.array(out, dtype=np.float32)
    return a[:3], a[3:]


# ----------------------------------------------------------------------------- graphics::Context slice + passes
class Context:
    def __init__(self, device=0, stream=None, **caps):
        import torch

        c = _lib.Caps()
        _lib.load().orbit_default_caps(C.byref(c))
        for k, v in caps.items():
            setattr(c, k, int(v))
        s = torch.cuda.current_stream() if stream is None else stream
        self._h = lib().orbit_host_context_create(device, C.c_void_p(s.cuda_stream), C.byref(c))
        if not self._h:
            raise Panic(lib().orbit_host_last_error().decode())

    def sync_meshlet_stream(self, meshlet_buffer, first, count, capacity=None):
        """Context::sync_meshlet_stream (orbit_host.hpp): the add_mesh-side hook of the derived meshlet streams."""
        cap = (meshlet_buffer.numel() * meshlet_buffer.element_size() // 32) if capacity is None else capacity
        _check(lib().orbit_host_sync_meshlet_stream(C.c_void_p(self._h), _dptr(meshlet_buffer), first, count, cap))

    def sync_meshlet_stream_materials(self, materials_buffer, material_count):
        """Context::sync_meshlet_stream_materials: the add_material-side hook (the stream's alpha classes)."""
        _check(lib().orbit_host_sync_meshlet_stream_materials(C.c_void_p(self._h), _dptr(materials_buffer), material_count))

    def close(self):
        if self._h:
            lib().orbit_host_context_destroy(C.c_void_p(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def Assets(meshlet_buffer, mesh_info_buffer, materials_buffer, material_count=0):
    return HostAssets(_dptr(meshlet_buffer), _dptr(mesh_info_buffer), _dptr(materials_buffer), material_count)


def Scene(entity_draw_count, entity_draw_buffer, entity_buffer, light_data_buffer=None,
          meshlet_visibility_buffer=None, light_count=0):
    return HostScene(entity_draw_count, _dptr(entity_draw_buffer), _dptr(entity_buffer), _dptr(light_data_buffer),
                     _dptr(meshlet_visibility_buffer), light_count)


def _tensor(buf):
    """uint8 torch view of a device buffer owned by the host Context."""
    import torch

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(buf.size),), "typestr": "|u1", "data": (int(buf.ptr), False),
                                  "version": 3}
    return torch.as_tensor(h, device="cuda")


def create_meshlet_dispatch_command(context, draw_commands_name, assets, scene, cull_info):
    gpu = np.zeros((), dtype=L.GPU_CULL_INFO)
    out = HostBuffer()
    _check(lib().orbit_host_create_meshlet_dispatch_command(C.c_void_p(context._h), draw_commands_name.encode(),
                                                            C.byref(assets), C.byref(scene), C.byref(cull_info),
                                                            gpu.ctypes.data_as(C.c_void_p), C.byref(out)))
    return gpu, out


def create_meshlet_draw_commands(context, draw_commands_name, assets, scene, cull_info, meshlet_dispatch_buffer):
    out = HostBuffer()
    _check(lib().orbit_host_create_meshlet_draw_commands(C.c_void_p(context._h), draw_commands_name.encode(),
                                                         C.byref(assets), C.byref(scene), C.byref(cull_info),
                                                         C.byref(meshlet_dispatch_buffer), C.byref(out)))
    return out


def create_draw_commands(context, draw_commands_name, assets, scene, cull_info):
    out = HostBuffer()
    _check(lib().orbit_host_create_draw_commands(C.c_void_p(context._h), draw_commands_name.encode(), C.byref(assets),
                                                 C.byref(scene), C.byref(cull_info), C.byref(out)))
    return out


def depth_pyramid_update(context, name, depth, width, height):
    """DepthPyramid::new/resize + update; returns (device pointer, (w0, h0, mips))."""
    ptr = C.c_void_p()
    dims = (C.c_uint32 * 3)()
    _check(lib().orbit_host_depth_pyramid_update(C.c_void_p(context._h), name.encode(), _dptr(depth), width, height,
                                                 C.byref(ptr), dims))
    return ptr.value, tuple(dims)


def compute_clusters(context, settings, view_matrix, projection, aspect_ratio, depth, depth_size, samples, scene):
    out = HostClusterOut()
    size = (C.c_uint32 * 2)(*depth_size)
    _check(lib().orbit_host_compute_clusters(C.c_void_p(context._h), C.byref(settings), _mat(view_matrix),
                                             C.byref(projection), C.c_float(aspect_ratio), _dptr(depth), size,
                                             samples, C.byref(scene), C.byref(out)))
    return out


buffer_tensor = _tensor
