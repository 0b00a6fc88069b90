"""Camera -> GpuCullInfo for callers that have no renderer around them (bench.py, smoke(), tools/).

Nothing is computed here: every number comes from the C++ host mirror (``liborbit_host.so`` through
``orbit_amd/passes.py``), composed the way the reference's frame does it — ``Camera::compute_matrix`` /
``Projection::compute_matrix`` (src/camera.rs:85-137), ``frustum_planes_from_matrix`` + ``normalize_plane``
(src/math.rs:71-89, planes [0..5) as in src/passes/forward.rs:264-268) and ``CullInfo::to_gpu``
(src/passes/draw_gen.rs:121-203).
"""
import numpy as np

from . import passes


def look_down_minus_z(position, fov_deg=90.0, aspect=16.0 / 9.0, z_near=0.01):
    """(view[16], proj[16], Projection) of the reference's default camera (src/app.rs:744-752: fov 90 deg, near 0.01)
    at `position` with identity orientation; column-major, view = inverse(model)."""
    projection = passes.Projection.Perspective(float(np.deg2rad(np.float32(fov_deg))), z_near)
    proj = passes.projection_compute_matrix(projection, aspect)
    model = np.eye(4, dtype=np.float32)
    model[:3, 3] = position
    view = passes.mat4_inverse(np.ascontiguousarray(model.T).reshape(16))
    return view, proj, projection


def frame_cull_info(position, n_planes=5, fov_deg=90.0, aspect=16.0 / 9.0, z_near=0.01, **cull_info_fields):
    """The 400-B GpuCullInfo of a frustum(+cone) cull from `position`: a numpy record of layouts.GPU_CULL_INFO."""
    view, proj, projection = look_down_minus_z(position, fov_deg, aspect, z_near)
    planes = passes.frustum_planes_from_matrix(proj)[:n_planes]
    host_ci = passes.CullInfo(view, planes, projection, aspect_ratio=aspect, **cull_info_fields)
    return passes.cull_info_to_gpu(host_ci)
