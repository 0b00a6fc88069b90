"""CPU tests of the C++ host mirror (orbit_amd/host) against the tests' own numpy helpers and
the oracle: the parameter derivations of src/math.rs, src/camera.rs, CullInfo::to_gpu and
ClusterSettings, with the reference's names."""
import numpy as np
import pytest

import scenes as sc
from orbit_amd import layouts as L, passes


def test_projection_matrices():
    fov, aspect, near = float(np.deg2rad(np.float32(90.0))), 16.0 / 9.0, 0.01
    got = passes.perspective_infinite_reverse_rh(fov, aspect, near)
    assert np.allclose(got, sc.mat4_cols(sc.perspective_infinite_reverse_rh(fov, aspect, near)), rtol=1e-6)
    assert got[11] == -1.0 and got[14] == np.float32(near) and got[10] == 0.0 and got[15] == 0.0
    o = passes.orthographic_rh(-25, 25, -10, 10, 80.0, 0.1)
    assert np.allclose(o, sc.mat4_cols(sc.orthographic_rh(-25, 25, -10, 10, 80.0, 0.1)), rtol=1e-6)
    # Projection::compute_matrix swaps far and near for reverse-Z (camera.rs:91-96)
    p = passes.projection_compute_matrix(passes.Projection.Orthographic(25.0, 0.1, 80.0), 2.5)
    assert np.allclose(p, passes.orthographic_rh(-25, 25, -10, 10, 80.0, 0.1))


def test_mat4_inverse_and_planes(oracle):
    cam = sc.default_camera(rot=(0.6, 0.8))
    proj = sc.mat4_cols(cam.proj)
    inv = passes.mat4_inverse(proj).reshape(4, 4).T
    assert np.allclose(inv @ cam.proj, np.eye(4), atol=1e-5)
    planes = passes.frustum_planes_from_matrix(proj)
    assert np.allclose(planes, oracle.frustum_planes_from_matrix(proj), rtol=1e-6, atol=1e-7)
    assert np.allclose(planes[:5], sc.frustum_planes(cam.proj, 5), rtol=1e-6, atol=1e-7)
    raw = passes.frustum_planes_from_matrix(proj, normalize=False)
    assert np.allclose(raw[4], [0, 0, -1, cam.z_near])  # r3 + r2 of the infinite reverse-Z matrix


@pytest.mark.parametrize("n,expect", [(1, 1), (2, 2), (1024, 11), (2048, 12), (3, 2)])
def test_mip_levels(n, expect, oracle):
    assert passes.mip_levels_from_size(n) == expect


def test_project_sphere_clip_space():
    assert passes.project_sphere_clip_space((0, 0, 0.5, 1.0), 0.01, 1.0, 1.0) is None  # c.z < r + znear
    a = passes.project_sphere_clip_space((0.0, 0.0, 10.0, 1.0), 0.01, 1.0, 1.0)
    assert a is not None and a[0] < 0 < a[2] and np.isclose(a[0], -a[2]) and np.isclose(a[1], -a[3])


def test_cull_info_to_gpu_matches_layout():
    cam = sc.default_camera()
    view = sc.mat4_cols(cam.view)
    proj = passes.Projection.Perspective(cam.fov, cam.z_near)
    # OcclusionCullInfo::None, forward.rs:266-284 without occlusion
    g0 = passes.cull_info_to_gpu(passes.CullInfo(view, cam.planes, proj))
    ref0 = sc.make_cull_info(cam.view, cam.planes)
    for f in ("view_matrix", "cull_planes", "cull_plane_count", "alpha_mode_flag", "occlusion_pass",
              "visibility_buffer", "meshlet_visibility_buffer", "depth_pyramid", "projection_type", "lod_base",
              "lod_step", "min_mesh_lod", "max_mesh_lod", "p00_or_width_recipx2", "z_near"):
        assert np.array_equal(g0[f], ref0[f]), f
    assert g0["visibility_buffer"] == L.NONE and g0["p00_or_width_recipx2"] == 0.0  # filled only for VisibilityWrite
    # VisibilityWrite, forward.rs:371-389
    ci2 = passes.CullInfo(view, cam.planes, proj, occlusion_kind=passes.VISIBILITY_WRITE, visibility_buffer=0x1000,
                          meshlet_visibility_buffer=0x2000, depth_pyramid=0x3000, depth_pyramid_size=(1024, 1024),
                          aspect_ratio=cam.aspect, lod_range=(1, 6), lod_target_pos_view_space=(1, 2, 3))
    g2 = passes.cull_info_to_gpu(ci2)
    assert g2["occlusion_pass"] == 2 and g2["visibility_buffer"] != L.NONE and g2["depth_pyramid"] != L.NONE
    assert np.isclose(g2["p00_or_width_recipx2"], cam.p00, rtol=1e-6) and np.isclose(g2["p11_or_height_recipx2"], cam.p11)
    assert g2["z_near"] == np.float32(cam.z_near) and (g2["min_mesh_lod"], g2["max_mesh_lod"]) == (1, 5)
    assert list(g2["lod_target_pos_view_space"]) == [1, 2, 3]
    # meshlet_visibility_buffer: Option::None -> u32::MAX (draw_gen.rs:143)
    g2n = passes.cull_info_to_gpu(passes.CullInfo(view, cam.planes, proj, occlusion_kind=passes.VISIBILITY_WRITE,
                                                  visibility_buffer=0x1000, depth_pyramid=0x3000))
    assert g2n["meshlet_visibility_buffer"] == L.NONE
    # orthographic VisibilityWrite: 2/width, 2/height, z_far (draw_gen.rs:186-198)
    go = passes.cull_info_to_gpu(passes.CullInfo(view, cam.planes, passes.Projection.Orthographic(25.0, 0.1, 80.0),
                                                 occlusion_kind=passes.VISIBILITY_WRITE, visibility_buffer=1,
                                                 depth_pyramid=1, aspect_ratio=2.0))
    assert go["projection_type"] == 1 and np.isclose(go["p00_or_width_recipx2"], 2.0 / 50.0)
    assert np.isclose(go["p11_or_height_recipx2"], 2.0 / 25.0) and go["z_far"] == np.float32(80.0)


def test_cluster_settings(oracle):
    s = passes.ClusterSettings(screen_resolution=(1920, 1080))
    d = passes.cluster_settings_derive(s, 0.01)
    assert list(d.tile_counts) == [240, 135] and list(d.cluster_counts) == [240, 135, 32]
    assert d.linear_cluster_count == 1_036_800 == d.linear_max_allocated_cluster_count and d.tile_px_size == 8
    zs, zb = oracle.cluster_grid_info(0.01, 200.0, 32)
    assert np.isclose(d.z_scale, zs, rtol=1e-5) and np.isclose(d.z_bias, zb, rtol=1e-5)
    s2 = passes.ClusterSettings(px_size_power=6, screen_resolution=(333, 65), z_slice_count=2)
    d2 = passes.cluster_settings_derive(s2, 0.1)
    assert list(d2.tile_counts) == [6, 2] and d2.linear_max_allocated_cluster_count == 6 * 2 * 4  # max(4, z) slices


def _quat_yxz(pitch, yaw):
    """glam Quat::from_euler(YXZ, pitch, yaw, 0) as (x, y, z, w): rotation about Y by `pitch`, then X by `yaw`."""
    cy, sy, cx, sx = np.cos(pitch / 2), np.sin(pitch / 2), np.cos(yaw / 2), np.sin(yaw / 2)
    return (float(cy * sx), float(sy * cx), float(-sy * sx), float(cy * cx))


@pytest.mark.parametrize("cascade", range(4))
def test_shadow_cascade_cull_info(cascade):
    """ShadowRenderer::render_cascaded_shadow (shadow_renderer.rs:466-706) restated in the host mirror: the CullInfo of a
    cascade is orthographic, carries the six planes of the light's box plus the camera planes that face the light
    (<= 11), keeps the cascade's sub-frustum inside, and snaps the light-space origin to shadow-map texels."""
    cam_pos, cam_q = (3.0, 2.0, -5.0), _quat_yxz(0.7, -0.2)
    direction = _quat_yxz(0.3, -1.0)  # a sun direction
    fov, near, aspect = float(np.float32(np.pi / 2)), 0.01, 16.0 / 9.0
    ci, lpm, world_size = passes.shadow_cascade(direction, cam_pos, cam_q, fov, near, aspect, cascade)
    n = ci.plane_count
    assert 6 <= n <= 11 and ci.projection.kind == 1 and ci.occlusion_kind == 0
    assert ci.alpha_mode_filter == L.ALPHA_OPAQUE | L.ALPHA_MASKED
    assert tuple(ci.lod_range) == ((0, 8) if cascade <= 1 else (2, 8))
    radius = ci.projection.half_width
    assert world_size == pytest.approx(2 * radius) and ci.projection.near_clip == pytest.approx(-radius - 80.0)
    planes = np.array(ci.view_space_cull_planes, dtype=np.float64)[:n]
    assert np.allclose(np.linalg.norm(planes[:, :3], axis=1), 1.0, atol=1e-5)
    assert np.all(planes[6:, 2] >= 0.0)  # only camera planes that face the light are kept (:637)
    # texel snapping: the light-space translation is a multiple of the texel size (:529-535)
    light = np.array(ci.view_matrix, dtype=np.float64).reshape(4, 4).T  # [row, col]
    texel = 2 * radius / 2048
    assert np.allclose(np.round(light[:3, 3] / texel), light[:3, 3] / texel, atol=2e-3)
    # the cascade's sub-frustum (camera space -> world -> light) is inside every culling plane and inside the box
    lam, far_clip = 0.80, 32.0
    split = lambda r: near * (far_clip / near) ** r * lam + (1 - lam) * (near + (far_clip - near) * r)
    zn, zf = split(cascade / 4), split((cascade + 1) / 4)
    th, tv = np.tan(fov / 2) * aspect, np.tan(fov / 2)
    corners = np.array([[sx * z * th, sy * z * tv, -z, 1.0] for z in (zn, zf) for sx in (-1, 1) for sy in (-1, 1)])
    x, y, z, w = cam_q
    rot = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                    [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                    [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    world = corners[:, :3] @ rot.T + np.array(cam_pos)
    ls = world @ light[:3, :3].T + light[:3, 3]
    assert np.all(np.abs(ls[:, :2]) <= radius * 1.001 + 2 * texel)
    d = ls @ planes[:, :3].T + planes[:, 3]
    # the near plane of the light box is documented as wrong in the reference ("only the near plane is wrong, but we
    # don't use that", :629-631): exclude the two z planes of the box, check the rest
    keep = [i for i in range(n) if i not in (4, 5)]
    assert np.all(d[:, keep] >= -radius * 1e-3 - 2 * texel)
    gpu = passes.cull_info_to_gpu(ci)
    assert gpu["projection_type"] == 1 and gpu["cull_plane_count"] == n


@pytest.mark.parametrize("W,H,tile", [(320, 180, 8), (333, 187, 16)])
def test_reference_cpu_twin_of_the_cluster_volume_contains_the_shader_box(oracle, W, H, tile):
    """cluster.rs:150-184 `compute_cluster_aabb` is the reference's own CPU restatement of the cluster volume (its
    debug draw), with the ANALYTIC near/far of the depth slice; light_culling.comp:62-90 builds the box from the
    MEASURED depth bounds of the slice, which lie inside the analytic ones.  So the oracle's box of every active
    cluster must lie inside the twin's box (up to rounding) — a check of the restatement against reference code that
    does not come from the shader text."""
    import test_gpu_parity as T

    zs, far = 32, 200.0
    push, depth, info, _ = T.cluster_inputs(oracle, 6, W, H, 0, tile, zs, far=far)
    masks, bounds = oracle.cluster_mark(push, depth)
    cx, cy, cz = (int(v) for v in push["cluster_count"])
    inv_proj = np.asarray(info["screen_to_view_matrix"], dtype=np.float32)
    z_near = float(push["z_near"])
    checked = 0
    rng = np.random.default_rng(1)
    tiles = rng.permutation(cx * cy)[:400]
    for t in tiles:
        for z in range(cz):
            if not (int(masks[t]) >> z) & 1:
                continue
            ci = int(t) + z * cx * cy
            omn, omx = oracle.cluster_aabb(info, bounds, ci)
            tmn, tmx = passes.compute_cluster_aabb(inv_proj, (W, H), tile, (cx, cy, cz), z_near, far,
                                                   (int(t) % cx, int(t) // cx, z))
            assert np.all(np.isfinite(omn)) and np.all(np.isfinite(omx))
            span = np.maximum(np.abs(tmn), np.abs(tmx)) + 1e-3
            tol = 2e-4 * span  # the slice edges are pow() in the twin and log2() on the device
            assert np.all(omn >= tmn - tol) and np.all(omx <= tmx + tol), (ci, omn, omx, tmn, tmx)
            checked += 1
    assert checked > 300


def test_shader_project_sphere_equals_the_reference_cpu_twin(oracle):
    """math::project_sphere_clip_space (src/math.rs:170-199) is the reference's CPU twin of project_sphere
    (entity_cull.comp:83-98): the same construction, stopping before the clip -> uv remap.  The host mirror restates
    the Rust (no contraction: rustc does not fuse), the oracle the COMPILED shader, where spirv-opt fused
    dot(c, c) - r * r and the uv remap (oracle/orbit_oracle.c project_sphere): the two agree to a few ulps — bit for
    bit in most cases — and never differ in structure."""
    rng = np.random.default_rng(8)
    n_some = n_same = 0
    for _ in range(3000):
        r = np.float32(rng.uniform(0.05, 5.0))
        c = np.array([rng.uniform(-40, 40), rng.uniform(-20, 20), rng.uniform(0.0, 80.0)], dtype=np.float32)
        p00, p11 = np.float32(rng.uniform(0.5, 2.0)), np.float32(rng.uniform(0.5, 3.0))
        znear = np.float32(0.01)
        twin = passes.project_sphere_clip_space((c[0], c[1], c[2], r), znear, p00, p11)
        if twin is None:
            assert c[2] < r + znear  # the same near-plane gate as entity_cull.comp:161-163 (`cullable`)
            continue
        n_some += 1
        twin = np.asarray(twin, dtype=np.float32)
        h, mh = np.float32(0.5), np.float32(-0.5)
        uv = np.array([twin[0] * h + h, twin[3] * mh + h, twin[2] * h + h, twin[1] * mh + h], dtype=np.float32)
        got = oracle.project_sphere(np.array([c[0], c[1], c[2], r], dtype=np.float32), p00, p11)
        n_same += int(np.array_equal(got.view(np.uint32), uv.view(np.uint32)))
        # the fused square root argument moves the bounds by an ulp or two of the clip-space value (|aabb| can be
        # far above 1 for spheres near the eye, so the tolerance is relative to the clip-space magnitude)
        tol = np.float32(4.0) * np.spacing(np.maximum(np.abs(uv), np.float32(1.0)))
        assert np.all(np.abs(got.astype(np.float64) - uv.astype(np.float64)) <= tol.astype(np.float64) * 64), (c, r, got, uv)
    assert n_some > 2000 and n_same > n_some // 4
