"""CPU suite: pins the C oracle against an independent numpy restatement,
structural properties and the committed golden fixtures.

The reference holds no golden vectors for this path, so these tests are the
oracle's pin; the GPU parity tests (test_gpu_parity.py) then compare the HIP
engine with the oracle.
"""
import numpy as np
import pytest

import np_restatement as npr
import scenes as sc
from orbit_amd import layouts as L


def popcount(words):
    return int(sum(bin(int(w)).count("1") for w in np.asarray(words).reshape(-1)))


# ----------------------------------------------------------------------------- log2
def test_log2_spec(oracle):
    xs = np.concatenate([np.exp2(np.arange(-126, 128, dtype=np.float64)).astype(np.float32),
                         np.random.default_rng(0).uniform(1e-6, 1e6, 5000).astype(np.float32),
                         np.array([1e-40, 3e-39, 0.0, np.inf], dtype=np.float32)])
    got = np.array([oracle.log2f(float(x)) for x in xs], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), npr.log2c(xs).view(np.uint32))  # two restatements agree bitwise
    finite = np.isfinite(got) & (xs > 0)
    ref = np.log2(xs[finite].astype(np.float64))
    assert np.all(np.abs(got[finite].astype(np.float64) - ref) <= 1.5e-6 + np.abs(ref) * 2.0 ** -23)
    p2 = np.exp2(np.arange(-126, 128, dtype=np.float64)).astype(np.float32)
    assert [oracle.log2f(float(x)) for x in p2] == list(range(-126, 128))  # exact at powers of two
    assert oracle.log2f(0.0) == -np.inf and np.isnan(oracle.log2f(-1.0)) and oracle.log2f(np.inf) == np.inf


def _round_f32(fr):
    """Exact rational -> nearest float32 (ties to even), normal range."""
    from fractions import Fraction
    if fr == 0:
        return 0.0
    sign, a = (-1 if fr < 0 else 1), abs(fr)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    while Fraction(2) ** e > a:
        e -= 1
    while Fraction(2) ** (e + 1) <= a:
        e += 1
    q = a / Fraction(2) ** (e - 23)
    n = q.numerator // q.denominator
    rem = q - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
        n += 1
    return sign * float(Fraction(n) * Fraction(2) ** (e - 23))


def test_snorm8_is_the_compiled_shaders_reciprocal_multiply():
    """int(i8) / 127.0 in the reference's BINARIES is a multiplication by the rounded reciprocal 0x3C010204
    (meshlet_cull.comp.spv, found by executing it: tests/golden/make_spirv_vectors.py); oracle, device and the numpy
    restatement follow the binary.  The product differs from the IEEE quotient for 16 of the 256 inputs — by one ulp."""
    from fractions import Fraction
    import np_restatement as npr

    rcp = float(npr.RCP127)
    assert np.array([rcp], np.float32).view(np.uint32)[0] == 0x3C010204
    assert _round_f32(Fraction(1, 127)) == rcp  # the reciprocal itself is correctly rounded
    differ = 0
    for i in range(-128, 128):
        prod = _round_f32(Fraction(i) * Fraction(rcp))
        assert prod == float(np.float32(i) * npr.RCP127), i
        quot = _round_f32(Fraction(i, 127))
        differ += prod != quot
        assert abs(Fraction(prod) - Fraction(quot)) <= Fraction(2) ** -23 * abs(Fraction(quot)) + 0, i
    assert differ == 16


def test_fma32_is_correctly_rounded():
    """np_restatement.fma32 (binary64 product + sum with the double-rounding repair) against exact rationals, on
    operands built to land on and around half-way points."""
    from fractions import Fraction
    import np_restatement as npr

    rng = np.random.default_rng(11)
    a = rng.standard_normal(4000).astype(np.float32)
    b = rng.standard_normal(4000).astype(np.float32)
    c = rng.standard_normal(4000).astype(np.float32) * np.float32(1e-3)
    # half-way cases: c = -(a*b rounded to 25 bits' last half) constructions
    a[:1000] = (rng.integers(1, 2 ** 12, 1000) * 2 + 1).astype(np.float32)
    b[:1000] = (rng.integers(1, 2 ** 12, 1000) * 2 + 1).astype(np.float32)
    c[:1000] = (rng.integers(-4, 5, 1000).astype(np.float32) * np.float32(2.0) ** rng.integers(-30, -20, 1000).astype(np.float32))
    got = npr.fma32(a, b, c)
    for k in range(len(a)):
        want = _round_f32(Fraction(float(a[k])) * Fraction(float(b[k])) + Fraction(float(c[k])))
        assert float(got[k]) == want, (k, a[k], b[k], c[k])


# ----------------------------------------------------------------------------- host derivations
def test_frustum_planes(oracle):
    cam = sc.default_camera()
    got = oracle.frustum_planes_from_matrix(sc.mat4_cols(cam.proj))
    assert np.array_equal(got[:5].view(np.uint32), sc.frustum_planes(cam.proj, 5).view(np.uint32))
    # infinite reverse-Z: plane 4 is (0,0,-1,near), a plane behind the eye
    assert np.allclose(got[4], [0, 0, -1, cam.z_near])


@pytest.mark.parametrize("size,expect", [((1920, 1080), (1024, 1024, 11)), ((2560, 1080), (2048, 1024, 12)),
                                         ((640, 360), (512, 256, 10)), ((1, 1), (1, 1, 1)), ((33, 2), (32, 1, 6))])
def test_pyramid_desc(oracle, size, expect):
    d = oracle.depth_pyramid_desc(*size)
    assert (d.width, d.height, d.mip_levels) == expect
    assert d.total_texels == sum(max(d.width >> k, 1) * max(d.height >> k, 1) for k in range(d.mip_levels))


# ----------------------------------------------------------------------------- depth_reduce
@pytest.mark.parametrize("size", [(96, 54), (130, 40), (64, 64), (17, 9), (200, 3)])
def test_depth_reduce_vs_numpy(oracle, size):
    w, h = size
    cam = sc.default_camera(aspect=w / h)
    depth = sc.make_depth(5, w, h, cam, n_occluders=12)
    pyr, d = oracle.depth_reduce(depth, w, h)
    ref, _ = npr.depth_reduce(depth, w, h)
    assert np.array_equal(pyr.view(np.uint32), ref.view(np.uint32))
    # structure: levels >= 1 are exact 2x2 block minima; the top level is the global min of level 0
    for k in range(1, d.mip_levels):
        sw_, sh_ = d.mip_width[k - 1], d.mip_height[k - 1]
        src = pyr[d.mip_offset[k - 1]:d.mip_offset[k - 1] + sw_ * sh_].reshape(sh_, sw_)
        dw, dh = d.mip_width[k], d.mip_height[k]
        dst = pyr[d.mip_offset[k]:d.mip_offset[k] + dw * dh].reshape(dh, dw)
        for y in range(dh):
            for x in range(dw):
                blk = src[min(2 * y, sh_ - 1):min(2 * y + 1, sh_ - 1) + 1, min(2 * x, sw_ - 1):min(2 * x + 1, sw_ - 1) + 1]
                assert dst[y, x] == blk.min()
    lvl0 = pyr[:d.mip_width[0] * d.mip_height[0]]
    assert pyr[-1] == lvl0.min()
    assert lvl0.min() >= depth.min()  # min-reduction never invents a farther depth


# ----------------------------------------------------------------------------- cull: oracle vs numpy restatement
def _run_both(oracle, scene, ci, evis=None, mvis=None, pyr=None, psize=(0, 0)):
    cap = scene.max_dispatches() + 8
    disp, evis_o, dropped = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count,
                                               scene.mesh_infos, scene.entities, cap, evis, pyr, psize)
    assert dropped == 0
    hdr, recs = L.dispatch_buffer_records(disp)
    vis_n, should_n, recs_n, evis_n = npr.entity_cull(ci, scene.entity_draws, scene.entity_draw_count,
                                                      scene.entity_draw_count, scene.mesh_infos, scene.entities, evis,
                                                      pyr, psize)
    assert list(hdr) == [len(recs_n), 1, 1]
    assert np.array_equal(recs.view(np.uint32), recs_n.view(np.uint32))
    if evis_n is not None:
        assert np.array_equal(evis_o, evis_n)
    draw_cap = int(recs["meshlet_count"].sum()) + 8
    draw, mvis_o, dropped = oracle.meshlet_cull(ci, disp, scene.meshlets, draw_cap, scene.entities, scene.materials,
                                                mvis, pyr, psize)
    assert dropped == 0
    n, cmds = L.draw_buffer_commands(draw)
    cmds_n, mvis_n = npr.meshlet_cull(ci, recs_n, scene.meshlets, scene.entities, scene.materials, mvis, pyr, psize)
    assert n == len(cmds_n)
    assert np.array_equal(cmds.view(np.uint32), cmds_n.view(np.uint32))
    if mvis_n is not None:
        assert np.array_equal(mvis_o, mvis_n)
    return recs, cmds, evis_o, mvis_o


@pytest.mark.parametrize("seed,lods,shuffle", [(1, 1, True), (2, 3, True), (3, 1, False)])
def test_pass0_frustum_cone(oracle, seed, lods, shuffle):
    scene = sc.make_scene(seed, 300, n_meshes=90, lods=lods, shuffle=shuffle)
    cam = sc.default_camera(rot=(0.8, 0.6))
    ci = sc.make_cull_info(cam.view, cam.planes, lod_base=8.0)
    recs, cmds, _, _ = _run_both(oracle, scene, ci)
    assert 0 < len(cmds) < int(recs["meshlet_count"].sum())
    if lods > 1:  # distant entities pick coarser LODs
        offs = scene.mesh_infos["mesh_lods"][:, 0, 0]
        assert np.any(~np.isin(recs["meshlet_offset"] - recs["meshlet_offset"] % 1, offs))


def test_pass0_no_planes_all_alpha(oracle):
    scene = sc.make_scene(4, 64, meshlets_per_mesh=(1, 70))
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, np.zeros((0, 4), np.float32), alpha_mode_flag=L.ALPHA_ALL)
    recs, cmds, _, _ = _run_both(oracle, scene, ci)
    assert len(recs) == scene.max_dispatches()  # frustum_culling = false -> every entity emits


def test_two_pass_sequence(oracle):
    """pass 1 (VisibilityRead) -> depth_reduce -> pass 2 (VisibilityWrite), forward.rs:266-403."""
    W, H = 320, 180
    scene = sc.make_scene(11, 400, n_meshes=100, extent=(40.0, 10.0, 40.0))
    cam = sc.default_camera()
    rng = np.random.default_rng(5)
    evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    ci1 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=1)
    recs1, cmds1, _, _ = _run_both(oracle, scene, ci1, evis, mvis)
    depth = sc.make_depth(9, W, H, cam)
    pyr, d = oracle.depth_reduce(depth, W, H)
    ci2 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
    recs2, cmds2, evis2, mvis2 = _run_both(oracle, scene, ci2, evis, mvis, pyr, (d.width, d.height))
    # pass 2 only draws meshlets that were NOT visible last frame (meshlet_cull.comp:210-213)
    drawn1 = set(zip(cmds1["cmd_first_instance"].tolist(), cmds1["meshlet_index"].tolist()))
    drawn2 = set(zip(cmds2["cmd_first_instance"].tolist(), cmds2["meshlet_index"].tolist()))
    assert drawn1.isdisjoint(drawn2)
    assert len(drawn2) > 0 and popcount(evis2) > 0 and popcount(mvis2) >= len(cmds2)
    # occlusion actually culls something relative to frustum+cone only
    ci0 = sc.make_cull_info(cam.view, cam.planes, alpha_mode_flag=L.ALPHA_ALL)
    _, cmds0, _, _ = _run_both(oracle, scene, ci0)
    zero_e, zero_m = np.zeros_like(evis), np.zeros_like(mvis)
    _, cmds2z, _, mvis2z = _run_both(oracle, scene, ci2, zero_e, zero_m, pyr, (d.width, d.height))
    assert popcount(mvis2z) == len(cmds2z) < len(cmds0)  # frame 0: everything visible is newly visible


def test_pass2_without_meshlet_occlusion(oracle):
    W, H = 256, 128
    scene = sc.make_scene(12, 150)
    cam = sc.default_camera(aspect=2.0)
    depth = sc.make_depth(2, W, H, cam)
    pyr, d = oracle.depth_reduce(depth, W, H)
    evis = np.zeros((scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    ci2 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, meshlet_visibility=False, p00=cam.p00,
                            p11=cam.p11, z_near=cam.z_near)
    _run_both(oracle, scene, ci2, evis, None, pyr, (d.width, d.height))


def test_orthographic_shadow_cascade(oracle):
    """The shadow renderer's CullInfo shape: ortho projection, 6 + k planes, no occlusion
    (shadow_renderer.rs:622-712) and the ortho HiZ branch (entity_cull.comp:165-175)."""
    scene = sc.make_scene(13, 250, extent=(20.0, 6.0, 20.0))
    proj = sc.orthographic_rh(-25, 25, -25, 25, 0.1, 80.0)
    planes = np.concatenate([sc.frustum_planes(proj, 6), sc.default_camera().planes[:3]])
    view = sc.translation(0.0, 0.0, -40.0)
    ci = sc.make_cull_info(view, planes, projection_type=1, lod_target=(1.0, 2.0, 3.0), lod_range=(0, 8))
    assert int(ci["cull_plane_count"]) == 9
    _run_both(oracle, scene, ci)
    W = H = 128
    depth = np.full((H, W), 0.5, dtype=np.float32)
    depth[:, : W // 2] = 0.9
    pyr, d = oracle.depth_reduce(depth, W, H)
    evis = np.zeros((scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = np.zeros(scene.vis_words, dtype=np.uint32)
    ci2 = sc.make_cull_info(view, planes, occlusion_pass=2, projection_type=1, p00=2.0 / 50.0, p11=2.0 / 50.0,
                            z_near=0.1, z_far=80.0)
    _run_both(oracle, scene, ci2, evis, mvis, pyr, (d.width, d.height))


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_multithreaded_oracle_equals_sequential(oracle, threads):
    """orbit_oracle_mt.c (cpu_baseline on all cores) composes the sequential oracle over shards: same bytes, all
    passes, including capacity cuts."""
    scene = sc.make_scene(77, 700, n_meshes=90, lods=2, meshlets_per_mesh=(1, 80))
    cam = sc.default_camera(rot=(0.2, 0.5))
    rng = np.random.default_rng(77)
    evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    W, H = 160, 90
    pyr, d = oracle.depth_reduce(sc.make_depth(7, W, H, cam), W, H)
    ps = (d.width, d.height)
    n = scene.entity_draw_count
    args = (scene.entity_draw_buffer(), n, scene.mesh_infos, scene.entities)
    for p in (0, 1, 2):
        ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=p, p00=cam.p00, p11=cam.p11, z_near=cam.z_near,
                               lod_base=8.0)
        for cap_d, cap_c in ((scene.max_dispatches() + 8, scene.lod0_meshlets + 8), (40, 100)):
            disp, ev, dd = oracle.entity_cull(ci, *args, cap_d, evis if p else None, pyr if p == 2 else None, ps)
            draw, mv, dc = oracle.meshlet_cull(ci, disp, scene.meshlets, cap_c, scene.entities, scene.materials,
                                               mvis if p else None, pyr if p == 2 else None, ps)
            mdisp, mdraw, mev, mmv, dropped = oracle.cull_frame_mt(
                ci, *args, scene.meshlets, scene.materials, cap_d, cap_c, threads, evis if p else None,
                mvis if p else None, pyr if p == 2 else None, ps)
            assert np.array_equal(mdisp, disp) and np.array_equal(mdraw, draw), (p, cap_d)
            assert dropped == (dd, dc)
            if p:
                assert np.array_equal(mev, ev) and np.array_equal(mmv, mv)


def test_native_flags_build_of_the_oracle_is_byte_identical(oracle):
    """bench.py's cpu_baseline runs the oracle built -O3 -march=native -ffp-contract=off: without
    contraction the optimiser may vectorise but every operation stays a correctly rounded IEEE one — same bytes as
    the -O2 build the parity tests use, all passes."""
    scene = sc.make_scene(78, 500, n_meshes=70, lods=2, meshlets_per_mesh=(1, 80))
    cam = sc.default_camera(rot=(0.2, 0.5))
    rng = np.random.default_rng(78)
    evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    W, H = 160, 90
    pyr, d = oracle.depth_reduce(sc.make_depth(8, W, H, cam), W, H)
    ps = (d.width, d.height)
    args = (scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos, scene.entities, scene.meshlets,
            scene.materials, scene.max_dispatches() + 8, scene.lod0_meshlets + 8, 4)
    for p in (0, 1, 2):
        ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=p, p00=cam.p00, p11=cam.p11, z_near=cam.z_near,
                               lod_base=8.0)
        kw = dict(visibility=evis if p else None, meshlet_visibility=mvis if p else None,
                  pyramid=pyr if p == 2 else None, pyramid_size=ps)
        a = oracle.cull_frame_mt(ci, *args, **kw)
        b = oracle.cull_frame_mt(ci, *args, native=True, **kw)
        for x, y in zip(a[:4], b[:4]):
            assert (x is None and y is None) or np.array_equal(x, y), p
        assert a[4] == b[4]


def test_task_path_agrees_with_the_draw_command_path(oracle):
    """forward_depth_prepass.task and meshlet_cull.comp take the same per-meshlet decision: the task records'
    (record, lane) survivors are exactly the draw commands, and the pass-2 visibility words differ only in the
    lanes past a record's meshlet_count (the task shader leaves `visible = true` there)."""
    scene = sc.make_scene(91, 300, n_meshes=50, lods=2, meshlets_per_mesh=(1, 75))
    cam = sc.default_camera(rot=(0.1, -0.7))
    rng = np.random.default_rng(91)
    evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    W, H = 160, 90
    pyr, d = oracle.depth_reduce(sc.make_depth(3, W, H, cam), W, H)
    ps = (d.width, d.height)
    for p in (0, 1, 2):
        ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=p, p00=cam.p00, p11=cam.p11, z_near=cam.z_near,
                               lod_base=8.0)
        disp, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                        scene.entities, scene.max_dispatches() + 8, evis if p else None,
                                        pyr if p == 2 else None, ps)
        draw, mv, _ = oracle.meshlet_cull(ci, disp, scene.meshlets, scene.lod0_meshlets + 8, scene.entities,
                                          scene.materials, mvis if p else None, pyr if p == 2 else None, ps)
        recs_t, mv_t = oracle.meshlet_task_cull(ci, disp, scene.meshlets, scene.entities, scene.materials,
                                                mvis if p else None, pyr if p == 2 else None, ps)
        _, recs = L.dispatch_buffer_records(disp)
        n, cmds = L.draw_buffer_commands(draw)
        assert len(recs_t) == len(recs) and int(recs_t["task_mesh_count"].sum()) == n > 0
        k = 0
        for r, t in zip(recs, recs_t):
            assert t["entity_index"] == r["entity_index"] and t["meshlet_offset"] == r["meshlet_offset"]
            c = int(t["task_mesh_count"])
            lanes = t["meshlet_indices"][:c].astype(np.int64)
            assert np.all(np.diff(lanes) > 0) and np.all(lanes < r["meshlet_count"]) and not t["meshlet_indices"][c:].any()
            assert np.array_equal(cmds["meshlet_index"][k:k + c], r["meshlet_offset"] + lanes)
            assert np.all(cmds["cmd_first_instance"][k:k + c] == r["entity_index"])
            k += c
        assert k == n
        if p == 2:
            inactive = np.zeros_like(mv)
            for r in recs:
                if r["meshlet_count"] < 32:
                    inactive[r["visibility_offset"]] = (0xFFFFFFFF << int(r["meshlet_count"])) & 0xFFFFFFFF
            touched = np.zeros(len(mv), dtype=bool)
            touched[recs["visibility_offset"]] = True
            assert np.array_equal(mv_t[touched], (mv | inactive)[touched]) and np.array_equal(mv_t[~touched], mv[~touched])


def test_cluster_lists_serve_the_fragment_shader_walk(oracle):
    """End-to-end property of the three cluster stages, from the consumer's side (forward.frag:352-363): a fragment
    at pixel (x, y) with depth d looks up tile = xy / tile_px, slice = linear_z_to_depth_slice(z_near / d), reads
    (offset, count) from the RG32UI image and walks light_indices[offset : offset + count].  Every point light whose
    sphere contains the fragment's view-space position must be in that walk (culling is conservative), lists are
    ascending and duplicate-free, non-point lights are in every active cluster."""
    W, H, tile, cz, far = 320, 180, 8, 32, 200.0
    cam = sc.default_camera(aspect=W / H)
    depth = sc.make_depth(8, W, H, cam, n_occluders=40)
    cx, cy = -(-W // tile), -(-H // tile)
    zs, zb = oracle.cluster_grid_info(cam.z_near, far, cz)
    push = np.zeros((), dtype=L.MARK_ACTIVE_PUSH)
    push["cluster_count"], push["tile_size_px"], push["screen_size"] = (cx, cy, cz), tile, (W, H)
    push["z_near"], push["z_far"], push["z_scale"], push["z_bias"] = cam.z_near, far, zs, zb
    push["depth_buffer_sample_count"] = 1
    info = np.zeros((), dtype=L.CLUSTER_CULL_INFO)
    info["world_to_view_matrix"] = sc.mat4_cols(cam.view)
    inv_proj = np.linalg.inv(cam.proj.astype(np.float64))
    info["screen_to_view_matrix"] = sc.mat4_cols(inv_proj.astype(np.float32))
    info["cluster_count"], info["tile_size_px"], info["screen_size"] = (cx, cy, cz), tile, (W, H)
    n_lights = 600
    info["z_near"], info["z_far"], info["global_light_count"] = cam.z_near, far, n_lights
    lights = sc.make_lights(8, n_lights, extent=(60.0, 15.0, 60.0), point_fraction=0.97)
    masks, bounds = oracle.cluster_mark(push, depth)
    total = cx * cy * cz
    unique, _ = oracle.cluster_compact((cx, cy, cz), masks, total)
    n_active = int(unique[12:16].view(np.uint32)[0])
    lidx, img, dropped = oracle.cluster_assign(info, unique, bounds, lights, n_active * 256 + 8, total)
    assert dropped == 0 and n_active > 100
    indices = lidx[4:].view(np.uint32)
    view = cam.view.astype(np.float64)
    lpos = (view[:3, :3] @ lights["position"].astype(np.float64).T).T + view[:3, 3]
    is_point = lights["light_type"] == L.LIGHT_TYPE_POINT
    rng = np.random.default_rng(8)
    checked = inside = 0
    for _ in range(4000):
        x, y = int(rng.integers(0, W)), int(rng.integers(0, H))
        d = float(depth[y, x])
        if d <= 0.0:
            continue  # sky: no fragment
        z = np.float32(cam.z_near) / np.float32(d)
        sl = int(np.uint32(max(np.float32(oracle.log2f(float(z))) * np.float32(zs) + np.float32(zb), 0)))  # cluster_common.glsl:18-20
        if sl >= cz:
            continue  # beyond the cluster far plane: imageLoad outside the image returns 0 lights (and mark skips it)
        assert (int(masks[(y // tile) * cx + x // tile]) >> sl) & 1  # the fragment's cluster is active
        off, cnt = (int(v) for v in img[(x // tile) + (y // tile) * cx + sl * cx * cy])
        walk = indices[off:off + min(cnt, 256)]
        assert np.all(np.diff(walk.astype(np.int64)) > 0)  # ascending, no duplicates
        assert set(np.flatnonzero(~is_point)).issubset(set(walk.tolist()))
        # fragment position in view space: unproject the pixel centre at NDC z = depth (reverse-Z, infinite far)
        ndc = np.array([(x + 0.5) / W * 2.0 - 1.0, 1.0 - (y + 0.5) / H * 2.0, d, 1.0])
        pv = inv_proj @ ndc
        pv = pv[:3] / pv[3]
        dist = np.linalg.norm(lpos - pv, axis=1)
        must = np.flatnonzero(is_point & (dist < 0.98 * lights["outer_radius"]))
        inside += len(must)
        assert set(must.tolist()).issubset(set(walk.tolist())), (x, y, sl, must, walk)
        checked += 1
    assert checked > 1500 and inside > 50  # the property was exercised, not vacuous


def test_config1_sphere_frustum(oracle):
    """BASELINE.json configs[0]: 8k spheres, frustum only (plumbing, CPU)."""
    n = 8192
    i = np.arange(n)
    spheres = np.stack([sc.rnd_range(69, 1, i, -128, 128), sc.rnd_range(69, 2, i, -128, 128),
                        sc.rnd_range(69, 3, i, -128, 128), sc.rnd_range(69, 4, i, 0.25, 4.0)], axis=1)
    cam = sc.default_camera()
    bits, cnt = oracle.sphere_frustum_cull(sc.mat4_cols(cam.view), cam.planes, spheres)
    c = spheres[:, :3] + cam.view[:3, 3]
    keep = np.ones(n, dtype=bool)
    for pl in cam.planes:
        d = ((pl[0] * c[:, 0] + pl[1] * c[:, 1]).astype(np.float32) + pl[2] * c[:, 2]).astype(np.float32) + pl[3]
        keep &= d > -spheres[:, 3]
    assert cnt == int(keep.sum()) == popcount(bits)
    assert 0.1 * n < cnt < 0.4 * n  # 90-degree frustum in a cube: roughly a quarter survives
    assert np.array_equal(np.unpackbits(bits.view(np.uint8), bitorder="little")[:n].astype(bool), keep)


def test_capacity_overflow_is_reported(oracle):
    scene = sc.make_scene(14, 50)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, np.zeros((0, 4), np.float32))
    disp, _, dropped = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                          scene.entities, 10)
    hdr, recs = L.dispatch_buffer_records(disp)
    assert hdr[0] == 10 and dropped == scene.max_dispatches() - 10
