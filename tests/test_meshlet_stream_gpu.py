"""GPU tests of the derived meshlet streams' own surface (include/orbit_abi.h, "Derived meshlet streams").  That the
stream path computes the same bits as the Meshlet-buffer path is covered by tests/test_gpu_parity.py, whose whole suite
runs a second time through tests/stream_engine.py; here: ranges under global indices, partial updates, stale data,
which calls take the path, argument errors."""
import ctypes as C

import numpy as np
import pytest

import scenes as sc
from orbit_amd import _lib, layouts as L
from test_gpu_parity import GpuScene, assert_same, dev, host, run_gpu, run_oracle, torch_mod  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture()
def engine(torch_mod):
    from orbit_amd.engine import Engine

    e = Engine(0, max_entities=50_000, max_dispatches=400_000, max_draws=1_000_000)
    yield e
    e.close()


def _scene(seed=31, n=900):
    scene = sc.make_scene(seed, n, meshlets_per_mesh=(3, 60), lods=2)
    cam = sc.default_camera(rot=(0.1, 0.3))
    return scene, sc.make_cull_info(cam.view, cam.planes)


def test_bound_stream_serves_passes_0_and_2_and_is_bit_identical(torch_mod, engine, oracle):
    torch = torch_mod
    scene, ci = _scene()
    gs = GpuScene(torch, scene)
    ref = run_oracle(oracle, scene, ci)
    plain = run_gpu(torch, engine, gs, ci)
    assert engine.meshlet_stream_culls() == 0
    ms = engine.meshlet_stream(gs.meshlets, 0, len(scene.meshlets))
    engine.bind_meshlet_stream(ms)
    got = run_gpu(torch, engine, gs, ci)
    assert engine.meshlet_stream_culls() == 1
    assert_same(got, ref)
    assert np.array_equal(got[1], plain[1])
    # LDS alpha table instead of the material gather
    got = run_gpu(torch, engine, gs, ci, material_count=len(scene.materials))
    assert engine.meshlet_stream_culls() == 2
    assert_same(got, ref)
    # another meshlet buffer (a copy at another address): the plain path, silently, same result
    other = GpuScene(torch, scene)
    got = run_gpu(torch, engine, other, ci)
    assert engine.meshlet_stream_culls() == 2
    assert_same(got, ref)
    engine.bind_meshlet_stream(None)
    run_gpu(torch, engine, gs, ci)
    assert engine.meshlet_stream_culls() == 2
    ms.close()


def test_stale_stream_culls_stale_data_until_updated(torch_mod, engine, oracle):
    """The contract of `update`: the stream is a copy.  Meshlets rewritten in place are culled from the old bytes
    until orbit_meshlet_stream_update re-derives them (a partial range suffices)."""
    torch = torch_mod
    scene, ci = _scene(seed=37, n=400)
    gs = GpuScene(torch, scene)
    ms = engine.meshlet_stream(gs.meshlets, 0, len(scene.meshlets))
    engine.bind_meshlet_stream(ms)
    before = run_gpu(torch, engine, gs, ci)
    assert_same(before, run_oracle(oracle, scene, ci))
    # move the first third of the meshlets far behind the camera, in place (same device address)
    k = len(scene.meshlets) // 3
    scene.meshlets["bounding_sphere"][:k, :3] += np.float32(1.0e6)
    gs.meshlets.copy_(dev(torch, scene.meshlets))
    stale = run_gpu(torch, engine, gs, ci)
    assert np.array_equal(stale[1], before[1]), "a stale stream must keep answering from its copy"
    ms.update(gs.meshlets, 0, k)  # only the rewritten range
    fresh = run_gpu(torch, engine, gs, ci)
    ref = run_oracle(oracle, scene, ci)
    assert_same(fresh, ref)
    assert L.draw_buffer_commands(fresh[1])[0] < L.draw_buffer_commands(before[1])[0]
    ms.close()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_streams_under_global_indices(torch_mod, engine, oracle, world):
    """A rank of the sharded engine holds only its meshlet shard, passed under global indices through a shifted base
    pointer; its stream covers [first_meshlet, first_meshlet + count) of that virtual buffer."""
    from orbit_amd.dist import shard_ranges

    torch = torch_mod
    scene = sc.make_scene(23, 700, meshlets_per_mesh=(8, 40), shuffle=False)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, cam.planes)
    gs = GpuScene(torch, scene)
    on, ocmds = L.draw_buffer_commands(run_oracle(oracle, scene, ci)[1])
    lod0 = scene.mesh_infos["mesh_lods"][:, 0]
    parts, served = [], 0
    for b, e in shard_ranges(scene.entity_draw_count, world):
        m0 = int(lod0[b, 0])
        m1 = int(lod0[e - 1, 0] + lod0[e - 1, 1])
        shard = dev(torch, scene.meshlets[m0:m1])
        base = shard.data_ptr() - m0 * 32
        ms = engine.meshlet_stream(base, m0, m1 - m0)
        engine.bind_meshlet_stream(ms)
        cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
        disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
        draw = torch.zeros(L.DRAW_HEADER + 28 * cap_c, dtype=torch.uint8, device="cuda")
        engine.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, e - b, cap_d, draw_first=b)
        engine.meshlet_cull(ci, disp, base, draw, gs.entities, gs.materials, cap_d, cap_c)
        torch.cuda.synchronize()
        served += 1
        assert engine.meshlet_stream_culls() == served
        parts.append(L.draw_buffer_commands(host(draw))[1])
        engine.bind_meshlet_stream(None)
        ms.close()
    got = np.concatenate(parts)
    assert len(got) == on and np.array_equal(got.view(np.uint32), ocmds.view(np.uint32))


def test_visible_list_and_expansion_from_the_stream(torch_mod, engine, oracle):
    torch = torch_mod
    scene, ci = _scene(seed=41, n=600)
    gs = GpuScene(torch, scene)
    on, ocmds = L.draw_buffer_commands(run_oracle(oracle, scene, ci)[1])
    ms = engine.meshlet_stream(gs.meshlets, 0, len(scene.meshlets))
    engine.bind_meshlet_stream(ms)
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
    vis = torch.full((L.VISIBLE_HEADER + 12 * cap_d + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    draw = torch.full((L.DRAW_HEADER + 28 * cap_c + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    engine.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, scene.entity_draw_count, cap_d)
    engine.meshlet_cull_visible_records(ci, disp, gs.meshlets, vis, gs.entities, gs.materials, cap_d, cap_d)
    engine.expand_visible_records(vis, gs.meshlets, draw, cap_c)
    torch.cuda.synchronize()
    engine.status()
    assert engine.meshlet_stream_culls() == 2  # the cull and the expansion
    n, cmds = L.draw_buffer_commands(host(draw))
    assert n == on and np.array_equal(cmds.view(np.uint32), ocmds.view(np.uint32))
    assert bool((host(draw)[4 + 28 * on:] == 0xCD).all())
    ms.close()


def test_pass1_keeps_reading_the_meshlet_buffer(torch_mod, engine, oracle):
    torch = torch_mod
    scene, _ = _scene(seed=43, n=500)
    gs = GpuScene(torch, scene)
    ms = engine.meshlet_stream(gs.meshlets, 0, len(scene.meshlets))
    engine.bind_meshlet_stream(ms)
    cam = sc.default_camera(rot=(0.1, 0.3))
    evis = np.full((scene.entity_draw_count + 31) // 32, 0xFFFFFFFF, np.uint32)
    mvis = np.full(scene.vis_words, 0x55555555, np.uint32)
    ci1 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=1)
    got = run_gpu(torch, engine, gs, ci1, evis=evis, mvis=mvis)
    assert engine.meshlet_stream_culls() == 0
    assert_same(got, run_oracle(oracle, scene, ci1, evis=evis, mvis=mvis))
    ms.close()


def test_cull_views_from_the_stream(torch_mod, oracle):
    """orbit_cull_views with a bound stream: a late pass (occlusion_pass 2 with meshlet occlusion: HiZ candidates in the
    LDS ring), a single-pass perspective view and a pass-1 view side by side — the first two evaluate from the
    streams, pass 1 from the Meshlet buffer; every view equals the oracle."""
    from orbit_amd.engine import Engine

    torch = torch_mod
    W, H = 320, 180
    scene = sc.make_scene(11, 1800, n_meshes=250, extent=(60.0, 12.0, 60.0))
    cam = sc.default_camera()
    gs = GpuScene(torch, scene)
    rng = np.random.default_rng(7)
    evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    depth = sc.make_depth(9, W, H, cam)
    opyr, d = oracle.depth_reduce(depth, W, H)
    ps = (d.width, d.height)
    cis = [sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, p00=cam.p00, p11=cam.p11, z_near=cam.z_near),
           sc.make_cull_info(cam.view, cam.planes),
           sc.make_cull_info(cam.view, cam.planes, occlusion_pass=1)]
    # (cull_path = 1: the launch chain — views of this size would otherwise take the one-launch cull, which reads the
    # Meshlet buffer whatever stream is bound)
    eng = Engine(0, max_entities=8192, max_dispatches=40_000, max_draws=200_000, max_views=3, cull_path=1)
    eng.bind_meshlet_stream(eng.meshlet_stream(gs.meshlets, 0, len(scene.meshlets)))
    disp_cap, draw_cap = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    pyr = dev(torch, opyr)
    views, bufs = [], []
    for ci in cis:
        occ = int(ci["occlusion_pass"]) != 0
        disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device="cuda")
        draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device="cuda")
        e_d, m_d = (dev(torch, evis), dev(torch, mvis)) if occ else (None, None)
        bufs.append((disp, draw, e_d, m_d))
        v = dict(cull_info=ci, entity_draw_buffer=gs.draws, mesh_info_buffer=gs.mesh_infos,
                 meshlet_dispatch_buffer=disp, entity_buffer=gs.entities, entity_draw_count=scene.entity_draw_count,
                 dispatch_capacity=disp_cap, meshlet_buffer=gs.meshlets, draw_commands_buffer=draw,
                 material_buffer=gs.materials, draw_capacity=draw_cap, visibility_buffer=e_d,
                 meshlet_visibility_buffer=m_d, material_count=len(scene.materials))
        if int(ci["occlusion_pass"]) == 2:
            v.update(depth_pyramid=pyr, depth_pyramid_size=ps)
        views.append(v)
    eng.cull_views(views)
    torch.cuda.synchronize()
    eng.status()
    assert eng.meshlet_stream_culls() == 2
    n_drawn = []
    for ci, (disp, draw, e_d, m_d) in zip(cis, bufs):
        p = int(ci["occlusion_pass"])
        ref = run_oracle(oracle, scene, ci, evis if p else None, mvis if p else None, opyr if p == 2 else None,
                         ps if p == 2 else (0, 0))
        _, cmds = assert_same((host(disp), host(draw), None if e_d is None else host(e_d, np.uint32),
                               None if m_d is None else host(m_d, np.uint32)), ref)
        n_drawn.append(len(cmds))
    assert all(n > 0 for n in n_drawn), n_drawn
    eng.close()


def test_meshlet_outside_the_stream_latches_range_error(torch_mod, engine, oracle):
    """A stream that does not cover what is culled: no out-of-range read, ORBIT_E_RANGE in the context's status; the
    covered part of the scene is still culled correctly."""
    from orbit_amd._lib import E_RANGE, OrbitError

    torch = torch_mod
    scene, ci = _scene(seed=47, n=300)
    gs = GpuScene(torch, scene)
    half = len(scene.meshlets) // 2
    ms = engine.meshlet_stream(gs.meshlets, 0, half)
    engine.bind_meshlet_stream(ms)
    with pytest.raises(OrbitError) as ei:
        run_gpu(torch, engine, gs, ci)
        engine.status()
    assert ei.value.code == E_RANGE
    engine.status()  # the latch is cleared by the read
    # expansion of a record list that names a meshlet past the stream: {entity 0, offset 1, bit 0}, {entity 0, offset
    # half + 3, bit 0}
    vis = np.zeros(L.VISIBLE_HEADER + 12 * 2, dtype=np.uint8)
    vis[:8].view(np.uint32)[:] = [2, 2]
    vis[8:].view(np.uint32)[:] = [0, 1, 1, 0, half + 3, 1]
    draw = torch.zeros(L.DRAW_HEADER + 28 * 4, dtype=torch.uint8, device="cuda")
    engine.expand_visible_records(dev(torch, vis), gs.meshlets, draw, 4)
    with pytest.raises(OrbitError) as ei:
        engine.status()
    assert ei.value.code == E_RANGE
    n, cmds = L.draw_buffer_commands(host(draw))
    m1 = scene.meshlets[1]
    assert n == 2 and cmds[0]["meshlet_vertex_offset"] == m1["vertex_offset"] and cmds[1]["meshlet_index"] == half + 3
    ms.close()


def test_stream_argument_errors(torch_mod, engine):
    torch = torch_mod
    lib = engine._lib
    h = C.c_void_p()
    buf = torch.zeros(32 * 100, dtype=torch.uint8, device="cuda")
    assert lib.orbit_meshlet_stream_create(None, 0, 10, C.byref(h)) == _lib.E_INVALID
    assert lib.orbit_meshlet_stream_create(engine._ctx, 0, 10, None) == _lib.E_MISSING
    assert lib.orbit_meshlet_stream_create(engine._ctx, 0, 0, C.byref(h)) == _lib.E_INVALID and not h.value
    assert lib.orbit_meshlet_stream_create(engine._ctx, 2**32 - 5, 10, C.byref(h)) == _lib.E_INVALID  # 32-bit indices
    assert lib.orbit_meshlet_stream_create(engine._ctx, 0, 2**32, C.byref(h)) == _lib.E_INVALID
    assert lib.orbit_meshlet_stream_create(engine._ctx, 50, 100, C.byref(h)) == _lib.OK and h.value
    p = C.c_void_p(buf.data_ptr() - 50 * 32)
    assert lib.orbit_meshlet_stream_update(engine._ctx, None, p, 50, 100, None) == _lib.E_MISSING
    assert lib.orbit_meshlet_stream_update(engine._ctx, h, None, 50, 100, None) == _lib.E_MISSING
    assert lib.orbit_meshlet_stream_update(engine._ctx, h, p, 49, 10, None) == _lib.E_CAPACITY
    assert lib.orbit_meshlet_stream_update(engine._ctx, h, p, 50, 101, None) == _lib.E_CAPACITY
    assert lib.orbit_meshlet_stream_update(engine._ctx, h, p, 140, 11, None) == _lib.E_CAPACITY
    assert b"outside the stream" in lib.orbit_last_error(engine._ctx)
    assert lib.orbit_meshlet_stream_update(engine._ctx, h, p, 140, 10, None) == _lib.OK
    assert lib.orbit_meshlet_stream_update(engine._ctx, h, p, 50, 0, None) == _lib.OK  # empty range: nothing enqueued
    assert lib.orbit_ctx_bind_meshlet_stream(None, h) == _lib.E_INVALID
    assert lib.orbit_ctx_bind_meshlet_stream(engine._ctx, h) == _lib.OK
    assert lib.orbit_ctx_bind_meshlet_stream(engine._ctx, None) == _lib.OK
    assert lib.orbit_ctx_meshlet_stream_culls(None) == 0
    torch.cuda.synchronize()
    assert lib.orbit_meshlet_stream_destroy(h) == _lib.OK
    assert lib.orbit_meshlet_stream_destroy(None) == _lib.OK


# ----------------------------------------------------------------------------- the derived command chain
# The emit launch derives a survivor's vertex_offset / data_offset from one gathered base per chain and a prefix
# sum over 2-B counts (orbit_amd/csrc/meshlet_emit.hip, meshlet_emit_chain_body), valid where the stream's link bits
# hold; everywhere else it gathers.  The scenes of tests/scenes.py mix all of that ("uploaded" layout); these tests
# put the breaks where the launch's cases meet.
def _relayout(scene, seed, mode):
    """Rewrites vertex_offset / data_offset of every meshlet; `mode` places the chain breaks."""
    m = scene.meshlets
    n = len(m)
    mid = np.arange(n)
    if mode == "max_counts":  # the largest data sizes a Meshlet can state: 255 + ceil(765 / 4) words
        m["vertex_count"], m["triangle_count"] = 255, 255
    size = m["vertex_count"].astype(np.int64) + (m["triangle_count"].astype(np.int64) * 3 + 3) // 4
    lod0 = scene.mesh_infos["mesh_lods"][:, :, 0].reshape(-1)
    starts = np.zeros(n, dtype=bool)
    starts[lod0[lod0 < n]] = True  # every (mesh, LOD) range begins an allocation
    if mode in ("one_chain", "max_counts"):
        jump = np.zeros(n, dtype=bool)  # not even between meshes: the whole buffer is one chain
        sub = np.zeros(n, dtype=bool)
    elif mode == "record_boundaries":  # breaks exactly where a dispatch record ends (offset + 32 j)
        rel = mid - np.maximum.accumulate(np.where(starts, mid, 0))
        jump = starts | ((rel % 32 == 0) & (sc.rnd_f32(seed, 1, mid) < np.float32(0.5)))
        sub = starts | ((rel % 32 == 0) & (sc.rnd_f32(seed, 2, mid) < np.float32(0.5)))
    elif mode == "second_lane":  # the link of a record's lane 1 / lane 31: the ends of the per-lane mask
        rel = mid - np.maximum.accumulate(np.where(starts, mid, 0))
        jump = starts | (rel % 32 == 1) & (sc.rnd_f32(seed, 1, mid) < np.float32(0.5))
        sub = starts | (rel % 32 == 31) & (sc.rnd_f32(seed, 2, mid) < np.float32(0.5))
    else:
        raise ValueError(mode)
    gap = np.where(jump, sc.rnd_int(seed, 3, mid, 1, 5000), 0)
    data = np.cumsum(gap) + np.concatenate([[0], np.cumsum(size)[:-1]]) + 0xFFFF0000  # wraps past 2^32 on the way
    m["data_offset"] = (data & 0xFFFFFFFF).astype(np.uint32)
    m["vertex_offset"] = (sc.rnd_u64(seed, 4, np.cumsum(sub)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


@pytest.mark.parametrize("mode", ["one_chain", "max_counts", "record_boundaries", "second_lane"])
@pytest.mark.parametrize("lods", [1, 3])
def test_command_chain_where_its_cases_meet(torch_mod, engine, oracle, mode, lods):
    torch = torch_mod
    scene = sc.make_scene(41, 700, n_meshes=90, meshlets_per_mesh=(1, 200), lods=lods)
    _relayout(scene, 7, mode)
    cam = sc.default_camera(rot=(0.1, 0.3))
    ci = sc.make_cull_info(cam.view, cam.planes)
    gs = GpuScene(torch, scene)
    ref = run_oracle(oracle, scene, ci)
    ms = engine.meshlet_stream(gs.meshlets, 0, len(scene.meshlets))
    engine.bind_meshlet_stream(ms)
    got = run_gpu(torch, engine, gs, ci)
    assert engine.meshlet_stream_culls() == 1
    assert_same(got, ref)
    engine.bind_meshlet_stream(None)
    ms.close()


def test_partial_updates_keep_the_links_around_them(torch_mod, engine, oracle):
    """Link bits live 32 to a word and the bit of the meshlet BEHIND an updated range depends on the range's last
    meshlet: pieces updated in any order, and single meshlets rewritten later, must leave the stream as one full
    update would."""
    torch = torch_mod
    scene = sc.make_scene(43, 500, n_meshes=70, meshlets_per_mesh=(1, 120), lods=2)
    cam = sc.default_camera(rot=(0.1, 0.3))
    ci = sc.make_cull_info(cam.view, cam.planes)
    gs = GpuScene(torch, scene)
    n = len(scene.meshlets)
    rng = np.random.default_rng(5)
    cuts = np.unique(np.concatenate([[0, n], rng.integers(1, n, 40)]))
    pieces = list(zip(cuts[:-1], cuts[1:]))
    rng.shuffle(pieces)
    from orbit_amd.engine import MeshletStream

    ms = MeshletStream.__new__(MeshletStream)  # create without the initial full update
    ms._engine, ms._lib, ms._h, ms.first, ms.capacity = engine, engine._lib, C.c_void_p(), 0, n
    _lib.check(engine._lib.orbit_meshlet_stream_create(engine._ctx, 0, n, C.byref(ms._h)), engine._ctx)
    for a, b in pieces:
        ms.update(gs.meshlets, int(a), int(b - a))
    engine.bind_meshlet_stream(ms)
    assert_same(run_gpu(torch, engine, gs, ci), run_oracle(oracle, scene, ci))
    # rewrite single meshlets' offsets in place: breaks a chain in front of and behind each of them
    for i in rng.integers(1, n - 1, 25):
        scene.meshlets["data_offset"][i] += 17
        scene.meshlets["vertex_offset"][i + 1] ^= 0x55
    gs.meshlets.copy_(dev(torch, scene.meshlets))
    rng.shuffle(pieces)
    for a, b in pieces:  # the same pieces, another order
        ms.update(gs.meshlets, int(a), int(b - a))
    assert_same(run_gpu(torch, engine, gs, ci), run_oracle(oracle, scene, ci))
    # and the minimal form: one meshlet rewritten, only that meshlet updated
    j = int(rng.integers(40, n - 40))
    scene.meshlets["data_offset"][j] += 3
    gs.meshlets.copy_(dev(torch, scene.meshlets))
    ms.update(gs.meshlets, j, 1)
    assert_same(run_gpu(torch, engine, gs, ci), run_oracle(oracle, scene, ci))
    engine.bind_meshlet_stream(None)
    ms.close()


# ----------------------------------------------------------------------------- alpha classes, validation, lifetime
def test_alpha_classes_follow_the_materials_and_fall_back(torch_mod, engine, oracle):
    """set_materials: classes 0..2 are the material's alpha_mode; a material outside the table the classes were derived
    from and an alpha_mode the two bit planes cannot hold (>= 3: `1 << mode` still decides, meshlet_cull.comp:207)
    are looked up at cull time.  Same bits as the oracle in all cases, for both alpha filters of the renderer."""
    torch = torch_mod
    scene = sc.make_scene(47, 800, n_meshes=120, meshlets_per_mesh=(1, 90), lods=2)
    scene.materials["alpha_mode"][3::5] = 3   # bit 3 of a flag word
    scene.materials["alpha_mode"][4::7] = 40  # 1 << 40 is 0 in GLSL's 32-bit shift as the kernels define it (shl1)
    cam = sc.default_camera(rot=(0.1, 0.3))
    gs = GpuScene(torch, scene)
    ms = engine.meshlet_stream(gs.meshlets, 0, len(scene.meshlets))
    engine.bind_meshlet_stream(ms)
    n_mat = len(scene.materials)
    for flags, noskip in ((L.ALPHA_OPAQUE | L.ALPHA_MASKED, 0), (0xF, 0), (L.ALPHA_OPAQUE, L.ALPHA_MASKED | 8)):
        ci = sc.make_cull_info(cam.view, cam.planes, alpha_mode_flag=flags, noskip_alphamode=noskip)
        ref = run_oracle(oracle, scene, ci)
        for count in (n_mat, n_mat // 2, 1):
            ms.set_materials(gs.materials, count)
            torch.cuda.synchronize()
            for mc in (0, n_mat):  # alpha modes gathered / staged in LDS
                assert_same(run_gpu(torch, engine, gs, ci, material_count=mc), ref)
    # a stream that holds meshlets of class 3 keeps evaluating through its material indices
    assert engine.meshlet_class_culls() == 0 and engine.meshlet_stream_culls() == 18
    # the same scene with plain alpha modes: now the classes decide — for every filter the renderer uses
    scene.materials["alpha_mode"][3::5] = 1
    scene.materials["alpha_mode"][4::7] = 2
    gs.materials.copy_(dev(torch, scene.materials))
    ms.set_materials(gs.materials, n_mat)
    for flags, noskip in ((L.ALPHA_OPAQUE | L.ALPHA_MASKED, 0), (0xF, 0), (L.ALPHA_OPAQUE, L.ALPHA_MASKED | 8),
                          (L.ALPHA_TRANSPARENT, 0)):
        ci = sc.make_cull_info(cam.view, cam.planes, alpha_mode_flag=flags, noskip_alphamode=noskip)
        ref = run_oracle(oracle, scene, ci)
        before = engine.meshlet_class_culls()
        # set_materials resolves the stream's class flag before it returns: the FIRST cull behind it already takes
        # the class kernel — which variant runs never depends on when a read-back happens to land
        assert_same(run_gpu(torch, engine, gs, ci), ref)
        assert engine.meshlet_class_culls() == before + 1
        assert_same(run_gpu(torch, engine, gs, ci), ref)
        assert engine.meshlet_class_culls() == before + 2
        ms.set_materials(gs.materials, n_mat)  # again, with the device busy: still resolved on return
    # a table that ends before the last material: class 3 again, and the library notices
    ms.set_materials(gs.materials, n_mat - 1)
    before = engine.meshlet_class_culls()
    assert_same(run_gpu(torch, engine, gs, ci), ref)
    assert engine.meshlet_class_culls() == before
    ms.set_materials(gs.materials, n_mat)
    torch.cuda.synchronize()
    # another material buffer (a copy): the classes are not its mirror -> material indices again, same result
    other = dev(torch, scene.materials)
    before = engine.meshlet_stream_culls()
    disp_cap, draw_cap = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device="cuda")
    draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device="cuda")
    engine.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, scene.entity_draw_count, disp_cap)
    engine.meshlet_cull(ci, disp, gs.meshlets, draw, gs.entities, other, disp_cap, draw_cap)
    torch.cuda.synchronize()
    assert engine.meshlet_stream_culls() == before + 1
    n = L.draw_buffer_commands(ref[1])[0]
    assert np.array_equal(host(draw)[:4 + 28 * n], ref[1][:4 + 28 * n])
    ms.set_materials(None, 0)
    assert_same(run_gpu(torch, engine, gs, ci), ref)
    engine.bind_meshlet_stream(None)
    ms.close()


def test_validate_finds_a_missing_update(torch_mod, oracle):
    """caps.validate_streams: every cull served from the stream first compares it with the meshlet buffer;
    orbit_meshlet_stream_validate does the same on request.  A rewritten meshlet without an update -> ORBIT_E_STALE."""
    from orbit_amd.engine import Engine

    torch = torch_mod
    scene, ci = _scene(seed=53, n=300)
    gs = GpuScene(torch, scene)
    eng = Engine(0, max_entities=50_000, max_dispatches=400_000, max_draws=1_000_000, validate_streams=1)
    ms = eng.meshlet_stream(gs.meshlets, 0, len(scene.meshlets))
    ms.set_materials(gs.materials, len(scene.materials))
    eng.bind_meshlet_stream(ms)
    assert_same(run_gpu(torch, eng, gs, ci), run_oracle(oracle, scene, ci))
    eng.status()  # in step
    for field, delta in (("data_offset", 1), ("bounding_sphere", np.float32(0.5)), ("triangle_count", 1)):
        k = len(scene.meshlets) // 2
        scene.meshlets[field][k] += delta
        gs.meshlets.copy_(dev(torch, scene.meshlets))
        run_gpu(torch, eng, gs, ci)
        with pytest.raises(_lib.OrbitError) as err:
            eng.status()
        assert err.value.code == _lib.E_STALE
        ms.update(gs.meshlets, k, 1)
        assert_same(run_gpu(torch, eng, gs, ci), run_oracle(oracle, scene, ci))
        eng.status()
    # a material whose alpha_mode changed without set_materials
    scene.materials["alpha_mode"][0] ^= 1
    gs.materials.copy_(dev(torch, scene.materials))
    ms.validate(gs.meshlets, gs.materials)
    with pytest.raises(_lib.OrbitError) as err:
        eng.status()
    assert err.value.code == _lib.E_STALE
    ms.set_materials(gs.materials, len(scene.materials))
    ms.validate(gs.meshlets, gs.materials)
    eng.status()
    eng.bind_meshlet_stream(None)
    ms.close()
    eng.close()


def test_a_bound_stream_cannot_be_destroyed(torch_mod, engine):
    torch = torch_mod
    scene, _ = _scene(seed=59, n=40)
    gs = GpuScene(torch, scene)
    ms = engine.meshlet_stream(gs.meshlets, 0, len(scene.meshlets))
    engine.bind_meshlet_stream(ms)
    assert engine._lib.orbit_meshlet_stream_destroy(ms._h) == _lib.E_INVALID  # refused, nothing freed
    scene2, ci = _scene(seed=59, n=40)
    run_gpu(torch, engine, gs, ci)  # still usable
    assert engine.meshlet_stream_culls() >= 1
    engine.bind_meshlet_stream(None)
    assert engine._lib.orbit_meshlet_stream_destroy(ms._h) == 0
    ms._h = C.c_void_p()


def test_a_new_source_buffer_starts_the_readable_range_over(torch_mod, engine, oracle):
    """after an update from ANOTHER buffer pointer only the range derived from that pointer may be read —
    a cull that dispatches meshlets outside it latches ORBIT_E_RANGE instead of answering from the old buffer's copy;
    meshlets inside the hull of two updates that neither reached are empty (zero) meshlets, not stale ones."""
    torch = torch_mod
    scene, ci = _scene(seed=61, n=300)
    gs = GpuScene(torch, scene)
    n = len(scene.meshlets)
    ms = engine.meshlet_stream(gs.meshlets, 0, n)
    engine.bind_meshlet_stream(ms)
    ref = run_oracle(oracle, scene, ci)
    assert_same(run_gpu(torch, engine, gs, ci), ref)
    other = GpuScene(torch, scene)  # the same meshlets at another address
    ms.update(other.meshlets, 0, n // 4)  # only a quarter re-derived from the new pointer
    run_gpu(torch, engine, other, ci)
    with pytest.raises(_lib.OrbitError) as err:
        engine.status()
    assert err.value.code == _lib.E_RANGE
    ms.update(other.meshlets, n // 4, n - n // 4)
    assert_same(run_gpu(torch, engine, other, ci), ref)
    engine.status()
    engine.bind_meshlet_stream(None)
    ms.close()
