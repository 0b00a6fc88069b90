"""The one-launch cull (orbit_amd/csrc/cull_fused.hip; orbit_cull_views, include/orbit_abi.h) beyond what the parity
suite checks through tests/fused_engine.py: its hand-offs between workgroups.  The launch keeps per-chunk / per-tile
flags and five counters in context scratch and relies on (a) tickets, so that no wait depends on dispatch order or
residency, (b) write-through stores + sc1 loads for everything one workgroup reads of another's, (c) the last
workgroup zeroing the flags.  So: the same context over and over with scenes of different sizes (a stale flag of a
larger launch would be read by the next one), under load from another stream, replayed from a graph, several views in one
launch, capacities that cut the lists, and a scene far larger than the path is meant for (thousands of tiles: the
look-back's multi-step walk).  Everything is compared with the oracle and with the launch chain, word for word."""
import numpy as np
import pytest

import scenes as sc
from orbit_amd import layouts as L
from test_gpu_parity import GpuScene, assert_same, dev, host, run_oracle, torch_mod  # noqa: F401

pytestmark = pytest.mark.gpu


def make_view(torch, gs, ci, disp_cap, draw_cap, evis=None, mvis=None, pyr=None, psize=(0, 0), fill=0):
    disp = torch.full((L.DISPATCH_HEADER + 16 * disp_cap + 256,), fill, dtype=torch.uint8, device="cuda")
    draw = torch.full((L.DRAW_HEADER + 28 * draw_cap + 256,), fill, dtype=torch.uint8, device="cuda")
    e_d = None if evis is None else dev(torch, evis)
    m_d = None if mvis is None else dev(torch, mvis)
    s = gs.scene
    v = dict(cull_info=ci, entity_draw_buffer=gs.draws, mesh_info_buffer=gs.mesh_infos, meshlet_dispatch_buffer=disp,
             entity_buffer=gs.entities, entity_draw_count=s.entity_draw_count, dispatch_capacity=disp_cap,
             meshlet_buffer=gs.meshlets, draw_commands_buffer=draw, material_buffer=gs.materials, draw_capacity=draw_cap,
             visibility_buffer=e_d, meshlet_visibility_buffer=m_d, material_count=len(s.materials))
    if pyr is not None:
        v.update(depth_pyramid=pyr, depth_pyramid_size=psize)
    return v, (disp, draw, e_d, m_d)


def result(bufs):
    disp, draw, e_d, m_d = bufs
    return (host(disp), host(draw), None if e_d is None else host(e_d, np.uint32),
            None if m_d is None else host(m_d, np.uint32))


def engines(**kw):
    from orbit_amd.engine import Engine

    caps = dict(max_entities=70_000, max_dispatches=400_000, max_draws=2_000_000)
    caps.update(kw)
    return Engine(0, cull_path=2, **caps), Engine(0, cull_path=1, **caps)


def test_same_context_scenes_of_changing_size_under_load(torch_mod, oracle):
    """40 culls on ONE context, scenes of 30 ... 6 000 entities in an order that makes every launch follow a larger or
    smaller one, all three occlusion passes, while another stream keeps the memory system busy: every launch must find
    the flags zeroed and every output must equal the oracle's."""
    torch = torch_mod
    fused, chain = engines()
    cam = sc.default_camera(rot=(0.6, 0.8))
    sizes = [6000, 30, 900, 2500, 64, 4100, 257, 1]
    scenes = [sc.make_scene(100 + i, n, n_meshes=max(1, n // 3), meshlets_per_mesh=(1, 70), lods=2) for i, n in enumerate(sizes)]
    gss = [GpuScene(torch, s) for s in scenes]
    W, H = 320, 180
    depth = sc.make_depth(5, W, H, cam)
    opyr, d = oracle.depth_reduce(depth, W, H)
    pyr, ps = dev(torch, opyr), (d.width, d.height)
    rng = np.random.default_rng(3)
    noise = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    refs = {}
    for it in range(40):
        k = it % len(scenes)
        occ = (it // len(scenes)) % 3
        scene, gs = scenes[k], gss[k]
        key = (k, occ)
        if key not in refs:
            evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32) if occ else None
            mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32) if occ else None
            kw = dict(p00=cam.p00, p11=cam.p11, z_near=cam.z_near) if occ == 2 else {}
            ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=occ, **kw)
            ref = run_oracle(oracle, scene, ci, evis, mvis, opyr if occ == 2 else None, ps if occ == 2 else (0, 0))
            refs[key] = (ci, evis, mvis, ref)
        ci, evis, mvis, ref = refs[key]
        with torch.cuda.stream(side):  # load next to the cull: streams 256 MB through the caches
            noise.mul_(1.0001)
        v, bufs = make_view(torch, gs, ci, scene.max_dispatches() + 8, scene.lod0_meshlets + 8, evis, mvis,
                            pyr if occ == 2 else None, ps, fill=0xAB)
        fused.cull_views([v])
        torch.cuda.synchronize()
        fused.status()
        assert_same(result(bufs), ref)
        disp, draw = bufs[0], bufs[1]  # red zones: nothing behind the capacities
        assert bool((disp[L.DISPATCH_HEADER + 16 * (scene.max_dispatches() + 8):] == 0xAB).all())
        assert bool((draw[L.DRAW_HEADER + 28 * (scene.lod0_meshlets + 8):] == 0xAB).all())
    assert fused.fused_culls() == 40 and chain.fused_culls() == 0
    fused.close()
    chain.close()


def test_far_beyond_its_size_class_equals_the_chain(torch_mod, oracle):
    """60 000 entities x up to 70 meshlets forced through the one-launch path (cull_path = 2): ~5 000 tiles on ~500
    workgroups, i.e. every wave takes several tickets and the look-back walks more than one step of 64 tiles.  Same
    records and commands as the launch chain, word for word (and as the oracle)."""
    torch = torch_mod
    fused, chain = engines()
    scene = sc.make_scene(77, 60_000, n_meshes=2000, meshlets_per_mesh=(1, 70), extent=(200.0, 30.0, 200.0))
    gs = GpuScene(torch, scene)
    cam = sc.default_camera(rot=(0.8, 0.6))
    ci = sc.make_cull_info(cam.view, cam.planes)
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    va, ba = make_view(torch, gs, ci, cap_d, cap_c)
    vb, bb = make_view(torch, gs, ci, cap_d, cap_c)
    for _ in range(3):
        fused.cull_views([va])
    chain.cull_views([vb])
    torch.cuda.synchronize()
    fused.status()
    chain.status()
    assert fused.fused_culls() == 3 and chain.fused_culls() == 0
    n_rec = int(ba[0][:4].view(torch.int32).item())
    n = int(ba[1][:4].view(torch.int32).item())
    assert n_rec > 16 * 1000 and n > 10_000, (n_rec, n)
    assert torch.equal(ba[0][:12 + 16 * n_rec], bb[0][:12 + 16 * n_rec]), "dispatch records differ from the launch chain's"
    assert torch.equal(ba[1][:4 + 28 * n], bb[1][:4 + 28 * n]), "draw commands differ from the launch chain's"
    assert_same(result(ba), run_oracle(oracle, scene, ci))
    fused.close()
    chain.close()


def test_capacities_cut_the_lists_and_say_so(torch_mod, oracle):
    """A dispatch buffer and a draw buffer that are too small: the entries that fit are the canonical prefix, nothing is
    written behind the capacity, ORBIT_E_CAPACITY is latched — as on the launch chain."""
    torch = torch_mod
    from orbit_amd._lib import E_CAPACITY, OrbitError

    fused, chain = engines()
    scene = sc.make_scene(21, 1500, n_meshes=300, meshlets_per_mesh=(1, 70))
    gs = GpuScene(torch, scene)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, cam.planes)
    full = run_oracle(oracle, scene, ci)
    n_rec, n = int(full[0][:4].view(np.uint32)[0]), int(full[1][:4].view(np.uint32)[0])
    assert n_rec > 200 and n > 400
    for cap_d, cap_c in ((scene.max_dispatches() + 8, n // 3), (n_rec // 2, scene.lod0_meshlets + 8)):
        ref = run_oracle(oracle, scene, ci, disp_cap=cap_d, draw_cap=cap_c)
        v, bufs = make_view(torch, gs, ci, cap_d, cap_c, fill=0xEE)
        fused.cull_views([v])
        torch.cuda.synchronize()
        with pytest.raises(OrbitError) as ei:
            fused.status()
        assert ei.value.code == E_CAPACITY
        disp, draw = host(bufs[0]), host(bufs[1])
        r_rec, r_n = int(ref[0][:4].view(np.uint32)[0]), int(ref[1][:4].view(np.uint32)[0])
        assert np.array_equal(disp[:12 + 16 * r_rec], ref[0][:12 + 16 * r_rec])
        assert np.array_equal(draw[:4 + 28 * r_n], ref[1][:4 + 28 * r_n])
        assert (disp[12 + 16 * cap_d:] == 0xEE).all() and (draw[4 + 28 * cap_c:] == 0xEE).all()
    # the context is as usable as before
    v, bufs = make_view(torch, gs, ci, scene.max_dispatches() + 8, scene.lod0_meshlets + 8)
    fused.cull_views([v])
    torch.cuda.synchronize()
    fused.status()
    assert_same(result(bufs), full)
    fused.close()
    chain.close()


def test_nothing_survives_and_nothing_to_cull(torch_mod, oracle):
    """Planes that reject everything: {0,1,1} dispatch header and an empty draw list (both written inside the launch);
    entity_draw_count = 0 is the launch chain's job (cleared headers)."""
    torch = torch_mod
    fused, chain = engines()
    scene = sc.make_scene(8, 700, n_meshes=100)
    gs = GpuScene(torch, scene)
    cam = sc.default_camera()
    away = np.array([[0.0, 0.0, 1.0, -1.0e6]], np.float32)  # in front of a plane a thousand kilometres behind the scene
    ci = sc.make_cull_info(cam.view, away)
    v, bufs = make_view(torch, gs, ci, scene.max_dispatches() + 8, scene.lod0_meshlets + 8, fill=0x55)
    fused.cull_views([v])
    torch.cuda.synchronize()
    fused.status()
    assert host(bufs[0])[:12].view(np.uint32).tolist() == [0, 1, 1]
    assert int(host(bufs[1])[:4].view(np.uint32)[0]) == 0
    assert_same(result(bufs)[:2] + (None, None), run_oracle(oracle, scene, ci))
    v["entity_draw_count"] = 0
    fused.cull_views([v])
    torch.cuda.synchronize()
    fused.status()
    assert fused.fused_culls() == 1  # the second call took the chain
    assert host(bufs[0])[:12].view(np.uint32).tolist() == [0, 1, 1] and int(host(bufs[1])[:4].view(np.uint32)[0]) == 0
    fused.close()
    chain.close()


def test_five_views_in_one_launch_group_and_as_a_replayed_graph(torch_mod, oracle):
    """The culls of one frame of the reference — the early pass and four orthographic cascades (forward.rs:286-403,
    shadow_renderer.rs:391-403) — as ONE call: two launches (perspective, orthographic), every view on flags of its own.
    Then the same call captured into a graph and replayed ten times onto changing input: the flags a replay finds are
    the ones the previous replay's last workgroup zeroed."""
    torch = torch_mod
    from orbit_amd import passes
    from orbit_amd.engine import Engine

    scene = sc.make_scene(17, 2600, n_meshes=400, extent=(70.0, 15.0, 70.0))
    gs = GpuScene(torch, scene)
    cam = sc.default_camera()
    rng = np.random.default_rng(17)
    evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    cis = [sc.make_cull_info(cam.view, cam.planes, occlusion_pass=1)]
    for cascade in range(4):
        hci, _, _ = passes.shadow_cascade(direction=(-0.45, 0.2, 0.1, 0.86), camera_position=(0.0, 2.0, 0.0),
                                          camera_orientation=(0.0, 0.0, 0.0, 1.0), camera_fov=float(cam.fov),
                                          camera_near_clip=cam.z_near, camera_aspect_ratio=cam.aspect,
                                          cascade_index=cascade, max_shadow_distance=80.0)
        cis.append(passes.cull_info_to_gpu(hci))
    eng = Engine(0, max_entities=8192, max_dispatches=40_000, max_draws=200_000, max_views=5)  # cull_path 0: the library's choice
    disp_cap, draw_cap = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    views, bufs = [], []
    for k, ci in enumerate(cis):
        v, b = make_view(torch, gs, ci, disp_cap, draw_cap, evis if k == 0 else None, mvis if k == 0 else None)
        views.append(v)
        bufs.append(b)
    refs = [run_oracle(oracle, scene, ci, evis if k == 0 else None, mvis if k == 0 else None) for k, ci in enumerate(cis)]
    for _ in range(3):
        eng.cull_views(views)
    torch.cuda.synchronize()
    eng.status()
    assert eng.fused_culls() == 15
    drawn = [len(assert_same(result(b), r)[1]) for b, r in zip(bufs, refs)]
    assert drawn[0] > 0 and sum(drawn[1:]) > 0, drawn
    # the same call as a graph; between replays the entities move (a new model matrix for every entity), so every replay
    # computes something else on the same flags
    arr, keep = eng.prepare_views(views)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        eng.cull_views_prepared(arr, stream=s)  # warm
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            eng.cull_views_prepared(arr, stream=s)
    ent0 = scene.entities.copy()
    for rep in range(10):
        moved = ent0.copy()
        m = moved["model_matrix"] if moved.dtype.names else None
        if m is not None:
            m[:, 12] += np.float32(0.75 * rep)  # translation x of the column-major matrix
        scene.entities = moved
        gs.entities.copy_(dev(torch, moved))
        g.replay()
        torch.cuda.synchronize()
        eng.status()
        for k, (b, ci) in enumerate(zip(bufs, cis)):
            assert_same(result(b), run_oracle(oracle, scene, ci, evis if k == 0 else None, mvis if k == 0 else None))
    scene.entities = ent0
    del keep
    eng.close()


def test_two_pass_frame_through_the_one_launch_path(torch_mod, oracle):
    """A whole two-pass frame as the renderer issues it (forward.rs:286-298,391-403): pass 1 -> depth_reduce -> pass 2,
    both culls on the one-launch path, 1080p pyramid, BASELINE config 3's shape — entity and meshlet bits included."""
    torch = torch_mod
    from orbit_amd.engine import Engine, depth_pyramid_desc

    W, H = 1920, 1080
    scene = sc.make_scene(33, 2900, n_meshes=600, meshlets_per_mesh=(4, 40), extent=(150.0, 40.0, 150.0))
    gs = GpuScene(torch, scene)
    cam = sc.default_camera()
    rng = np.random.default_rng(33)
    evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    depth = sc.make_depth(33, W, H, cam)
    d = depth_pyramid_desc(W, H)
    eng = Engine(0, max_entities=8192, max_dispatches=40_000, max_draws=200_000)
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    ci1 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=1)
    ci2 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
    gpyr = torch.zeros(d.total_texels, dtype=torch.float32, device="cuda")
    v1, b1 = make_view(torch, gs, ci1, cap_d, cap_c, evis, mvis)
    v2, b2 = make_view(torch, gs, ci2, cap_d, cap_c, evis, mvis, gpyr, (d.width, d.height))
    # the two passes share the visibility buffers (scene.rs:375-382): pass 2 reads what pass 1 read, and writes
    v2["visibility_buffer"], v2["meshlet_visibility_buffer"] = v1["visibility_buffer"], v1["meshlet_visibility_buffer"]
    eng.cull_views([v1])
    eng.depth_reduce(dev(torch, depth).view(torch.float32), W, H, gpyr)
    eng.cull_views([v2])
    torch.cuda.synchronize()
    eng.status()
    assert eng.fused_culls() == 2
    opyr, _ = oracle.depth_reduce(depth, W, H)
    r1 = run_oracle(oracle, scene, ci1, evis, mvis)
    assert_same((host(b1[0]), host(b1[1]), None, None), r1[:2] + (None, None))
    r2 = run_oracle(oracle, scene, ci2, evis, mvis, opyr, (d.width, d.height))
    assert_same((host(b2[0]), host(b2[1]), host(b1[2], np.uint32), host(b1[3], np.uint32)), r2)
    eng.close()
