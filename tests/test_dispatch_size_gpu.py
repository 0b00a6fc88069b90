"""OrbitCaps.dispatch_size 64 / 128 (MESHLET_DISPATCH_SIZE, spec constant 0 of the two cull shaders:
src/graphics/device.rs:369-372 derives it from the device; shaders default to 64, entity_cull.comp:26): entity_cull writes
S-sized records, meshlet_cull consumes them — occlusion pass 0.  Held to the reference's own binaries run with that spec
constant (tests/golden/spirv_cull_dispatch_sizes.npz) and to the oracle on larger scenes, from the Meshlet buffer and from
the derived streams, as single calls and through orbit_cull_views; everything the reference itself leaves inconsistent at
S != 32 (the occlusion passes' visibility words, the task payload's 32 entries) is refused by name."""
import os

import numpy as np
import pytest

import scenes as sc
from orbit_amd import layouts as L
from test_gpu_parity import GpuScene, assert_same, dev, host, run_gpu, torch_mod  # noqa: F401
from test_spirv_vectors_cpu import GOLD, GOLD_S, S_CASES, load_s_case

pytestmark = pytest.mark.gpu

CAPS = dict(max_entities=100_000, max_dispatches=400_000, max_draws=2_000_000)
SOURCES = ["meshlet_buffer", "meshlet_stream", "meshlet_stream_classes"]


def make_engine(source, S, **extra):
    from orbit_amd.engine import Engine
    from stream_engine import StreamEngine

    kw = dict(CAPS, dispatch_size=S, **extra)
    return Engine(0, **kw) if source == "meshlet_buffer" else StreamEngine(0, classes=source == "meshlet_stream_classes", **kw)


def run_oracle_s(oracle, scene, ci, S):
    with oracle.dispatch_size(S):
        disp, _, d1 = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                         scene.entities, scene.max_dispatches() + 8)
        draw, _, d2 = oracle.meshlet_cull(ci, disp, scene.meshlets, scene.lod0_meshlets + 8, scene.entities, scene.materials)
    assert d1 == 0 and d2 == 0
    return disp, draw, None, None


@pytest.mark.parametrize("source", SOURCES)
@pytest.mark.parametrize("S,proj,kind", S_CASES)
def test_product_equals_the_reference_binaries_at_other_dispatch_sizes(torch_mod, source, S, proj, kind):
    torch = torch_mod
    c = load_s_case(np.load(GOLD), np.load(GOLD_S), S, proj, kind)
    n_draws = int(np.frombuffer(c["draws"][:4].tobytes(), np.uint32)[0])
    eng = make_engine(source, S)
    cap_d, cap_c = c["caps"]
    disp = torch.full((L.DISPATCH_HEADER + 16 * cap_d + 64,), 0xAB, dtype=torch.uint8, device="cuda")
    draw = torch.full((L.DRAW_HEADER + 28 * cap_c + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    mesh, ents, mlt, mats = (dev(torch, c[k]) for k in ("mesh_infos", "entities", "meshlets", "materials"))
    eng.entity_cull(c["ci"], dev(torch, c["draws"]), mesh, disp, ents, n_draws, cap_d)
    eng.meshlet_cull(c["ci"], disp, mlt, draw, ents, mats, cap_d, cap_c)
    torch.cuda.synchronize()
    eng.status()
    nrec, ndraw = int(c["spv_dispatch"][:4].view(np.uint32)[0]), int(c["spv_draw"][:4].view(np.uint32)[0])
    assert np.array_equal(host(disp)[:L.DISPATCH_HEADER + 16 * nrec], c["spv_dispatch"]), "dispatch records differ from the binary's"
    assert np.array_equal(host(draw)[:L.DRAW_HEADER + 28 * ndraw], c["spv_draw"]), "draw commands differ from the binary's"
    assert bool((disp[L.DISPATCH_HEADER + 16 * cap_d:] == 0xAB).all()) and bool((draw[L.DRAW_HEADER + 28 * cap_c:] == 0xCD).all())
    eng.close()


@pytest.mark.parametrize("source", SOURCES)
@pytest.mark.parametrize("S", [64, 128])
@pytest.mark.parametrize("seed,n,lods,ortho", [(61, 3000, 2, False), (62, 700, 1, True), (63, 37, 3, False)])
def test_pass0_at_other_dispatch_sizes_against_the_oracle(torch_mod, oracle, source, S, seed, n, lods, ortho):
    """Larger scenes, meshes of 1 .. 300 meshlets (records of every fill, entities of up to 3 / 5 / 10 records at 128 / 64 /
    32), LOD chains, both projections; the same list as the S = 32 oracle's."""
    torch = torch_mod
    scene = sc.make_scene(seed, n, meshlets_per_mesh=(1, 300), lods=lods)
    cam = sc.default_camera(rot=(0.3, 0.2))
    if ortho:
        ci = sc.make_cull_info(cam.view, cam.planes[:4], projection_type=1, alpha_mode_flag=L.ALPHA_ALL)
    else:
        ci = sc.make_cull_info(cam.view, cam.planes)
    gs = GpuScene(torch, scene)
    eng = make_engine(source, S)
    ref = run_oracle_s(oracle, scene, ci, S)
    recs, cmds = assert_same(run_gpu(torch, eng, gs, ci), ref)
    assert recs["meshlet_count"].max() > 32 and recs["meshlet_count"].max() <= S and len(cmds) > 0
    ref32 = run_oracle_s(oracle, scene, ci, 32)
    assert np.array_equal(ref[1], ref32[1]), "the command list depends on the dispatch size?"
    eng.close()


@pytest.mark.parametrize("S", [64, 128])
def test_cull_views_at_other_dispatch_sizes(torch_mod, oracle, S):
    """orbit_cull_views (and through it orbit_frame_late's cull chains): several views in one call, each view's S-sized
    records split on its own scratch; the one-launch cull is not taken (it hands 32-meshlet records over inside itself),
    whatever cull_path says."""
    torch = torch_mod
    from orbit_amd import passes
    from orbit_amd.engine import Engine

    scene = sc.make_scene(64, 1500, meshlets_per_mesh=(1, 200), n_meshes=300, extent=(60.0, 15.0, 60.0))
    gs = GpuScene(torch, scene)
    cam = sc.default_camera()
    cis = [sc.make_cull_info(cam.view, cam.planes)]
    for cascade in range(3):
        hci, _, _ = passes.shadow_cascade(direction=(-0.45, 0.2, 0.1, 0.86), camera_position=(0.0, 2.0, 0.0),
                                          camera_orientation=(0.0, 0.0, 0.0, 1.0), camera_fov=float(cam.fov),
                                          camera_near_clip=cam.z_near, camera_aspect_ratio=cam.aspect,
                                          cascade_index=cascade, max_shadow_distance=80.0)
        cis.append(passes.cull_info_to_gpu(hci))
    for cull_path in (0, 2):
        eng = Engine(0, dispatch_size=S, cull_path=cull_path, max_views=4, **CAPS)
        cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
        views, bufs = [], []
        for ci in cis:
            disp = torch.full((L.DISPATCH_HEADER + 16 * cap_d,), 0xAB, dtype=torch.uint8, device="cuda")
            draw = torch.full((L.DRAW_HEADER + 28 * cap_c,), 0xCD, dtype=torch.uint8, device="cuda")
            bufs.append((disp, draw))
            views.append(dict(cull_info=ci, entity_draw_buffer=gs.draws, mesh_info_buffer=gs.mesh_infos,
                              meshlet_dispatch_buffer=disp, entity_buffer=gs.entities, entity_draw_count=scene.entity_draw_count,
                              dispatch_capacity=cap_d, meshlet_buffer=gs.meshlets, draw_commands_buffer=draw,
                              material_buffer=gs.materials, draw_capacity=cap_c))
        for rep in range(2):
            eng.cull_views(views)
            torch.cuda.synchronize()
            eng.status()
        assert eng.fused_culls() == 0
        total = 0
        for ci, (disp, draw) in zip(cis, bufs):
            _, cmds = assert_same((host(disp), host(draw), None, None), run_oracle_s(oracle, scene, ci, S))
            total += len(cmds)
        assert total > 0
        eng.close()


def test_capacities_at_another_dispatch_size(torch_mod, oracle):
    """dispatch_capacity counts the caller's S-sized records: an overflow latches ORBIT_E_CAPACITY, the header is clamped,
    nothing is written past either buffer, and the commands of the records that fit are the oracle's prefix."""
    torch = torch_mod
    from orbit_amd._lib import E_CAPACITY, OrbitError

    scene = sc.make_scene(65, 400, meshlets_per_mesh=(40, 120))
    gs = GpuScene(torch, scene)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, [])  # no planes: every entity dispatches
    eng = make_engine("meshlet_buffer", 64)
    with oracle.dispatch_size(64):
        full, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                        scene.entities, scene.max_dispatches() + 8)
        nfull = int(full[:4].view(np.uint32)[0])
        cut = nfull - 17
        od, _, dropped = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                            scene.entities, cut)
        assert dropped == 17
        oc, _, _ = oracle.meshlet_cull(ci, od, scene.meshlets, scene.lod0_meshlets + 8, scene.entities, scene.materials)
    gpu = run_gpu(torch, eng, gs, ci, disp_cap=cut)  # (run_gpu checks the red zones behind both capacities)
    with pytest.raises(OrbitError) as ei:
        eng.status()
    assert ei.value.code == E_CAPACITY
    assert_same(gpu, (od, oc, None, None))
    eng.close()


def test_what_the_reference_leaves_inconsistent_is_refused_by_name(torch_mod):
    torch = torch_mod
    from orbit_amd._lib import E_INVALID, OrbitError
    from orbit_amd.engine import Engine

    scene = sc.make_scene(66, 100)
    gs = GpuScene(torch, scene)
    cam = sc.default_camera()
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
    draw = torch.zeros(L.DRAW_HEADER + 28 * cap_c, dtype=torch.uint8, device="cuda")
    words = torch.zeros(4096, dtype=torch.int32, device="cuda")
    with pytest.raises(OrbitError):
        Engine(0, dispatch_size=48)
    with pytest.raises(OrbitError):
        Engine(0, dispatch_size=256)
    eng = Engine(0, dispatch_size=64, **CAPS)
    ci0 = sc.make_cull_info(cam.view, cam.planes)
    for p in (1, 2):
        ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=p, p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
        with pytest.raises(OrbitError, match="scene.rs:427") as ei:
            eng.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, scene.entity_draw_count, cap_d, visibility_buffer=words)
        assert ei.value.code == E_INVALID
        with pytest.raises(OrbitError, match="scene.rs:427"):
            eng.meshlet_cull(ci, disp, gs.meshlets, draw, gs.entities, gs.materials, cap_d, cap_c, meshlet_visibility_buffer=words)
    eng.entity_cull(ci0, gs.draws, gs.mesh_infos, disp, gs.entities, scene.entity_draw_count, cap_d)
    task = torch.zeros(44 * cap_d, dtype=torch.uint8, device="cuda")
    with pytest.raises(OrbitError, match="types.glsl:196-200"):
        eng.meshlet_task_cull(ci0, disp, gs.meshlets, task, gs.entities, gs.materials, cap_d)
    rec = torch.zeros(L.VISIBLE_HEADER + 12 * cap_d, dtype=torch.uint8, device="cuda")
    with pytest.raises(OrbitError, match="dispatch_size 32"):
        eng.meshlet_cull_visible_records(ci0, disp, gs.meshlets, rec, gs.entities, gs.materials, cap_d, cap_d)
    with pytest.raises(OrbitError, match="dispatch_size 32"):
        eng.cull_shard(ci0, gs.draws, gs.mesh_infos, disp, gs.entities, 0, scene.entity_draw_count, cap_d, gs.meshlets,
                       gs.materials, rec, cap_d)
    torch.cuda.synchronize()
    eng.status()  # nothing was enqueued by the refused calls, nothing latched
    eng.close()
