"""The entity stage from the mesh side table (orbit_meshlet_stream_update_meshes, include/orbit_abi_ext.h): a derived 32-B
{bounding sphere, lod_count, mesh_lods[0]} entry per mesh read instead of the mesh's 128-B MeshInfo line.  The dispatch
records (and, in pass 2, the entity bitset) must be the oracle's bit for bit — with every LOD count, with meshes no update
reached (they fall back to the MeshInfo), through every entry point that runs the entity stage — and a table that no
longer mirrors its buffer must be found."""
import numpy as np
import pytest

import scenes as sc
from orbit_amd import layouts as L
from test_gpu_parity import GpuScene, dev, host, run_oracle, torch_mod  # noqa: F401

pytestmark = pytest.mark.gpu

CAPS = dict(max_entities=300_000, max_dispatches=2_000_000, max_draws=4_000_000)


def _engine(torch, gs, meshes=None, **kw):
    """An engine with a stream of gs.meshlets bound whose mesh table holds `meshes` = (first, count) (None: all)."""
    from orbit_amd.engine import Engine

    eng = Engine(0, **dict(CAPS, **kw))
    ms = eng.meshlet_stream(gs.meshlets, 0, gs.meshlets.numel() // 32)
    n_meshes = gs.mesh_infos.numel() // 128
    first, count = (0, n_meshes) if meshes is None else meshes
    ms.update_meshes(gs.mesh_infos, first, count)
    eng.bind_meshlet_stream(ms)
    torch.cuda.synchronize()
    return eng, ms


def _close(eng, ms):
    eng.bind_meshlet_stream(None)
    ms.close()
    eng.close()


def _records(buf):
    n, recs = L.dispatch_buffer_records(buf)
    return int(n[0]), recs


@pytest.mark.parametrize("lods,meshes", [(1, None), (3, None), (8, None), (3, (10, 57)), (2, (0, 0))])
@pytest.mark.parametrize("occlusion_pass", [0, 2])
def test_entity_cull_from_the_side_table_is_the_oracle(torch_mod, oracle, lods, meshes, occlusion_pass):
    """Every LOD count (the table holds LOD 0 only: entities that pick another LOD read the MeshInfo), a table that holds
    only meshes 10 .. 66 (the others were never derived: fallback), an empty update; passes 0 and 2 (pyramid, bitset)."""
    torch = torch_mod
    scene = sc.make_scene(90 + lods, 3000, n_meshes=180, lods=lods, meshlets_per_mesh=(1, 70), extent=(60.0, 20.0, 60.0))
    cam = sc.default_camera(rot=(0.3, 0.7))
    rng = np.random.default_rng(9)
    kw, evis, mvis, pyr, ps = {}, None, None, None, (0, 0)
    if occlusion_pass == 2:
        kw = dict(p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
        evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
        mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
        pyr, d = oracle.depth_reduce(sc.make_depth(5, 320, 180, cam), 320, 180)
        ps = (d.width, d.height)
    ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=occlusion_pass, **kw)
    gs = GpuScene(torch, scene)
    eng, ms = _engine(torch, gs, meshes)
    cap_d = scene.max_dispatches() + 8
    disp = torch.full((L.DISPATCH_HEADER + 16 * cap_d + 64,), 0xAB, dtype=torch.uint8, device="cuda")
    evis_d = None if evis is None else dev(torch, evis)
    pyr_d = None if pyr is None else dev(torch, pyr)
    before = eng.mesh_side_culls()
    eng.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, scene.entity_draw_count, cap_d, visibility_buffer=evis_d,
                    depth_pyramid=pyr_d, depth_pyramid_size=ps)
    torch.cuda.synchronize()
    eng.status()
    assert eng.mesh_side_culls() - before == (0 if meshes == (0, 0) else 1)
    ref = run_oracle(oracle, scene, ci, evis=evis, mvis=mvis, pyr=pyr, psize=ps)
    n, recs = _records(host(disp))
    on, orecs = _records(ref[0])
    assert n == on > 0 and np.array_equal(recs.view(np.uint32), orecs.view(np.uint32)), "dispatch records differ"
    assert bool((host(disp)[12 + 16 * cap_d:] == 0xAB).all())
    if lods > 1:  # the scene must have exercised LODs other than 0 (read from the MeshInfo, not from the table)
        lod0 = {int(m["mesh_lods"][0][0]) for m in scene.mesh_infos}
        starts = {int(m["mesh_lods"][k][0]) for m in scene.mesh_infos for k in range(1, lods)}
        assert any(int(r["meshlet_offset"]) in starts and int(r["meshlet_offset"]) not in lod0 for r in orecs)
    if evis_d is not None:
        assert np.array_equal(host(evis_d, np.uint32), ref[2]), "entity visibility bitset differs"
    _close(eng, ms)


def test_every_entity_entry_point_reads_the_table(torch_mod, oracle):
    """orbit_cull_views (one launch and the chain) and orbit_cull_shard (one launch and the chain) with the table bound:
    the oracle's records and commands; a cull of ANOTHER mesh_info buffer (a copy) does not touch the table."""
    torch = torch_mod
    scene = sc.make_scene(95, 2600, n_meshes=300, lods=3, meshlets_per_mesh=(1, 70), extent=(60.0, 20.0, 60.0))
    cam = sc.default_camera(rot=(0.2, 0.4))
    ci = sc.make_cull_info(cam.view, cam.planes, alpha_mode_flag=L.ALPHA_ALL)
    gs = GpuScene(torch, scene)
    ref = run_oracle(oracle, scene, ci)
    on, orecs = _records(ref[0])
    ocn, ocmds = L.draw_buffer_commands(ref[1])
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    for cull_path in (1, 2):
        eng, ms = _engine(torch, gs, cull_path=cull_path)
        disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
        draw = torch.zeros(L.DRAW_HEADER + 28 * cap_c, dtype=torch.uint8, device="cuda")
        before = eng.mesh_side_culls()
        eng.cull_views([dict(cull_info=ci, entity_draw_buffer=gs.draws, mesh_info_buffer=gs.mesh_infos,
                             meshlet_dispatch_buffer=disp, entity_buffer=gs.entities, entity_draw_count=scene.entity_draw_count,
                             dispatch_capacity=cap_d, meshlet_buffer=gs.meshlets, draw_commands_buffer=draw,
                             material_buffer=gs.materials, draw_capacity=cap_c)])
        torch.cuda.synchronize()
        eng.status()
        assert eng.mesh_side_culls() - before == 1 and (eng.fused_culls() == 1) == (cull_path == 2)
        assert np.array_equal(host(disp)[:12 + 16 * on], ref[0][:12 + 16 * on])
        assert np.array_equal(host(draw)[:4 + 28 * ocn], ref[1][:4 + 28 * ocn])
        rec = torch.zeros(L.VISIBLE_HEADER + 12 * cap_d, dtype=torch.uint8, device="cuda")
        disp.zero_()
        draw.zero_()
        eng.cull_shard(ci, gs.draws, gs.mesh_infos, disp, gs.entities, 0, scene.entity_draw_count, cap_d, gs.meshlets,
                       gs.materials, rec, cap_d, draw_commands_buffer=draw, draw_capacity=cap_c)
        torch.cuda.synchronize()
        eng.status()
        assert eng.mesh_side_culls() - before == 2 and (eng.shard_culls() == 1) == (cull_path == 2)
        assert np.array_equal(host(disp)[:12 + 16 * on], ref[0][:12 + 16 * on])
        assert np.array_equal(host(draw)[:4 + 28 * ocn], ref[1][:4 + 28 * ocn])
        # a copy of the mesh infos is another buffer: the table is not consulted (and the results are the same)
        other = gs.mesh_infos.clone()
        disp.zero_()
        eng.entity_cull(ci, gs.draws, other, disp, gs.entities, scene.entity_draw_count, cap_d)
        torch.cuda.synchronize()
        assert eng.mesh_side_culls() - before == 2
        assert np.array_equal(host(disp)[:12 + 16 * on], ref[0][:12 + 16 * on])
        _close(eng, ms)


def test_a_stale_side_table_is_found(torch_mod, oracle):
    """A MeshInfo rewritten without an update: the cull uses the OLD sphere (that is what a derived copy does) — and both
    orbit_meshlet_stream_validate and a context with validate_streams latch ORBIT_E_STALE; after the update all is well."""
    from orbit_amd._lib import E_STALE, OrbitError

    torch = torch_mod
    scene = sc.make_scene(97, 500, n_meshes=60, lods=2, meshlets_per_mesh=(1, 40))
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, cam.planes)
    gs = GpuScene(torch, scene)
    eng, ms = _engine(torch, gs, validate_streams=1)
    cap_d = scene.max_dispatches() + 8
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
    eng.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, scene.entity_draw_count, cap_d)
    torch.cuda.synchronize()
    eng.status()  # in step
    good = disp.clone()
    # mesh 7's bounding sphere moves far away in the buffer; the table still holds the old one
    mi = gs.mesh_infos.view(torch.float32).view(-1, 32)
    keep = mi[7, :4].clone()
    mi[7, 0] = 1.0e6
    eng.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, scene.entity_draw_count, cap_d)
    torch.cuda.synchronize()
    with pytest.raises(OrbitError) as ei:
        eng.status()
    assert ei.value.code == E_STALE
    assert torch.equal(disp, good), "the cull must have read the derived copy"
    ng0 = int(good[:4].view(torch.int32).item())
    assert ng0 > 0
    ms.validate(gs.meshlets)
    torch.cuda.synchronize()
    with pytest.raises(OrbitError) as ei:
        eng.status()
    assert ei.value.code == E_STALE
    ms.update_meshes(gs.mesh_infos, 7, 1)
    eng.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, scene.entity_draw_count, cap_d)
    torch.cuda.synchronize()
    eng.status()
    scene.mesh_infos["bounding_sphere"][7][0] = 1.0e6
    on, orecs = _records(run_oracle(oracle, scene, ci)[0])
    n, recs = _records(host(disp))
    assert n == on and np.array_equal(recs.view(np.uint32), orecs.view(np.uint32))
    mi[7, :4] = keep
    ms.update_meshes(None, 0, 0)  # forget: the MeshInfos are read again
    before = eng.mesh_side_culls()
    eng.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, scene.entity_draw_count, cap_d)
    torch.cuda.synchronize()
    eng.status()
    ng = int(good[:4].view(torch.int32).item())  # (behind the records the buffer keeps the longer list's tail)
    assert eng.mesh_side_culls() == before and torch.equal(disp[:12 + 16 * ng], good[:12 + 16 * ng])
    _close(eng, ms)
