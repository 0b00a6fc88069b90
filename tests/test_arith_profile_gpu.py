"""OrbitCaps.arith_profile = ORBIT_ARITH_CONTRACTED: the one point of the cull shaders SPIR-V leaves to a driver's compiler
that a GPU compiler most likely decides the other way — every OpDot / OpMatrixTimesVector / OpMatrixTimesMatrix / Length /
Distance as an fma chain.  The PRODUCT's contracted build (orbit_amd/csrc/*_contracted.hip) against the reference's
binaries executed that way (tests/golden/spirv_cull_contracted.npz; the oracle's restatement is held to the same vectors on
the CPU) — every cull path and both meshlet sources — and against the contracted oracle on larger scenes in every pass.
The canonical profile stays the default and stays what every other test of the suite checks."""
import numpy as np
import pytest

import scenes as sc
from orbit_amd import layouts as L
from test_gpu_parity import GpuScene, assert_same, dev, host, run_gpu, torch_mod  # noqa: F401
from test_spirv_vectors_cpu import CASES, load_contracted_case, vectors  # noqa: F401

pytestmark = pytest.mark.gpu

PATHS = ["meshlet_buffer", "meshlet_stream", "meshlet_stream_classes", "one_launch"]


def make_engine(path, **caps):
    from orbit_amd.engine import Engine
    from stream_engine import StreamEngine

    kw = dict(arith_profile=1, **caps)
    if path == "one_launch":
        from fused_engine import FusedEngine

        return FusedEngine(0, **kw)
    return Engine(0, cull_path=1, **kw) if path == "meshlet_buffer" else StreamEngine(
        0, classes=path == "meshlet_stream_classes", cull_path=1, **kw)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("name", CASES)
def test_contracted_product_equals_the_binaries_run_contracted(torch_mod, vectors, name, path):
    torch = torch_mod
    c = load_contracted_case(vectors, name)
    p = int(c["ci"]["occlusion_pass"])
    n_draws = int(np.frombuffer(c["draws"][:4].tobytes(), np.uint32)[0])
    cap_d, cap_c = c["caps"]
    eng = make_engine(path, max_entities=4096, max_dispatches=cap_d + 64, max_draws=cap_c + 64)
    g = {k: dev(torch, c[k]) for k in ("draws", "mesh_infos", "entities", "meshlets", "materials")}
    evis = dev(torch, c["evis"]) if p else None
    mvis = dev(torch, c["mvis"]) if p else None
    pyr = dev(torch, c["pyr"]) if p == 2 else None
    ps = c["ps"] if p == 2 else (0, 0)
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
    draw = torch.zeros(L.DRAW_HEADER + 28 * cap_c, dtype=torch.uint8, device="cuda")
    eng.entity_cull(c["ci"], g["draws"], g["mesh_infos"], disp, g["entities"], n_draws, cap_d, visibility_buffer=evis,
                    depth_pyramid=pyr, depth_pyramid_size=ps)
    eng.meshlet_cull(c["ci"], disp, g["meshlets"], draw, g["entities"], g["materials"], cap_d, cap_c,
                     meshlet_visibility_buffer=mvis, depth_pyramid=pyr, depth_pyramid_size=ps,
                     material_count=len(c["materials"]))
    torch.cuda.synchronize()
    eng.status()
    nrec, ndraw = int(c["spv_dispatch"][:4].view(np.uint32)[0]), int(c["spv_draw"][:4].view(np.uint32)[0])
    assert np.array_equal(host(disp)[:L.DISPATCH_HEADER + 16 * nrec], c["spv_dispatch"]), "dispatch records differ"
    assert np.array_equal(host(draw)[:L.DRAW_HEADER + 28 * ndraw], c["spv_draw"]), "draw commands differ"
    if p:
        assert np.array_equal(host(evis, np.uint32), c["spv_evis"]) and np.array_equal(host(mvis, np.uint32), c["spv_mvis"])
    if path != "one_launch":  # the mesh-shading path on the binary's records, from the same visibility words
        mvis_t = dev(torch, c["mvis"]) if p else None
        disp_t = torch.zeros_like(disp)
        disp_t[:len(c["spv_dispatch"])] = dev(torch, c["spv_dispatch"])
        task = torch.full((44 * cap_d + 64,), 0xEE, dtype=torch.uint8, device="cuda")
        eng.meshlet_task_cull(c["ci"], disp_t, g["meshlets"], task, g["entities"], g["materials"], cap_d,
                              meshlet_visibility_buffer=mvis_t, depth_pyramid=pyr, depth_pyramid_size=ps)
        torch.cuda.synchronize()
        eng.status()
        n = len(c["spv_task_records"])
        assert np.array_equal(host(task)[:44 * n], c["spv_task_records"].view(np.uint8).reshape(-1)), "task records differ"
        if p == 2:
            assert np.array_equal(host(mvis_t, np.uint32), c["spv_task_mvis"])
    eng.close()


def test_the_two_profiles_differ_on_the_knife_edge_and_only_the_contracted_one_matches(torch_mod, vectors):
    """The default context still gives the canonical bytes on the case the contracted vectors decide the other way."""
    torch = torch_mod
    from orbit_amd.engine import Engine

    name = "p0_persp_knife"
    c = load_contracted_case(vectors, name)
    canonical_draw = vectors[f"{name}/spv_draw"]
    assert not np.array_equal(canonical_draw, c["spv_draw"])
    n_draws = int(np.frombuffer(c["draws"][:4].tobytes(), np.uint32)[0])
    cap_d, cap_c = c["caps"]
    g = {k: dev(torch, c[k]) for k in ("draws", "mesh_infos", "entities", "meshlets", "materials")}
    for profile, want in ((0, canonical_draw), (1, c["spv_draw"])):
        eng = Engine(0, max_entities=4096, max_dispatches=cap_d + 64, max_draws=cap_c + 64, arith_profile=profile)
        disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
        draw = torch.zeros(L.DRAW_HEADER + 28 * cap_c, dtype=torch.uint8, device="cuda")
        eng.entity_cull(c["ci"], g["draws"], g["mesh_infos"], disp, g["entities"], n_draws, cap_d)
        eng.meshlet_cull(c["ci"], disp, g["meshlets"], draw, g["entities"], g["materials"], cap_d, cap_c)
        torch.cuda.synchronize()
        eng.status()
        assert np.array_equal(host(draw)[:len(want)], want), f"profile {profile}"
        eng.close()
    from orbit_amd._lib import OrbitError

    with pytest.raises(OrbitError):
        Engine(0, arith_profile=2)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("occlusion_pass,ortho", [(0, False), (1, False), (2, False), (2, True), (0, True)])
def test_contracted_product_equals_the_contracted_oracle_on_larger_scenes(torch_mod, oracle, path, occlusion_pass, ortho):
    """3 000 entities with LOD chains, non-uniform scales and rotations; every pass, both projections."""
    torch = torch_mod
    scene = sc.make_scene(71 + occlusion_pass, 3000, meshlets_per_mesh=(1, 90), lods=2)
    cam = sc.default_camera(rot=(0.3, 0.2))
    W, H = 320, 180
    depth = sc.make_depth(71, W, H, cam)
    opyr, d = oracle.depth_reduce(depth, W, H)
    ps = (d.width, d.height)
    rng = np.random.default_rng(7)
    evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32) if occlusion_pass else None
    mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32) if occlusion_pass else None
    kw = {}
    if occlusion_pass == 2:
        kw = dict(p00=1.0 / 16, p11=1.0 / 16, z_near=0.7, z_far=61.3) if ortho else dict(p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
    ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=occlusion_pass, projection_type=1 if ortho else 0, **kw)
    pk = dict(pyr=opyr, psize=ps) if occlusion_pass == 2 else {}
    with oracle.arith_profile(1):
        from test_gpu_parity import run_oracle

        ref = run_oracle(oracle, scene, ci, evis, mvis, *( (opyr, ps) if occlusion_pass == 2 else ()))
    eng = make_engine(path, max_entities=8192, max_dispatches=40_000, max_draws=400_000)
    gs = GpuScene(torch, scene)
    recs, cmds = assert_same(run_gpu(torch, eng, gs, ci, evis, mvis, **pk), ref)
    assert len(recs) > 0
    eng.close()


def test_contracted_shard_cull_and_views(torch_mod, oracle):
    """The sharded engine's one-launch shard cull and orbit_cull_views under the contracted profile: the same routing."""
    torch = torch_mod
    from orbit_amd.engine import Engine

    scene = sc.make_scene(77, 5000, meshlets_per_mesh=(1, 90))
    cam = sc.default_camera(rot=(0.3, 0.2))
    ci = sc.make_cull_info(cam.view, cam.planes)
    gs = GpuScene(torch, scene)
    with oracle.arith_profile(1):
        from test_gpu_parity import run_oracle

        ref = run_oracle(oracle, scene, ci)
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    eng = Engine(0, max_entities=8192, max_dispatches=cap_d, max_draws=cap_c, arith_profile=1)
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
    rec = torch.zeros(L.VISIBLE_HEADER + 12 * cap_d, dtype=torch.uint8, device="cuda")
    cmd = torch.zeros(L.DRAW_HEADER + 28 * cap_c, dtype=torch.uint8, device="cuda")
    eng.cull_shard(ci, gs.draws, gs.mesh_infos, disp, gs.entities, 0, scene.entity_draw_count, cap_d, gs.meshlets, gs.materials,
                   rec, cap_d, draw_commands_buffer=cmd, draw_capacity=cap_c)
    torch.cuda.synchronize()
    eng.status()
    assert eng.shard_culls() == 1
    assert_same((host(disp), host(cmd), None, None), ref)
    disp2, cmd2 = torch.zeros_like(disp), torch.zeros_like(cmd)
    eng.cull_views([dict(cull_info=ci, entity_draw_buffer=gs.draws, mesh_info_buffer=gs.mesh_infos, meshlet_dispatch_buffer=disp2,
                         entity_buffer=gs.entities, entity_draw_count=scene.entity_draw_count, dispatch_capacity=cap_d,
                         meshlet_buffer=gs.meshlets, draw_commands_buffer=cmd2, material_buffer=gs.materials,
                         draw_capacity=cap_c)])
    torch.cuda.synchronize()
    eng.status()
    assert eng.fused_culls() == 1
    assert_same((host(disp2), host(cmd2), None, None), ref)
    eng.close()
