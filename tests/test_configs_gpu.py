"""BASELINE.json configs 2, 3 and 4 on the EXACT workloads tools/bench_configs.py times (tests/config_scenes.py,
SURVEY.md §8d) — not scenes of similar size: every path the timing tool reports (launch chain from the Meshlet buffer,
from the derived streams, the one-launch cull) against the oracle, sequence-exact lists and bit-exact bitsets."""
import numpy as np
import pytest

import config_scenes as cs
from orbit_amd import layouts as L
from test_gpu_parity import GpuScene, assert_same, dev, host, run_oracle, torch_mod  # noqa: F401

pytestmark = pytest.mark.gpu

PATHS = ["launch_chain", "launch_chain_from_streams", "one_launch"]


def make_engine(path, torch, gs):
    from orbit_amd.engine import Engine

    eng = Engine(0, max_entities=8192, max_dispatches=40_000, max_draws=200_000, cull_path=2 if path == "one_launch" else 1)
    if path == "launch_chain_from_streams":
        ms = eng.meshlet_stream(gs.meshlets, 0, len(gs.scene.meshlets))
        ms.set_materials(gs.materials, len(gs.scene.materials))
        eng.bind_meshlet_stream(ms)
    return eng


def view_of(torch, gs, ci, evis=None, mvis=None, pyr=None, psize=(0, 0)):
    s = gs.scene
    cap_d, cap_c = s.max_dispatches() + 8, s.lod0_meshlets + 8
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
    draw = torch.zeros(L.DRAW_HEADER + 28 * cap_c, dtype=torch.uint8, device="cuda")
    v = dict(cull_info=ci, entity_draw_buffer=gs.draws, mesh_info_buffer=gs.mesh_infos, meshlet_dispatch_buffer=disp,
             entity_buffer=gs.entities, entity_draw_count=s.entity_draw_count, dispatch_capacity=cap_d,
             meshlet_buffer=gs.meshlets, draw_commands_buffer=draw, material_buffer=gs.materials, draw_capacity=cap_c,
             visibility_buffer=evis, meshlet_visibility_buffer=mvis, material_count=len(s.materials))
    if pyr is not None:
        v.update(depth_pyramid=pyr, depth_pyramid_size=psize)
    return v, disp, draw


@pytest.mark.parametrize("path", PATHS)
def test_config2_frustum_and_cone_pass0(torch_mod, oracle, path):
    torch = torch_mod
    scene, cam = cs.config2_scene(), cs.camera()
    assert scene.entity_draw_count == 400 and 5000 < scene.lod0_meshlets < 7500
    gs = GpuScene(torch, scene)
    ci = cs.pass0_cull_info(cam)
    eng = make_engine(path, torch, gs)
    v, disp, draw = view_of(torch, gs, ci)
    eng.cull_views([v])
    torch.cuda.synchronize()
    eng.status()
    assert eng.fused_culls() == (1 if path == "one_launch" else 0)
    assert eng.meshlet_stream_culls() == (1 if path == "launch_chain_from_streams" else 0)
    recs, cmds = assert_same((host(disp), host(draw), None, None), run_oracle(oracle, scene, ci))
    assert 0 < len(recs) < scene.max_dispatches() and 0 < len(cmds) < scene.lod0_meshlets
    eng.close()


@pytest.mark.parametrize("path", PATHS)
def test_config3_two_pass_frame_1080p_64_occluders(torch_mod, oracle, path):
    """pass 1 (bits = the oracle's frame-0 result) -> depth_reduce -> pass 2, exactly the sequence the tool times."""
    torch = torch_mod
    from orbit_amd.engine import depth_pyramid_desc

    scene, cam = cs.config3_scene(), cs.camera()
    assert scene.entity_draw_count == 2900 and 55_000 < scene.lod0_meshlets < 66_000
    W, H = cs.SCREEN
    depth = cs.config3_depth(cam)
    ev0, mv0 = cs.frame0_bits(oracle, scene, cam, depth)
    assert ev0.any() and mv0.any()
    gs = GpuScene(torch, scene)
    ci1, ci2 = cs.two_pass_cull_infos(cam)
    d = depth_pyramid_desc(W, H)
    ps = (d.width, d.height)
    eng = make_engine(path, torch, gs)
    evis, mvis = dev(torch, ev0), dev(torch, mv0)
    gpyr = torch.zeros(d.total_texels, dtype=torch.float32, device="cuda")
    v1, disp1, draw1 = view_of(torch, gs, ci1, evis, mvis)
    v2, disp2, draw2 = view_of(torch, gs, ci2, evis, mvis, gpyr, ps)
    eng.cull_views([v1])
    eng.depth_reduce(dev(torch, depth).view(torch.float32), W, H, gpyr)
    eng.cull_views([v2])
    torch.cuda.synchronize()
    eng.status()
    assert eng.fused_culls() == (2 if path == "one_launch" else 0)
    opyr, _ = oracle.depth_reduce(depth, W, H)
    assert np.array_equal(host(gpyr, np.uint32), opyr.view(np.uint32).reshape(-1)), "pyramid differs"
    r1 = run_oracle(oracle, scene, ci1, ev0, mv0)
    _, c1 = assert_same((host(disp1), host(draw1), None, None), r1[:2] + (None, None))
    r2 = run_oracle(oracle, scene, ci2, ev0, mv0, opyr, ps)
    _, c2 = assert_same((host(disp2), host(draw2), host(evis, np.uint32), host(mvis, np.uint32)), r2)
    # (a static camera: pass 2 finds nothing NEWLY visible behind frame 0's own bits — it still tests every record of
    # every entity against the pyramid and rewrites both bitsets, which is what the comparison above covers)
    assert len(c1) > 0 and int(host(disp2)[:4].view(np.uint32)[0]) > 0, (len(c1), len(c2))
    eng.close()


def test_config4_clusters_on_config3s_depth(torch_mod, oracle):
    """mark -> compact -> assign on C3's 1080p depth with 10 000 lights: masks, bounds, the compacted list, every
    cluster's (offset, count) and its ascending light indices."""
    torch = torch_mod
    from orbit_amd.engine import Engine

    cam = cs.camera()
    depth = cs.config3_depth(cam)
    push, info, lights = cs.config4_inputs(oracle, cam)
    cx, cy, cz = cs.CLUSTERS
    total = cx * cy * cz
    eng = Engine(0, max_lights=12_000, max_clusters=total)
    om, ob = oracle.cluster_mark(push, depth)
    cap = cx * cy * max(4, cz)
    ou, dropped = oracle.cluster_compact((cx, cy, cz), om, cap)
    assert dropped == 0
    lcap = total * 32  # cluster.rs:546-553
    ol, oimg, ldropped = oracle.cluster_assign(info, ou, ob, lights, lcap, total)
    assert ldropped == 0
    gm = torch.zeros(cx * cy, dtype=torch.int32, device="cuda")
    gb = torch.zeros((total, 2), dtype=torch.int32, device="cuda")
    gu = torch.zeros(L.COMPACT_HEADER + 4 * cap, dtype=torch.uint8, device="cuda")
    gl = torch.zeros(L.LIGHT_INDEX_HEADER + 4 * lcap, dtype=torch.uint8, device="cuda")
    gimg = torch.zeros((total, 2), dtype=torch.int32, device="cuda")
    eng.compute_clusters(push, info, dev(torch, depth).view(torch.float32), dev(torch, lights), gm, gb, gu, cap, gl, lcap, gimg)
    torch.cuda.synchronize()
    eng.status()
    assert np.array_equal(host(gm, np.uint32), om) and np.array_equal(host(gb, np.uint32).reshape(-1, 2), ob)
    n_active = int(ou[12:16].view(np.uint32)[0])
    assert n_active > 10_000
    assert np.array_equal(host(gu)[:16 + 4 * n_active], ou[:16 + 4 * n_active]), "compacted cluster list differs"
    n_idx = int(ol[:4].view(np.uint32)[0])
    assert n_idx > 0 and np.array_equal(host(gl)[:4 + 4 * n_idx], ol[:4 + 4 * n_idx]), "light index list differs"
    act = ou[16:16 + 4 * n_active].view(np.uint32)
    assert np.array_equal(host(gimg, np.uint32).reshape(-1, 2)[act], oimg[act]), "(offset, count) image differs"
    eng.close()
