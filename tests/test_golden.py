"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle must
keep reproducing them on CPU, and the HIP engine must reproduce them on the GPU."""
import os

import numpy as np
import pytest

from orbit_amd import layouts as L

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CULL = ["cull_small.npz", "cull_ragged.npz"]


def load(name):
    return dict(np.load(os.path.join(GOLD, name)))


@pytest.mark.parametrize("name", CULL)
def test_oracle_reproduces_cull_fixture(oracle, name):
    g = load(name)
    n, (cap_d, cap_c), ps = int(g["entity_draw_count"]), g["caps"], tuple(int(v) for v in g["pyramid_size"])
    pyr, _ = oracle.depth_reduce(g["depth"], g["depth"].shape[1], g["depth"].shape[0])
    assert np.array_equal(pyr.view(np.uint32), g["pyramid"].view(np.uint32))
    for p in (0, 1, 2):
        ci = g[f"ci{p}"]
        disp, ev, _ = oracle.entity_cull(ci, g["entity_draw_buffer"], n, g["mesh_infos"], g["entities"], int(cap_d),
                                         g["evis"] if p else None, g["pyramid"] if p == 2 else None, ps)
        draw, mv, _ = oracle.meshlet_cull(ci, disp, g["meshlets"], int(cap_c), g["entities"], g["materials"],
                                          g["mvis"] if p else None, g["pyramid"] if p == 2 else None, ps)
        assert np.array_equal(disp[:len(g[f"disp{p}"])], g[f"disp{p}"])
        assert np.array_equal(draw[:len(g[f"draw{p}"])], g[f"draw{p}"])
        if p == 2:
            assert np.array_equal(ev, g["evis_out"]) and np.array_equal(mv, g["mvis_out"])


def test_oracle_reproduces_config1_fixture(oracle):
    """BASELINE.json configs[0]: 8k spheres, frustum-only cull on the CPU (plumbing, no GPU)."""
    import scenes as sc

    g = load("config1_spheres.npz")
    n = 8192
    i = np.arange(n)
    spheres = np.stack([sc.rnd_range(69, 1, i, -128, 128), sc.rnd_range(69, 2, i, -128, 128),
                        sc.rnd_range(69, 3, i, -128, 128), sc.rnd_range(69, 4, i, 0.25, 4.0)], axis=1)
    assert np.array_equal(spheres[:16], g["first_spheres"])  # the generator still makes the fixture's inputs
    assert np.array_equal(spheres.astype(np.float64).sum(axis=0), g["sphere_sum"])
    cam = sc.default_camera()
    assert np.array_equal(sc.mat4_cols(cam.view), g["view"]) and np.array_equal(cam.planes, g["planes"])
    bits, cnt = oracle.sphere_frustum_cull(g["view"], g["planes"], spheres)
    assert len(bits) == 256 and cnt == int(g["count"]) and np.array_equal(bits, g["bitset"])


@pytest.mark.parametrize("placement", ["model", "mesh"])
def test_oracle_entity_cull_reproduces_config1_fixture(oracle, placement):
    """Config 1 through the cull path itself: the fixture's spheres as one-meshlet entities
    through oracle entity_cull (entity_cull.comp:138-143).  The entities of the dispatch records are exactly the
    fixture's set bits; in pass 2 against an all-far pyramid the entity bitset written IS the fixture bitset."""
    import scenes as sc

    g = load("config1_spheres.npz")
    spheres = sc.config1_spheres()
    edb, mesh, ent, _, _ = sc.config1_scene(spheres, placement)
    n = len(spheres)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, g["planes"])
    disp, _, dropped = oracle.entity_cull(ci, edb, n, mesh, ent, n + 8)
    assert dropped == 0
    _, recs = L.dispatch_buffer_records(disp)
    want = np.flatnonzero(np.unpackbits(g["bitset"].view(np.uint8), bitorder="little")[:n])
    assert len(recs) == int(g["count"]) and np.array_equal(recs["entity_index"], want)
    assert bool((recs["meshlet_count"] == 1).all()) and np.array_equal(recs["meshlet_offset"], want)
    ci2 = sc.make_cull_info(cam.view, g["planes"], occlusion_pass=2, p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
    d = oracle.depth_pyramid_desc(64, 64)
    pyr = np.zeros(d.total_texels, np.float32)  # reverse-Z 0 = infinitely far: nothing is occluded
    _, ev, _ = oracle.entity_cull(ci2, edb, n, mesh, ent, n + 8, np.zeros(256, np.uint32), pyr, (d.width, d.height))
    assert np.array_equal(ev, g["bitset"])


def test_host_mirror_derives_config1_planes():
    """The C++ host mirror (camera.rs / math.rs restated) derives the fixture's five view-space planes."""
    from orbit_amd import passes

    g = load("config1_spheres.npz")
    proj = passes.projection_compute_matrix(passes.Projection.Perspective(float(np.float32(np.pi / 2)), 0.01), 16.0 / 9.0)
    planes = passes.frustum_planes_from_matrix(proj)[:5]
    assert np.array_equal(planes.view(np.uint32), np.ascontiguousarray(g["planes"], np.float32).view(np.uint32))


def test_oracle_reproduces_cluster_fixture(oracle):
    g = load("cluster_small.npz")
    cc = [int(v) for v in g["push"][:12].view(np.uint32)]
    masks, bounds = oracle.cluster_mark(g["push"], g["depth"])
    assert np.array_equal(masks, g["masks"]) and np.array_equal(bounds, g["bounds"])
    unique, _ = oracle.cluster_compact(cc, masks, cc[0] * cc[1] * cc[2])
    assert np.array_equal(unique[:len(g["unique"])], g["unique"])
    na = int(unique[12:16].view(np.uint32)[0])
    lidx, img, _ = oracle.cluster_assign(g["info"], unique, bounds, g["lights"], na * 256 + 8, cc[0] * cc[1] * cc[2])
    assert np.array_equal(lidx[:len(g["light_indices"])], g["light_indices"]) and np.array_equal(img, g["offset_image"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CULL)
def test_gpu_reproduces_cull_fixture(name):
    import torch

    from orbit_amd.engine import Engine

    g = load(name)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
    n, (cap_d, cap_c), ps = int(g["entity_draw_count"]), [int(v) for v in g["caps"]], tuple(int(v) for v in g["pyramid_size"])
    eng = Engine(0)
    H, W = g["depth"].shape
    gpyr = torch.zeros(len(g["pyramid"]), dtype=torch.float32, device="cuda")
    eng.depth_reduce(dev(g["depth"]), W, H, gpyr)
    torch.cuda.synchronize()
    assert np.array_equal(gpyr.cpu().numpy().view(np.uint32), g["pyramid"].view(np.uint32))
    bufs = {k: dev(g[k]) for k in ("entity_draw_buffer", "mesh_infos", "entities", "meshlets", "materials")}
    for p in (0, 1, 2):
        ci = g[f"ci{p}"]
        evis, mvis = (dev(g["evis"]), dev(g["mvis"])) if p else (None, None)
        disp = torch.zeros(12 + 16 * cap_d, dtype=torch.uint8, device="cuda")
        draw = torch.zeros(4 + 28 * cap_c, dtype=torch.uint8, device="cuda")
        kw = dict(depth_pyramid=gpyr if p == 2 else None, depth_pyramid_size=ps if p == 2 else (0, 0))
        eng.entity_cull(ci, bufs["entity_draw_buffer"], bufs["mesh_infos"], disp, bufs["entities"], n, cap_d,
                        visibility_buffer=evis, **kw)
        eng.meshlet_cull(ci, disp, bufs["meshlets"], draw, bufs["entities"], bufs["materials"], cap_d, cap_c,
                         meshlet_visibility_buffer=mvis, material_count=len(g["materials"]) // 80, **kw)
        torch.cuda.synchronize()
        assert np.array_equal(disp.cpu().numpy()[:len(g[f"disp{p}"])], g[f"disp{p}"])
        assert np.array_equal(draw.cpu().numpy()[:len(g[f"draw{p}"])], g[f"draw{p}"])
        if p == 2:
            assert np.array_equal(evis.cpu().numpy().view(np.uint32), g["evis_out"])
            assert np.array_equal(mvis.cpu().numpy().view(np.uint32), g["mvis_out"])
    eng.close()


@pytest.mark.gpu
def test_gpu_reproduces_cluster_fixture():
    import torch

    from orbit_amd.engine import Engine

    g = load("cluster_small.npz")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
    cc = [int(v) for v in g["push"][:12].view(np.uint32)]
    total = cc[0] * cc[1] * cc[2]
    eng = Engine(0)
    gm = torch.zeros(cc[0] * cc[1], dtype=torch.int32, device="cuda")
    gb = torch.zeros((total, 2), dtype=torch.int32, device="cuda")
    eng.cluster_mark(g["push"], dev(g["depth"]), gm, gb)
    gu = torch.zeros(16 + 4 * total, dtype=torch.uint8, device="cuda")
    eng.cluster_compact(cc, gm, gu, total)
    na = (len(g["unique"]) - 16) // 4
    gl = torch.zeros(4 + 4 * (na * 256 + 8), dtype=torch.uint8, device="cuda")
    gimg = torch.zeros((total, 2), dtype=torch.int32, device="cuda")
    eng.cluster_assign(g["info"], gu, gb, dev(g["lights"]), gl, na * 256 + 8, gimg)
    torch.cuda.synchronize()
    assert np.array_equal(gm.cpu().numpy().view(np.uint32), g["masks"])
    assert np.array_equal(gb.cpu().numpy().view(np.uint32), g["bounds"])
    assert np.array_equal(gu.cpu().numpy()[:len(g["unique"])], g["unique"])
    assert np.array_equal(gl.cpu().numpy()[:len(g["light_indices"])], g["light_indices"])
    assert np.array_equal(gimg.cpu().numpy().view(np.uint32), g["offset_image"])
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("placement", ["model", "mesh"])
def test_gpu_config1_through_entity_cull(placement):
    """BASELINE.json configs[0] through the PRODUCT: the fixture's 8192 spheres as one-meshlet
    entities through orbit_entity_cull + orbit_meshlet_cull.  The dispatch records name exactly the fixture's set bits
    (entity_cull.comp:138-143 with the planes of math.rs:71-89), the draw list holds one command per visible sphere
    (meshlet sphere == entity sphere, no cone), and in pass 2 against an all-far pyramid the entity-visibility words
    the device writes ARE the fixture's 256-word bitset."""
    import scenes as sc
    import torch

    from orbit_amd.engine import Engine, depth_pyramid_desc

    g = load("config1_spheres.npz")
    spheres = sc.config1_spheres()
    assert np.array_equal(spheres[:16], g["first_spheres"])
    edb, mesh, ent, meshlets, materials = sc.config1_scene(spheres, placement)
    n = len(spheres)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
    cam = sc.default_camera()
    want = np.flatnonzero(np.unpackbits(g["bitset"].view(np.uint8), bitorder="little")[:n])
    assert len(want) == int(g["count"])
    eng = Engine(0)
    b = [dev(a) for a in (edb, mesh, ent, meshlets, materials)]
    cap = n + 8
    disp = torch.zeros(12 + 16 * cap, dtype=torch.uint8, device="cuda")
    draw = torch.zeros(4 + 28 * cap, dtype=torch.uint8, device="cuda")
    ci = sc.make_cull_info(cam.view, g["planes"])
    eng.entity_cull(ci, b[0], b[1], disp, b[2], n, cap)
    eng.meshlet_cull(ci, disp, b[3], draw, b[2], b[4], cap, cap, material_count=1)
    torch.cuda.synchronize()
    eng.status()
    head, recs = L.dispatch_buffer_records(disp.cpu().numpy())
    assert list(head) == [len(want), 1, 1]
    assert np.array_equal(recs["entity_index"], want) and np.array_equal(recs["meshlet_offset"], want)
    assert bool((recs["meshlet_count"] == 1).all())
    nd, cmds = L.draw_buffer_commands(draw.cpu().numpy())
    assert nd == len(want) and np.array_equal(cmds["cmd_first_instance"], want)
    assert np.array_equal(cmds["meshlet_index"], want)
    # pass 2, nothing occluding: the device's entity bitset == the fixture's bitset, word for word
    d = depth_pyramid_desc(64, 64)
    pyr = torch.zeros(d.total_texels, dtype=torch.float32, device="cuda")
    evis = torch.zeros(256, dtype=torch.int32, device="cuda")
    ci2 = sc.make_cull_info(cam.view, g["planes"], occlusion_pass=2, p00=cam.p00, p11=cam.p11, z_near=cam.z_near,
                            meshlet_visibility=False)
    eng.entity_cull(ci2, b[0], b[1], disp, b[2], n, cap, visibility_buffer=evis, depth_pyramid=pyr,
                    depth_pyramid_size=(d.width, d.height))
    torch.cuda.synchronize()
    assert np.array_equal(evis.cpu().numpy().view(np.uint32), g["bitset"])
    eng.close()
