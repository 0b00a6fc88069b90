"""glTF -> `.orbitscene` -> the product on the GPU, bit-exact against the oracle (SURVEY.md §8f-3 end to end): a .glb
with five meshes, eight LOD levels each, three alpha modes and a node hierarchy goes through load_gltf + the LOD chain
(host mirror), SceneData::update_scene, the dump format, and then both cull stages — single pass and the two-pass HiZ
frame — with the LOD pick active (instances from 4 m to 420 m)."""
import importlib.util
import os

import numpy as np
import pytest

import scenes as sc
from orbit_amd import gltf, orbitscene
from orbit_amd import layouts as L

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_glb_to_orbitscene_to_both_cull_stages(tmp_path, oracle):
    import torch

    from orbit_amd.engine import Engine, depth_pyramid_desc

    spec = importlib.util.spec_from_file_location("make_test_glb", os.path.join(ROOT, "tools", "make_test_glb.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    glb = str(tmp_path / "scene.glb")
    tool.write(glb, instances=200)
    b = gltf.to_scene_buffers(gltf.load(glb))
    dump = str(tmp_path / "scene.orbitscene")
    orbitscene.write(dump, **{k: b[k] for k in ("entity_draws", "mesh_infos", "entities", "meshlets", "materials")})
    d = orbitscene.read(dump)
    n = int(d["entity_draws"][:4].view(np.uint32)[0])
    assert n > 150 and int(d["mesh_infos"]["lod_count"].min()) >= 2
    draws = d["entity_draws"][4:4 + 12 * n].view(L.ENTITY_DRAW)
    per_draw_max = d["mesh_infos"]["mesh_lods"][draws["mesh_index"]][:, :, 1].max(axis=1)
    cap_d, cap_c = int((per_draw_max // 32 + 1).sum()) + 8, int(per_draw_max.sum()) + 8
    vis_words = int(draws["visibility_offset"].max()) + int(per_draw_max.max()) // 32 + 2
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
    g = {k: dev(d[k]) for k in ("entity_draws", "mesh_infos", "entities", "meshlets", "materials")}
    eng = Engine(0, max_entities=n + 256, max_dispatches=cap_d, max_draws=cap_c)
    cam = sc.default_camera(position=(0.0, 1.0, 6.0))
    W, H = 640, 360
    depth = sc.make_depth(5, W, H, cam, n_occluders=12)
    pd = depth_pyramid_desc(W, H)
    ps = (pd.width, pd.height)
    gpyr = torch.zeros(pd.total_texels, dtype=torch.float32, device="cuda")
    eng.depth_reduce(dev(depth), W, H, gpyr)
    opyr, _ = oracle.depth_reduce(depth, W, H)
    rng = np.random.default_rng(5)
    evis0 = rng.integers(0, 2 ** 32, (n + 31) // 32, dtype=np.uint32)
    mvis0 = rng.integers(0, 2 ** 32, vis_words, dtype=np.uint32)
    lods_seen = set()
    for p in (0, 1, 2):
        ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=p, p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
        evis, mvis = (dev(evis0), dev(mvis0)) if p else (None, None)
        disp = torch.zeros(12 + 16 * cap_d, dtype=torch.uint8, device="cuda")
        draw = torch.zeros(4 + 28 * cap_c, dtype=torch.uint8, device="cuda")
        kw = dict(depth_pyramid=gpyr if p == 2 else None, depth_pyramid_size=ps if p == 2 else (0, 0))
        eng.entity_cull(ci, g["entity_draws"], g["mesh_infos"], disp, g["entities"], n, cap_d, visibility_buffer=evis, **kw)
        eng.meshlet_cull(ci, disp, g["meshlets"], draw, g["entities"], g["materials"], cap_d, cap_c,
                         meshlet_visibility_buffer=mvis, material_count=len(d["materials"]), **kw)
        torch.cuda.synchronize()
        eng.status()
        odisp, oev, dd = oracle.entity_cull(ci, d["entity_draws"], n, d["mesh_infos"], d["entities"], cap_d,
                                            evis0 if p else None, opyr if p == 2 else None, ps)
        odraw, omv, dc = oracle.meshlet_cull(ci, odisp, d["meshlets"], cap_c, d["entities"], d["materials"],
                                             mvis0 if p else None, opyr if p == 2 else None, ps)
        assert dd == 0 and dc == 0
        assert np.array_equal(disp.cpu().numpy(), odisp), f"pass {p}: dispatch records differ"
        nd = int(odraw[:4].view(np.uint32)[0])
        assert np.array_equal(draw.cpu().numpy()[:4 + 28 * nd], odraw[:4 + 28 * nd]), f"pass {p}: draw commands differ"
        if p == 2:
            assert np.array_equal(evis.cpu().numpy().view(np.uint32), oev)
            assert np.array_equal(mvis.cpu().numpy().view(np.uint32), omv)
        if p == 0:
            assert nd > 500
            _, recs = L.dispatch_buffer_records(odisp)
            for m in d["mesh_infos"]:
                for k, (off, cnt) in enumerate(m["mesh_lods"][:int(m["lod_count"])]):
                    if cnt and np.any((recs["meshlet_offset"] >= off) & (recs["meshlet_offset"] < off + cnt)):
                        lods_seen.add(k)
    assert len(lods_seen) >= 4, lods_seen  # the LOD pick of entity_cull.comp:204-209 walked the chain
    eng.close()
