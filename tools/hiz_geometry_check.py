"""CPU: how conservative is the reference's HiZ test against EXACT depth?  A small scene of real triangle meshes is
rasterised in numpy (reverse-Z depth and the id of the meshlet that owns every pixel), the depth goes through the oracle's
depth_reduce, and every meshlet is put through the pass-2 occlusion test (oracle, all-zero bitsets so that "visible"
means "drawn").  A meshlet that owns a pixel is visible by construction; if the test rejects it, the test (pyramid
level choice, 2x2 footprint, the non-integer screen -> level-0 ratio) is not conservative there.
usage: python tools/hiz_geometry_check.py [--seed S] [--entities N] [--width W --height H]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import scenes as sc  # noqa: E402
from oracle import oracle  # noqa: E402
from orbit_amd import assets, layouts as L  # noqa: E402


def rasterise(tris_clip, tri_owner, W, H):
    """tris_clip: (n, 3, 4) clip-space corners (w > 0).  Returns (depth (H, W) f32 reverse-Z, nearest = largest;
    owner (H, W) int, -1 = background).  Pixel centres, no culling, top-left rule ignored (edges count as inside)."""
    depth = np.zeros((H, W), dtype=np.float64)
    owner = np.full((H, W), -1, dtype=np.int64)
    ndc = tris_clip[:, :, :3] / tris_clip[:, :, 3:4]
    sx = (ndc[:, :, 0] * 0.5 + 0.5) * W
    sy = (1.0 - (ndc[:, :, 1] * 0.5 + 0.5)) * H
    sz = ndc[:, :, 2]  # reverse-Z depth, affine in screen space
    for t in range(len(tris_clip)):
        x, y, z = sx[t], sy[t], sz[t]
        x0, x1 = int(max(np.floor(x.min()), 0)), int(min(np.ceil(x.max()), W - 1))
        y0, y1 = int(max(np.floor(y.min()), 0)), int(min(np.ceil(y.max()), H - 1))
        if x0 > x1 or y0 > y1:
            continue
        px, py = np.meshgrid(np.arange(x0, x1 + 1) + 0.5, np.arange(y0, y1 + 1) + 0.5)
        den = (y[1] - y[2]) * (x[0] - x[2]) + (x[2] - x[1]) * (y[0] - y[2])
        if abs(den) < 1e-12:
            continue
        l0 = ((y[1] - y[2]) * (px - x[2]) + (x[2] - x[1]) * (py - y[2])) / den
        l1 = ((y[2] - y[0]) * (px - x[2]) + (x[0] - x[2]) * (py - y[2])) / den
        l2 = 1.0 - l0 - l1
        inside = (l0 >= 0) & (l1 >= 0) & (l2 >= 0)
        if not inside.any():
            continue
        d = l0 * z[0] + l1 * z[1] + l2 * z[2]
        sub_d, sub_o = depth[y0:y1 + 1, x0:x1 + 1], owner[y0:y1 + 1, x0:x1 + 1]
        win = inside & (d > sub_d)
        sub_d[win] = d[win]
        sub_o[win] = tri_owner[t]
    return depth.astype(np.float32), owner


def build(seed, n_entities, spread=8.0, zfar=30.0):
    rng = np.random.default_rng(seed)
    meshes = [sc.uv_sphere(12, 24, 1.0), sc.torus(48, 12), sc.uv_sphere(8, 12, 0.5)]
    infos = np.zeros(len(meshes), dtype=L.MESH_INFO)
    all_m, all_d, geo = [], [], []
    for k, (pos, idx) in enumerate(meshes):
        idx = sc.outward_winding(pos, idx)
        m, d = assets.compute_meshlets(pos, idx, data_offset_base=sum(len(x) for x in all_d))
        mn, mx, sp = assets.compute_mesh_bounds(pos)
        infos["bounding_sphere"][k] = sp
        infos["lod_count"][k] = 1
        infos["mesh_lods"][k, 0] = (sum(len(x) for x in all_m), len(m))
        all_m.append(m)
        all_d.append(d)
        geo.append((pos, idx))
    meshlets, data = np.concatenate(all_m), np.concatenate(all_d)
    ents = np.zeros(n_entities, dtype=L.ENTITY_DATA)
    draws = np.zeros(n_entities, dtype=L.ENTITY_DRAW)
    models, words = [], 0
    for i in range(n_entities):
        ang, k = rng.uniform(0, 2 * np.pi), rng.choice([0.5, 1.0])
        c, s = np.cos(ang), np.sin(ang)
        model = np.eye(4, dtype=np.float32)
        model[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float32) * np.float32(k)
        model[:3, 3] = (rng.uniform(-spread, spread), rng.uniform(-0.4 * spread, 0.6 * spread), rng.uniform(-zfar, -3.0))  # in front of the camera (-z)
        mesh = int(rng.integers(len(meshes)))
        ents["model_matrix"][i] = sc.mat4_cols(model)
        ents["normal_matrix"][i] = np.eye(4, dtype=np.float32).reshape(16)
        draws[i] = (i, mesh, words)
        words += (int(infos["mesh_lods"][mesh, 0, 1]) + 31) // 32
        models.append(model)
    counts = infos["mesh_lods"][draws["mesh_index"], 0, 1].astype(np.int64)
    mats = np.zeros(1, dtype=L.MATERIAL)
    scene = sc.Scene(draws, ents, infos, meshlets, mats, words, int(counts.sum()))
    return scene, data, geo, models


def check(seed=1, n_entities=40, W=320, H=180, spread=8.0, zfar=30.0):
    scene, data, geo, models = build(seed, n_entities, spread, zfar)
    cam = sc.default_camera(position=(0.0, 1.0, 2.0), aspect=W / H)
    vp = cam.proj.astype(np.float64) @ cam.view.astype(np.float64)
    tris, owners = [], []
    for e, model in enumerate(models):
        mesh = int(scene.entity_draws["mesh_index"][e])
        pos, _ = geo[mesh]
        world = np.c_[pos.astype(np.float64) @ model[:3, :3].astype(np.float64).T + model[:3, 3], np.ones(len(pos))]
        clip = world @ vp.T
        first, count = (int(v) for v in scene.mesh_infos["mesh_lods"][mesh, 0])
        for m in range(first, first + count):
            t = assets.meshlet_triangles(scene.meshlets[m], data)
            tris.append(clip[t])
            owners.append(np.full(len(t), e * 100000 + m))  # (entity, meshlet)
    tris, owners = np.concatenate(tris), np.concatenate(owners)
    assert (tris[:, :, 3] > cam.z_near).all(), "scene must be wholly in front of the near plane"
    depth, owner = rasterise(tris, owners, W, H)
    visible = set(int(v) for v in np.unique(owner) if v >= 0)
    opyr, d = oracle.depth_reduce(depth, W, H)
    ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, alpha_mode_flag=L.ALPHA_ALL, p00=cam.p00, p11=cam.p11,
                           z_near=cam.z_near)
    evis = np.zeros((scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = np.zeros(scene.vis_words, dtype=np.uint32)
    disp, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                    scene.entities, scene.max_dispatches() + 8, evis, opyr, (d.width, d.height))
    draw, _, _ = oracle.meshlet_cull(ci, disp, scene.meshlets, scene.lod0_meshlets + 8, scene.entities, scene.materials,
                                     mvis, opyr, (d.width, d.height))
    _, cmds = L.draw_buffer_commands(draw)
    drawn = set(int(e) * 100000 + int(m) for e, m in zip(cmds["cmd_first_instance"], cmds["meshlet_index"]))
    lost = sorted(visible - drawn)
    return dict(pixels_covered=int((owner >= 0).sum()), instances=len(set(owners.tolist())), visible=len(visible),
                drawn=len(drawn), visible_but_rejected=len(lost), examples=lost[:5])


def frame_sequence(seed=3, n_entities=40, W=240, H=135, spread=3.0, zfar=14.0, frames=3):
    """The two-pass protocol over a camera path, judged by the picture: per frame pass 1 redraws what last frame's
    bits say, its depth (rasterised from exactly those meshlets) feeds depth_reduce, pass 2 adds what became visible
    and rewrites the bits (forward.rs:213-430).  Rasterising the union of the two draw lists must give the same depth
    image as rasterising everything — culling may not change the picture."""
    scene, data, geo, models = build(seed, n_entities, spread, zfar)
    evis = np.zeros((scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = np.zeros(scene.vis_words, dtype=np.uint32)
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    report = []
    for f in range(frames):
        ang = 0.12 * f  # the camera turns and strafes a little every frame
        cam = sc.default_camera(position=(0.6 * f, 1.0, 2.0), rot=(np.cos(ang), np.sin(ang)), aspect=W / H)
        vp = cam.proj.astype(np.float64) @ cam.view.astype(np.float64)
        tris, owners = {}, []
        for e, model in enumerate(models):
            mesh = int(scene.entity_draws["mesh_index"][e])
            pos, _ = geo[mesh]
            world = np.c_[pos.astype(np.float64) @ model[:3, :3].astype(np.float64).T + model[:3, 3], np.ones(len(pos))]
            clip = world @ vp.T
            first, count = (int(v) for v in scene.mesh_infos["mesh_lods"][mesh, 0])
            for m in range(first, first + count):
                tris[(e, m)] = clip[assets.meshlet_triangles(scene.meshlets[m], data)]
        assert all((t[:, :, 3] > cam.z_near).all() for t in tris.values()), "geometry must stay in front of the near plane"

        def raster(keys):
            keys = sorted(keys)
            if not keys:
                return np.zeros((H, W), dtype=np.float32)
            tt = np.concatenate([tris[k] for k in keys])
            oo = np.concatenate([np.full(len(tris[k]), i) for i, k in enumerate(keys)])
            return rasterise(tt, oo, W, H)[0]

        def cull(occlusion_pass, pyr=None, psize=(0, 0)):
            nonlocal evis, mvis
            kw = dict(p00=cam.p00, p11=cam.p11, z_near=cam.z_near) if occlusion_pass == 2 else {}
            ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=occlusion_pass, alpha_mode_flag=L.ALPHA_ALL, **kw)
            disp, e_out, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                                scene.entities, cap_d, evis, pyr, psize)
            draw, m_out, _ = oracle.meshlet_cull(ci, disp, scene.meshlets, cap_c, scene.entities, scene.materials, mvis,
                                                 pyr, psize)
            if occlusion_pass == 2:
                evis, mvis = e_out, m_out
            _, cmds = L.draw_buffer_commands(draw)
            return set((int(e), int(m)) for e, m in zip(cmds["cmd_first_instance"], cmds["meshlet_index"]))

        drawn1 = cull(1)
        depth1 = raster(drawn1)
        opyr, d = oracle.depth_reduce(depth1, W, H)
        drawn2 = cull(2, opyr, (d.width, d.height))
        assert not (drawn1 & drawn2), "pass 2 redrew something pass 1 had drawn"
        picture, reference = raster(drawn1 | drawn2), raster(tris.keys())
        report.append(dict(frame=f, instances=len(tris), pass1=len(drawn1), pass2=len(drawn2),
                           wrong_pixels=int((picture.view(np.uint32) != reference.view(np.uint32)).sum())))
    return report


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--entities", type=int, default=40)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--height", type=int, default=180)
    ap.add_argument("--spread", type=float, default=8.0, help="half-width of the region the entities fill (smaller = more overlap)")
    ap.add_argument("--zfar", type=float, default=30.0, help="entities sit between 3 and zfar metres in front of the camera")
    a = ap.parse_args()
    oracle.build()
    print(check(a.seed, a.entities, a.width, a.height, a.spread, a.zfar))
    for row in frame_sequence(a.seed):
        print(row)
