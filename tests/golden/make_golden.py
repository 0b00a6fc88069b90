"""Generates tests/golden/*.npz: small seeded inputs + the oracle's outputs for every stage of
the path.  The reference holds no fixtures for this path and cannot run here,
so these are regression pins of the oracle (itself pinned by tests/np_restatement.py), not vectors
from the reference.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import scenes as sc  # noqa: E402
from oracle import oracle  # noqa: E402
from orbit_amd import layouts as L  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def cull_case(name, seed, n, lods, W, H):
    scene = sc.make_scene(seed, n, n_meshes=max(1, n // 3), lods=lods, extent=(35.0, 8.0, 35.0),
                          meshlet_layout="scattered")  # the layout the committed fixtures were made with
    cam = sc.default_camera(rot=(0.8, 0.6))
    rng = np.random.default_rng(seed)
    evis = rng.integers(0, 2 ** 32, (n + 31) // 32, dtype=np.uint32)
    mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    depth = sc.make_depth(seed, W, H, cam, n_occluders=24)
    pyr, d = oracle.depth_reduce(depth, W, H)
    cap_d, cap_c = scene.max_dispatches() + 4, scene.lod0_meshlets + 4
    out = dict(entity_draw_buffer=scene.entity_draw_buffer(), mesh_infos=scene.mesh_infos.view(np.uint8),
               entities=scene.entities.view(np.uint8), meshlets=scene.meshlets.view(np.uint8),
               materials=scene.materials.view(np.uint8), evis=evis, mvis=mvis, depth=depth, pyramid=pyr,
               pyramid_size=np.array([d.width, d.height], np.uint32), caps=np.array([cap_d, cap_c], np.uint32),
               entity_draw_count=np.uint32(n))
    for p in (0, 1, 2):
        ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=p, p00=cam.p00, p11=cam.p11, z_near=cam.z_near,
                               lod_base=8.0)
        disp, ev, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), n, scene.mesh_infos, scene.entities, cap_d,
                                         evis if p else None, pyr if p == 2 else None, (d.width, d.height))
        draw, mv, _ = oracle.meshlet_cull(ci, disp, scene.meshlets, cap_c, scene.entities, scene.materials,
                                          mvis if p else None, pyr if p == 2 else None, (d.width, d.height))
        nr, nd = int(disp[:4].view(np.uint32)[0]), int(draw[:4].view(np.uint32)[0])
        out[f"ci{p}"] = np.frombuffer(ci.tobytes(), dtype=np.uint8)
        out[f"disp{p}"], out[f"draw{p}"] = disp[:12 + 16 * nr], draw[:4 + 28 * nd]
        if p == 2:
            out["evis_out"], out["mvis_out"] = ev, mv
    np.savez_compressed(os.path.join(OUT, name), **out)


def cluster_case(name, seed, W, H, n_lights):
    cam = sc.default_camera(aspect=W / H)
    depth = sc.make_depth(seed, W, H, cam, n_occluders=20)
    cx, cy, cz = -(-W // 8), -(-H // 8), 32
    zs, zb = oracle.cluster_grid_info(cam.z_near, 200.0, cz)
    push = np.zeros((), dtype=L.MARK_ACTIVE_PUSH)
    push["cluster_count"], push["tile_size_px"], push["screen_size"] = (cx, cy, cz), 8, (W, H)
    push["z_near"], push["z_far"], push["z_scale"], push["z_bias"] = cam.z_near, 200.0, zs, zb
    push["depth_buffer_sample_count"] = 1
    info = np.zeros((), dtype=L.CLUSTER_CULL_INFO)
    info["world_to_view_matrix"] = sc.mat4_cols(cam.view)
    info["screen_to_view_matrix"] = sc.mat4_cols(np.linalg.inv(cam.proj.astype(np.float64)).astype(np.float32))
    info["cluster_count"], info["tile_size_px"], info["screen_size"] = (cx, cy, cz), 8, (W, H)
    info["z_near"], info["z_far"], info["global_light_count"] = cam.z_near, 200.0, n_lights
    lights = sc.make_lights(seed, n_lights, point_fraction=0.95)
    masks, bounds = oracle.cluster_mark(push, depth)
    unique, _ = oracle.cluster_compact((cx, cy, cz), masks, cx * cy * cz)
    na = int(unique[12:16].view(np.uint32)[0])
    lidx, img, _ = oracle.cluster_assign(info, unique, bounds, lights, na * 256 + 8, cx * cy * cz)
    nl = int(lidx[:4].view(np.uint32)[0])
    np.savez_compressed(os.path.join(OUT, name), push=np.frombuffer(push.tobytes(), np.uint8),
                        info=np.frombuffer(info.tobytes(), np.uint8), depth=depth, lights=lights.view(np.uint8),
                        masks=masks, bounds=bounds, unique=unique[:16 + 4 * na], light_indices=lidx[:4 + 4 * nl],
                        offset_image=img)


def config1_case(name):
    """BASELINE.json configs[0] / SURVEY.md §8d C1: 8192 spheres, centre ~ U[-128,128]^3 (app.rs:658), radius ~
    U[0.25,4], camera at (0,2,0) looking down -z, fov 90 deg, near 0.01, 16:9 (app.rs:744-752), planes =
    frustum_planes_from_matrix(projection)[0..5] (forward.rs:264,268), test dot(n,c)+d > -r (math.rs).  splitmix64
    streams of seed 69 (tests/scenes.py).  Stored: the inputs' defining parameters, the 256-word bitset, the count."""
    n = 8192
    i = np.arange(n)
    spheres = np.stack([sc.rnd_range(69, 1, i, -128, 128), sc.rnd_range(69, 2, i, -128, 128),
                        sc.rnd_range(69, 3, i, -128, 128), sc.rnd_range(69, 4, i, 0.25, 4.0)], axis=1)
    cam = sc.default_camera()
    bits, cnt = oracle.sphere_frustum_cull(sc.mat4_cols(cam.view), cam.planes, spheres)
    np.savez_compressed(os.path.join(OUT, name), view=sc.mat4_cols(cam.view), planes=cam.planes,
                        first_spheres=spheres[:16], sphere_sum=spheres.astype(np.float64).sum(axis=0),
                        bitset=bits, count=np.uint32(cnt))


if __name__ == "__main__":
    oracle.build()
    if len(sys.argv) > 1 and sys.argv[1] == "config1":  # add the C1 fixture without rewriting the others
        config1_case("config1_spheres.npz")
        sys.exit(0)
    config1_case("config1_spheres.npz")
    cull_case("cull_small.npz", 101, 180, 2, 160, 90)
    cull_case("cull_ragged.npz", 102, 45, 1, 96, 54)
    cluster_case("cluster_small.npz", 103, 160, 90, 120)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
