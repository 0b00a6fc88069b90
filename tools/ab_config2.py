"""A/B of library builds on BASELINE config 2's one-launch cull (tests/config_scenes.py), alternating in one process:
usage: python tools/ab_config2.py tools/variants/a.so [b.so ...]   (the product library is always variant 0)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import config_scenes as cs
from orbit_amd import _lib
from orbit_amd.engine import Engine

paths = [None] + sys.argv[1:]
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
cam = cs.camera()
for scene_name, scene in (("config2 (400 entity-draws)", cs.config2_scene()),
                          ("one chunk (250 entity-draws)", __import__("scenes").make_scene(2, 250, n_meshes=250, meshlets_per_mesh=(4, 26), extent=(15.0, 6.0, 9.0)))):
    ci = cs.pass0_cull_info(cam)
    bufs = [dev(a) for a in (scene.entity_draw_buffer(), scene.mesh_infos, scene.entities, scene.meshlets, scene.materials)]
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    runs = []
    for p in paths:
        eng = Engine(0, _library=None if p is None else _lib.load_variant(os.path.abspath(p)), max_entities=8192,
                     max_dispatches=40_000, max_draws=200_000, cull_path=2)
        disp = torch.zeros(12 + 16 * cap_d, dtype=torch.uint8, device="cuda")
        draw = torch.zeros(4 + 28 * cap_c, dtype=torch.uint8, device="cuda")
        arr, keep = eng.prepare_views([dict(cull_info=ci, entity_draw_buffer=bufs[0], mesh_info_buffer=bufs[1], meshlet_dispatch_buffer=disp,
                                            entity_buffer=bufs[2], entity_draw_count=scene.entity_draw_count, dispatch_capacity=cap_d,
                                            meshlet_buffer=bufs[3], draw_commands_buffer=draw, material_buffer=bufs[4], draw_capacity=cap_c)])
        runs.append((eng, arr, keep, disp, draw))
    ts = [[] for _ in runs]
    for rep in range(60):
        for k, (eng, arr, *_rest) in enumerate(runs):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            eng.cull_views_prepared(arr); eng.cull_views_prepared(arr)
            a.record(); eng.cull_views_prepared(arr); b.record()
            torch.cuda.synchronize()
            ts[k].append(a.elapsed_time(b) * 1e3)
    same = all(torch.equal(runs[0][4], r[4]) and torch.equal(runs[0][3], r[3]) for r in runs[1:])
    print(scene_name, {("product" if p is None else os.path.basename(p)): round(float(np.median(t[10:])), 2) for p, t in zip(paths, ts)}, "us; outputs identical:", same)
