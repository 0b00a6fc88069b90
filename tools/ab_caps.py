"""BASELINE config 2's one-launch cull (tests/config_scenes.py) under the OrbitCaps that change the arithmetic or the record
size, alternating in one process: dispatch_size 32 / 64 / 128 and arith_profile canonical / contracted.
usage: python tools/ab_caps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import json
import numpy as np, torch
import config_scenes as cs
from orbit_amd.engine import Engine

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
cam, scene = cs.camera(), cs.config2_scene()
ci = cs.pass0_cull_info(cam)
bufs = [dev(a) for a in (scene.entity_draw_buffer(), scene.mesh_infos, scene.entities, scene.meshlets, scene.materials)]
cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
variants = [("S32 canonical", {}), ("S64", dict(dispatch_size=64)), ("S128", dict(dispatch_size=128)), ("contracted", dict(arith_profile=1))]
out = {}
for cull_path, label in ((2, "one_launch"), (0, "separate_launches")):
    runs = []
    for name, caps in variants:
        eng = Engine(0, max_entities=8192, max_dispatches=40_000, max_draws=200_000, cull_path=cull_path, **caps)
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
    out[label] = {"us": {n: round(float(np.median(t[10:])), 2) for (n, _), t in zip(variants, ts)},
                  "draws_identical_to_S32": {n: bool(torch.equal(runs[0][4], r[4])) for (n, _), r in zip(variants[1:], runs[1:])}}
print(json.dumps(out))
