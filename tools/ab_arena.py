"""Does orbit_ctx_create's placement probe (OrbitCaps.arena_candidates) pick the arena on which the strict drop-in's
evaluation (meshlet_eval_kernel<0,0,false,false,false>, SURVEY §8d's Meshlet-buffer path) is fast?  One process, config-5
scene: engines created with ONE candidate (the round-5 behaviour: whatever hipMalloc hands out) and engines created with
the probe (4 candidates each); per engine the probe's times and the evaluation's median (HIP events of the library,
every launch timed).  Run it in several fresh processes: `for i in 1 2 3 4 5; do python tools/ab_arena.py; done`.
GPU box, repo root.  ENGINES (per kind, default 3), REPS (default 10), CAND (default 4)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from orbit_amd import camera, layouts as L, synth
from orbit_amd.engine import Engine

N = int(os.environ.get("ENGINES", 3))
REPS = int(os.environ.get("REPS", 10))
CAND = int(os.environ.get("CAND", 4))
dev = torch.device("cuda", 0)
spec = synth.C5Spec()
E, M = spec.entities, spec.meshlets_per_entity
draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
meshlets = synth.gen_meshlets(spec, 0, E, dev, half)
materials = synth.gen_materials(spec, dev)
ci = camera.frame_cull_info((0.0, 0.0, 1300.0))
disp_cap, draw_cap = E * spec.records_per_entity + 8, E * M // 2 + 1024
disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev)


def timed(eng):
    out = []
    for _ in range(REPS):
        eng.profile(True)
        eng.meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
        torch.cuda.synchronize()
        out.append(eng.profile_read()[0])
    return float(np.median(out[2:]) * 1e3)


rows = []
for kind, cand in (("first", 1), ("probed", CAND)) * N:
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    eng = Engine(0, max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap, arena_candidates=cand)
    create_ms = (time.perf_counter() - t0) * 1e3
    eng.entity_cull(ci, draws, mesh, disp, ent, E, disp_cap)
    ms, chosen = eng.arena_probe()
    us = timed(eng)
    rows.append(dict(kind=kind, eval_us=round(us, 1), probe_us=[round(v * 1e3, 2) for v in ms], chosen=chosen,
                     create_ms=round(create_ms, 1)))
    print(json.dumps(rows[-1]), flush=True)
    eng.close()
f = [r["eval_us"] for r in rows if r["kind"] == "first"]
p = [r["eval_us"] for r in rows if r["kind"] == "probed"]
print(json.dumps(dict(first=f, probed=p, first_max=max(f), probed_max=max(p))))
