"""What the Meshlet-buffer evaluation's speed (meshlet_eval_kernel<0,0,false,false,false>, BASELINE config 5) belongs to:
one process, one engine; every input buffer of the cull replaced in turn by a fresh copy (allocated while the old one is
still alive, so on other pages), then the engine itself, then the same again after an idle pause.  If nothing moves the
time inside a process while processes differ (profiles/r06_arena_vmm.txt: 310 ... 345 us), the speed is a state of the
process / device, not a placement.  GPU box, repo root: `for i in 1 2 3 4; do python tools/ab_process_mode.py; done`."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from orbit_amd import camera, layouts as L, synth
from orbit_amd.engine import Engine

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
caps = dict(max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap)
eng = Engine(0, **caps)
eng.entity_cull(ci, draws, mesh, disp, ent, E, disp_cap)
torch.cuda.synchronize()
b = dict(meshlets=meshlets, disp=disp, draw=draw, ent=ent, materials=materials)


def timed(reps=9):
    out = []
    for _ in range(reps):
        eng.profile(True)
        eng.meshlet_cull(ci, b["disp"], b["meshlets"], b["draw"], b["ent"], b["materials"], disp_cap, draw_cap,
                         material_count=spec.materials)
        torch.cuda.synchronize()
        out.append(eng.profile_read()[0])
    return round(float(np.median(out[2:]) * 1e3), 1)


for _ in range(300):  # settle
    eng.meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
torch.cuda.synchronize()
rows = [("start", timed())]
keep = []
for name in ("meshlets", "disp", "draw", "ent", "materials"):
    old = b[name]
    pad = torch.empty(int(np.random.default_rng(len(keep)).integers(1, 64)) << 20, dtype=torch.uint8, device=dev)  # shifts what comes next
    b[name] = old.clone()
    keep += [old, pad]
    rows.append((f"fresh {name}", timed()))
eng2 = Engine(0, **caps)
eng, old_eng = eng2, eng
rows.append(("fresh engine", timed()))
time.sleep(3.0)
rows.append(("after 3 s idle (cold clocks)", timed(5)))
for _ in range(300):
    eng.meshlet_cull(ci, b["disp"], b["meshlets"], b["draw"], b["ent"], b["materials"], disp_cap, draw_cap, material_count=spec.materials)
torch.cuda.synchronize()
rows.append(("settled again", timed()))
print(json.dumps(dict(rows)))
