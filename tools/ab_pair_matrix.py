"""Meshlet-buffer evaluation (meshlet_eval_kernel<0,0,false,false,false>, config 5) for every pair (engine, copy of the
meshlet buffer) of one process, with the addresses of both — is the speed a property of the buffer, of the engine's
arena, or of the pair?  Needs a library build that prints its arena (tools/variants/abi_dbg.so: ORBIT_DEBUG_ARENA).
GPU box, repo root: python tools/ab_pair_matrix.py 2> arena.txt"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from orbit_amd import _lib, camera, layouts as L, synth
from orbit_amd.engine import Engine

os.environ["ORBIT_DEBUG_ARENA"] = "1"
lib = _lib.load_variant(os.path.join(ROOT, "tools/variants/abi_dbg.so"))
dev = torch.device("cuda", 0)
spec = synth.C5Spec(meshlets_per_entity=int(os.environ.get("M", 256)))
E, M = spec.entities, spec.meshlets_per_entity
draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
meshlets = synth.gen_meshlets(spec, 0, E, dev, half)
materials = synth.gen_materials(spec, dev)
ci = camera.frame_cull_info((0.0, 0.0, 1300.0))
disp_cap, draw_cap = E * spec.records_per_entity + 8, E * M // 2 + 1024
disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev)
caps = dict(max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap)


def engine():
    sys.stderr.flush()
    return Engine(0, _library=lib, **caps)


def timed(eng, buf, reps=7):
    out = []
    for _ in range(reps):
        eng.profile(True)
        eng.meshlet_cull(ci, disp, buf, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
        torch.cuda.synchronize()
        out.append(eng.profile_read()[0])
    return round(float(np.median(out[2:]) * 1e3), 1)


NE = int(os.environ.get("ENGINES", 5))
engs, bufs, pads = [engine()], [meshlets], []
engs[0].entity_cull(ci, draws, mesh, disp, ent, E, disp_cap)
torch.cuda.synchronize()
for k in range(3):
    pads.append(torch.empty((7 + 13 * k) << 20, dtype=torch.uint8, device=dev))
    bufs.append(meshlets.clone())
    if len(engs) < NE:
        engs.append(engine())
if len(engs) < NE:
    engs.append(engine())
for _ in range(300):
    engs[0].meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
torch.cuda.synchronize()
rows = {"meshlets_per_entity": M, "buffers": [hex(b.data_ptr()) for b in bufs], "disp": hex(disp.data_ptr()),
        "us[engine][buffer]": [[timed(e, b) for b in bufs] for e in engs]}
rows["again"] = [[timed(e, b) for b in bufs] for e in engs]
print(json.dumps(rows))
