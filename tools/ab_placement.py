"""Does the PLACEMENT of a context's scratch (payload / masks / counts) relative to the meshlet buffer matter?
N engines of the same library in one process (each with its own arena, i.e. different addresses), the config-5 frame
through each in turn; prints per-engine eval medians and the arena addresses modulo a few powers of two."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from orbit_amd import camera, layouts as L, synth
from orbit_amd.engine import Engine

N = int(os.environ.get("ENGINES", 6))
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
engs, pads = [], []
for k in range(N):
    pads.append(torch.empty(int(os.environ.get("PAD_KB", 0)) * 1024 * (k + 1) + 1, dtype=torch.uint8, device=dev))  # perturbs the allocator
    engs.append(Engine(0, max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap))
for eng in engs:
    eng.entity_cull(ci, draws, mesh, disp, ent, E, disp_cap)
    for _ in range(3):
        eng.meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
torch.cuda.synchronize()
te = [[] for _ in engs]
for eng in engs:
    eng.profile(True)
for r in range(int(os.environ.get("REPS", 20))):
    for k, eng in enumerate(engs):
        eng.meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
        torch.cuda.synchronize()
        te[k].append(eng.profile_read()[0]); eng.profile(True)
print("meshlet buffer at 0x%x" % meshlets.data_ptr())
for k in range(N):
    print(f"engine {k}: eval median {np.median(te[k]):.4f} ms  (p10 {np.percentile(te[k], 10):.4f}, p90 {np.percentile(te[k], 90):.4f})")
