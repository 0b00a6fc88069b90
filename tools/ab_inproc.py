"""In-process alternating A/B of ORBIT_SP_DEBUG flag sets (the flags are read when a context is created): the same
config-5 frame through engine A and engine B in turn, HIP-event medians of the whole frame and of the dominant kernel.
Process-to-process and box-to-box spread (several %) never enters the comparison.
usage: python tools/ab_inproc.py 0 64 [more flag sets ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _triage; _triage.use_triage_library()  # -DORBIT_TRIAGE build (bash tools/mktriage.sh)
import numpy as np, torch
from orbit_amd import camera, layouts as L, synth
from orbit_amd.engine import Engine

flagsets = [int(a) for a in sys.argv[1:]] or [0, 64]
dev = torch.device("cuda", 0)
spec = synth.C5Spec(entities=int(os.environ.get("E", 195_313)))
E, M = spec.entities, spec.meshlets_per_entity
draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
meshlets = synth.gen_meshlets(spec, 0, E, dev, half)
materials = synth.gen_materials(spec, dev)
ci = camera.frame_cull_info((0.0, 0.0, 1300.0))
disp_cap, draw_cap = E * spec.records_per_entity + 8, E * M // 2 + 1024
disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev)
engs = []
for f in flagsets:
    os.environ["ORBIT_SP_DEBUG"] = str(f)
    engs.append(Engine(0, max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap))


def frame(eng):
    eng.entity_cull(ci, draws, mesh, disp, ent, E, disp_cap)
    eng.meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)


ref = None
for eng in engs:
    for _ in range(3):
        frame(eng)
    torch.cuda.synchronize()
    eng.status()
    n = int(draw[:4].view(torch.int32).item())
    cur = draw[:4 + 28 * n].clone()
    assert ref is None or torch.equal(ref, cur), "flag sets disagree on the draw list"
    ref = cur
R = int(os.environ.get("REPS", 30))
tf = [[] for _ in engs]
te = [[] for _ in engs]
for eng in engs:
    eng.profile(True)
for r in range(R):
    for k, eng in enumerate(engs):
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        frame(eng)
        z.record()
        torch.cuda.synchronize()
        tf[k].append(a.elapsed_time(z))
        ms, cnt = eng.profile_read()
        te[k].append(ms)
        eng.profile(True)
for k, f in enumerate(flagsets):
    print(f"flags {f:4d}: frame median {np.median(tf[k]):.4f} ms (p10 {np.percentile(tf[k], 10):.4f}, p90 {np.percentile(tf[k], 90):.4f})"
          f"   eval median {np.median(te[k]):.4f} ms   rest {np.median(np.array(tf[k]) - np.array(te[k])):.4f}")
