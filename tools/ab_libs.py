"""In-process alternating A/B of two BUILDS of the library (the product and variants under tools/variants/, e.g. the
previous commit built with `git stash; make; cp orbit_amd/liborbit_cull.so tools/variants/prev.so; git stash pop`):
the same config-5 frame through an engine of each library in turn, HIP-event medians of the frame and of its parts.
usage: python tools/ab_libs.py tools/variants/prev.so [more.so ...]      (the product library is always variant 0)
Give it at least two variants and run both orders, with a copy of the product as a control: on some boxes the FIRST
engine of a process runs several per cent slower than the others whatever its code (profiles/r02_notes.md)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from orbit_amd import _lib, camera, layouts as L, synth
from orbit_amd.engine import Engine

paths = [None] + sys.argv[1:]
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
engs = [Engine(0, _library=None if p is None else _lib.load_variant(os.path.abspath(p)), max_entities=E + 256,
               max_dispatches=disp_cap, max_draws=draw_cap) for p in paths]
# STREAM=1: every engine culls from derived meshlet streams built by its own library; STREAM=odd / even: only the
# engines at odd / even positions do (same library twice = the stream path against the Meshlet-buffer path)
smode = os.environ.get("STREAM", "0")
with_stream = [smode == "1" or (smode == "odd" and k % 2 == 1) or (smode == "even" and k % 2 == 0) for k in range(len(engs))]
for eng, w in zip(engs, with_stream):
    if w:
        ms = eng.meshlet_stream(meshlets, 0, E * M)
        if os.environ.get("CLASSES", "0") == "1":  # the stream's alpha classes (evaluation without material indices)
            ms.set_materials(materials, spec.materials)
        eng.bind_meshlet_stream(ms)
# placement probes: TAIL_MB allocates (and keeps) a block behind the last engine's arrays; ORDER=reverse times the
# engines in reverse order of creation
tail = torch.empty(int(os.environ.get("TAIL_MB", 0)) << 20, dtype=torch.uint8, device=dev)
ref = None
for eng in engs:
    for _ in range(3):
        eng.entity_cull(ci, draws, mesh, disp, ent, E, disp_cap)
        eng.meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
    torch.cuda.synchronize()
    n = int(draw[:4].view(torch.int32).item())
    cur = draw[:4 + 28 * n].clone()
    assert ref is None or os.environ.get("CHECK", "1") == "0" or torch.equal(ref, cur), "builds disagree on the draw list"
    ref = cur
R = int(os.environ.get("REPS", 30))
tf, tm, te = [[] for _ in engs], [[] for _ in engs], [[] for _ in engs]
for eng in engs:
    eng.profile(True)
for r in range(R):
    for k, eng in (list(enumerate(engs))[::-1] if os.environ.get("ORDER") == "reverse" else enumerate(engs)):
        a, m, z = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record()
        eng.entity_cull(ci, draws, mesh, disp, ent, E, disp_cap)
        m.record()
        eng.meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
        z.record()
        torch.cuda.synchronize()
        tf[k].append(a.elapsed_time(z)); tm[k].append(m.elapsed_time(z))
        te[k].append(eng.profile_read()[0]); eng.profile(True)
for k, p in enumerate(paths):
    print(f"{('product' if p is None else os.path.basename(p)) + (' +stream' if with_stream[k] else ''):24s} frame {np.median(tf[k]):.4f} ms   meshlet_cull {np.median(tm[k]):.4f}   "
          f"eval {np.median(te[k]):.4f}   rest of meshlet_cull {np.median(np.array(tm[k]) - np.array(te[k])):.4f}   entity_cull {np.median(np.array(tf[k]) - np.array(tm[k])):.4f}")
