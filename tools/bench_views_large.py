"""orbit_cull_views on a LARGE scene: four pass-0 views of a config-5-class scene (48 828 entities x 256 = 12.5 M
meshlets, all in view of every camera) in one call, from the Meshlet buffer and from the derived meshlet streams —
and the same four culls one after the other.  HIP-event medians."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from orbit_amd import camera, layouts as L, synth
from orbit_amd.engine import Engine

dev = torch.device("cuda", 0)
spec = synth.C5Spec(entities=int(os.environ.get("E", 48_828)))
E, M = spec.entities, spec.meshlets_per_entity
draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
meshlets = synth.gen_meshlets(spec, 0, E, dev, half)
materials = synth.gen_materials(spec, dev)
cis = [camera.frame_cull_info(p) for p in ((0.0, 0.0, 1300.0), (0.0, 0.0, 1500.0), (0.0, 0.0, 1800.0), (0.0, 0.0, 2200.0))]
disp_cap, draw_cap = E * spec.records_per_entity + 8, E * M // 2 + 1024
eng = Engine(0, max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap, max_views=len(cis))
views = []
for ci in cis:
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
    draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev)
    views.append(dict(cull_info=ci, entity_draw_buffer=draws, mesh_info_buffer=mesh, meshlet_dispatch_buffer=disp,
                      entity_buffer=ent, entity_draw_count=E, dispatch_capacity=disp_cap, meshlet_buffer=meshlets,
                      draw_commands_buffer=draw, material_buffer=materials, draw_capacity=draw_cap,
                      material_count=spec.materials))
arr, keep = eng.prepare_views(views)


def sequential():
    for v in views:
        eng.entity_cull(v["cull_info"], draws, mesh, v["meshlet_dispatch_buffer"], ent, E, disp_cap)
        eng.meshlet_cull(v["cull_info"], v["meshlet_dispatch_buffer"], meshlets, v["draw_commands_buffer"], ent, materials,
                         disp_cap, draw_cap, material_count=spec.materials)


def timeit(fn, reps=30, warm=40):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); z.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(z))
    return float(np.median(ts))


out = {"scene": f"{E} entities x {M} = {E * M} meshlets, {len(cis)} pass-0 views"}
out["buffer_one_after_the_other_ms"] = timeit(sequential)
out["buffer_cull_views_ms"] = timeit(lambda: eng.cull_views_prepared(arr))
ref = [v["draw_commands_buffer"].clone() for v in views]
ms = eng.meshlet_stream(meshlets, 0, E * M)
eng.bind_meshlet_stream(ms)
out["stream_one_after_the_other_ms"] = timeit(sequential)
out["stream_cull_views_ms"] = timeit(lambda: eng.cull_views_prepared(arr))
out["identical_outputs"] = all(torch.equal(r, v["draw_commands_buffer"]) for r, v in zip(ref, views))
out["survivors_per_view"] = [int(v["draw_commands_buffer"][:4].view(torch.int32).item()) for v in views]
print(json.dumps(out, indent=1))
