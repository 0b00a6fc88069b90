"""Host-side cost of enqueueing one config-5 frame (entity_cull + meshlet_cull) from Python, piece by piece."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from orbit_amd import camera, layouts as L, synth
from orbit_amd.engine import Engine
dev = torch.device("cuda", 0)
spec = synth.C5Spec(entities=int(os.environ.get("E", 195_313)))
E, M = spec.entities, spec.meshlets_per_entity
draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
meshlets = synth.gen_meshlets(spec, 0, E, dev, half)
materials = synth.gen_materials(spec, dev)
ci = camera.frame_cull_info((0.0, 0.0, 1300.0))
disp_cap, draw_cap = E * spec.records_per_entity + 8, E * M // 2 + 1024
eng = Engine(0, max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap)
disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(device=dev)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(200)]
N = 50
def t(name, f):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N): f(i)
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f"{name:50s} host {th / N * 1e6:8.1f} us/iter   total {tt / N * 1e6:8.1f} us/iter")
t("entity_cull only", lambda i: eng.entity_cull(ci, draws, mesh, disp, ent, E, disp_cap, stream=st))
t("meshlet_cull only", lambda i: eng.meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials, stream=st))
def both(i):
    eng.entity_cull(ci, draws, mesh, disp, ent, E, disp_cap, stream=st)
    eng.meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials, stream=st)
t("frame", both)
eng.profile(True)
t("frame, profile hook on", both)
eng.profile_read(); eng.profile(False)
def with_events(i):
    eng.entity_cull(ci, draws, mesh, disp, ent, E, disp_cap, stream=st)
    evs[2 * i].record(st)
    eng.meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials, stream=st)
    evs[2 * i + 1].record(st)
t("frame + 2 torch timing events", with_events)
def with_ctx(i):
    with torch.cuda.stream(st):
        both(i)
t("frame inside `with torch.cuda.stream`", with_ctx)
t("event record only", lambda i: evs[i].record(st))
e2 = torch.cuda.Event()
t("plain event record + wait_event", lambda i: (e2.record(st), st.wait_event(e2)))
