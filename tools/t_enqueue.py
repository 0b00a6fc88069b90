"""Host cost of enqueueing one sharded step's cull (entity_cull + meshlet_cull_visible_records through the ctypes
binding) on a scene small enough that the GPU is never the bottleneck: what bench.py's Python loop adds per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from orbit_amd import camera, layouts as L, synth
from orbit_amd.engine import Engine

dev = torch.device("cuda", 0)
spec = synth.C5Spec(entities=512)
E, M = spec.entities, spec.meshlets_per_entity
draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
meshlets = synth.gen_meshlets(spec, 0, E, dev, half)
materials = synth.gen_materials(spec, dev)
ci = camera.frame_cull_info((0.0, 0.0, 1300.0))
disp_cap = E * spec.records_per_entity + 8
eng = Engine(0, max_entities=E + 256, max_dispatches=disp_cap, max_draws=E * M)
disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
lst = torch.zeros(L.VISIBLE_HEADER + 12 * disp_cap, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(device=dev)
def step():
    eng.entity_cull(ci, draws, mesh, disp, ent, E, disp_cap, stream=st)
    eng.meshlet_cull_visible_records(ci, disp, meshlets, lst, ent, materials, disp_cap, disp_cap,
                                     material_count=spec.materials, stream=st)
for _ in range(200): step()
torch.cuda.synchronize()
N = 3000
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6 * (t1 - t0) / N:.1f} us per step (2 calls, 6 launches); with the device draining {1e6 * (t2 - t0) / N:.1f} us")
ev = [torch.cuda.Event() for _ in range(4)]
t0 = time.perf_counter()
for _ in range(N):
    ev[0].record(st); st.wait_event(ev[1]); ev[2].record(st); st.wait_event(ev[3])
t1 = time.perf_counter()
print(f"four event operations {1e6 * (t1 - t0) / N:.1f} us")
