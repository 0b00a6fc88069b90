"""Where the emit launch of a config-5 meshlet cull spends its time: wall-clock stamps of every workgroup's phases from
a -DORBIT_EMIT_STAMPS build (bash tools/mkvariant_any.sh emit_stamps meshlet_emit -DORBIT_EMIT_STAMPS), 100 MHz ticks.
usage (GPU box, repo root): [SOURCE=stream|buffer] [E=195313] python tools/emit_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes as sc
from orbit_amd import _lib, layouts as L, synth
from orbit_amd.engine import Engine

lib = _lib.load_variant(os.path.join(ROOT, "tools", "variants", os.environ.get("VARIANT", "emit_stamps") + ".so"))
lib.orbit_debug_read_wave_stamps.restype = C.c_int32
lib.orbit_debug_read_wave_stamps.argtypes = [C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
spec = synth.C5Spec(entities=int(os.environ.get("E", 195_313)))
E, M = spec.entities, spec.meshlets_per_entity
draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
meshlets = synth.gen_meshlets(spec, 0, E, dev, half)
materials = synth.gen_materials(spec, dev)
cam = sc.default_camera(position=(0.0, 0.0, 1300.0))
ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=0)
disp_cap, draw_cap = E * spec.records_per_entity + 8, E * M // 2 + 1024
eng = Engine(0, _library=lib, max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap)
if os.environ.get("SOURCE", "stream") == "stream":
    ms = eng.meshlet_stream(meshlets, 0, E * M)
    ms.set_materials(materials, spec.materials)
    eng.bind_meshlet_stream(ms)
disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev)
def frame():
    eng.entity_cull(ci, draws, mesh, disp, ent, E, disp_cap)
    eng.meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
for _ in range(5):
    frame()
torch.cuda.synchronize()
out = np.zeros(2 * 8192, dtype=np.uint64)
assert lib.orbit_debug_read_wave_stamps(eng._ctx, out.ctypes.data) == 0
st = out.reshape(2048, 8)[:, :5].astype(np.int64)
n = int((st[:, 0] != 0).sum())
st = st[:n]
us = (st - st[:, 0].min()) / 100.0
names = ["start", "scan duty done", "ramp issued", "sums seen", "tiles done"]
print(f"{n} workgroups stamped; us from the first workgroup's start")
for k, nm in enumerate(names):
    c = us[:, k]
    print(f"  {nm:16s} min {c.min():6.2f}  median {np.median(c):6.2f}  p90 {np.percentile(c, 90):6.2f}  max {c.max():6.2f}")
late = us[:, 0] > 5.0
print(f"  workgroups that started later than 5 us: {int(late.sum())} (indices {np.nonzero(late)[0][:6].tolist()} ..)")
for b in (0, 1, 47, 48, 100, 500, n - 1):
    print(f"  wg {b:4d}: " + " ".join(f"{v:6.2f}" for v in us[b]))
