"""perf triage: per-phase wave-cycle shares of meshlet_eval_kernel (pass 0; ORBIT_SP_DEBUG=8 selects the
phase-stamped build of the kernel, the stamps are read with the non-public orbit_debug_read_cycles)."""
import ctypes as C, os, sys
os.environ["ORBIT_SP_DEBUG"] = str(8 | int(os.environ.get("EXTRA_FLAGS", "0")))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _triage; _triage.use_triage_library()  # -DORBIT_TRIAGE build (bash tools/mktriage.sh)
import torch, numpy as np
import scenes as sc
from orbit_amd import synth, layouts as L, _lib
from orbit_amd.engine import Engine
spec = synth.C5Spec()
dev = torch.device("cuda", 0)
draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
ml = synth.gen_meshlets(spec, 0, spec.entities, dev, half)
mat = synth.gen_materials(spec, dev)
cam = sc.default_camera(position=(0.0, 0.0, 1300.0)); ci = sc.make_cull_info(cam.view, cam.planes)
E = spec.entities; dc = E * 8 + 8; cc = E * 128
eng = Engine(0, max_entities=E + 256, max_dispatches=dc, max_draws=cc)
disp = torch.zeros(12 + 16 * dc, dtype=torch.uint8, device=dev); draw = torch.zeros(4 + 28 * cc, dtype=torch.uint8, device=dev)
eng.entity_cull(ci, draws, mesh, disp, ent, E, dc)
lib = _lib.load(); out = (C.c_ulonglong * 8)()
for _ in range(3): eng.meshlet_cull(ci, disp, ml, draw, ent, mat, dc, cc)
lib.orbit_debug_read_cycles(eng._ctx, out)
n = 5
for _ in range(n): eng.meshlet_cull(ci, disp, ml, draw, ent, mat, dc, cc)
lib.orbit_debug_read_cycles(eng._ctx, out)
names = ["rows 0-1", "rows 2-3 + setup loads", "rows 4-5", "rows 6-7", "slab write + mask store"]
tot = sum(out[k] for k in range(5)); waves = out[7]
print("waves", waves // n, "cycles/wave", tot / waves)
for k in range(5): print(f"  {names[k]:20s} {100.0 * out[k] / tot:5.1f}%   {out[k] / waves / 100.0:8.2f} us/wave (100 MHz ticks?)")
