"""Why does the Meshlet-buffer evaluation (meshlet_eval_kernel<0,0,false,false,false>, the strict drop-in of SURVEY §8d)
run at 313-316 us in some processes and at ~340 us in others while the stream evaluation beside it does not move
?  In ONE process, alternating, config-5 scene:

  A  the same meshlet buffer through N engines (each engine's arena — payload, ballots, counts — is its own hipMalloc,
     i.e. its own physical placement);
  B  one engine, the meshlet buffer at N different places: copies allocated one after the other (the allocator hands
     out different physical pages), read at offsets of 0 / 256 B / 4 KiB / 64 KiB / 2 MiB into their allocation;
  C  the stream evaluation of the same engine and scene as the control that "does not move".

Prints eval medians (HIP events of the library around the evaluation launch, every launch timed) and the addresses
modulo 2 MiB / 1 GiB.  Run it several times (separate processes): `for i in 1 2 3; do python tools/ab_strict.py; done`.
GPU box, repo root."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from orbit_amd import _lib, camera, layouts as L, synth

if os.environ.get("LIB"):  # another build of the library (tools/mkvariant_any.sh), e.g. LIB=tools/variants/paystore2.so
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["LIB"])
    print("library", _lib.LIB_PATH)
from orbit_amd.engine import Engine

N = int(os.environ.get("ENGINES", 4))
REPS = int(os.environ.get("REPS", 12))
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
nbytes = meshlets.numel()


def addr(t):
    p = t.data_ptr()
    return f"0x{p:x} (mod 2 MiB {p % (2 << 20):#x}, mod 1 GiB {p % (1 << 30):#x})"


def timed(eng, buf, reps=REPS):
    out = []
    for _ in range(reps):
        eng.profile(True)
        eng.meshlet_cull(ci, disp, buf, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
        torch.cuda.synchronize()
        out.append(eng.profile_read()[0])
    return out


engs = [Engine(0, max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap) for _ in range(N)]
engs[0].entity_cull(ci, draws, mesh, disp, ent, E, disp_cap)
torch.cuda.synchronize()
# settle: a second of back-to-back frames (the power controller, the runtime's one long stall)
for _ in range(600):
    engs[0].meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
torch.cuda.synchronize()
print("meshlet buffer", addr(meshlets))
# ---- A: engines
res = [[] for _ in engs]
for r in range(3):
    for k, eng in enumerate(engs):
        res[k] += timed(eng, meshlets, REPS // 3 + 1)
for k in range(N):
    print(f"A engine {k}: eval median {np.median(res[k]) * 1e3:7.1f} us  (p10 {np.percentile(res[k], 10) * 1e3:7.1f}, p90 {np.percentile(res[k], 90) * 1e3:7.1f})")
# ---- B: placements of the meshlet buffer
offs = [0, 256, 4096, 65536, 2 << 20]
copies = []
for k, o in enumerate(offs):
    raw = torch.empty(nbytes + (4 << 20), dtype=torch.uint8, device=dev)
    v = raw[o:o + nbytes]
    v.copy_(meshlets)
    copies.append((raw, v))
torch.cuda.synchronize()
resb = [[] for _ in copies]
for r in range(3):
    for k, (_, v) in enumerate(copies):
        resb[k] += timed(engs[0], v, REPS // 3 + 1)
for k, (_, v) in enumerate(copies):
    print(f"B copy {k} at {addr(v)}: eval median {np.median(resb[k]) * 1e3:7.1f} us  (p10 {np.percentile(resb[k], 10) * 1e3:7.1f}, p90 {np.percentile(resb[k], 90) * 1e3:7.1f})")
del copies
# ---- C: the stream evaluation, same engine
ms = engs[0].meshlet_stream(meshlets, 0, E * M)
ms.set_materials(materials, spec.materials)
engs[0].bind_meshlet_stream(ms)
torch.cuda.synchronize()
for _ in range(50):
    engs[0].meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
rc = timed(engs[0], meshlets, REPS * 2)
print(f"C stream evaluation (engine 0): eval median {np.median(rc) * 1e3:7.1f} us  (p10 {np.percentile(rc, 10) * 1e3:7.1f}, p90 {np.percentile(rc, 90) * 1e3:7.1f})")
engs[0].bind_meshlet_stream(None)
rb = timed(engs[0], meshlets, REPS)
print(f"A' engine 0 again, Meshlet buffer: eval median {np.median(rb) * 1e3:7.1f} us")
