"""Frames in flight: the config-5 frame (entity_cull + meshlet_cull, pass 0) issued round-robin on F streams, each
with its own OrbitCtx (scratch), dispatch and draw buffers — the reference keeps frames in flight the same way
(per-frame transient buffers, context.rs:1275-1316).  Frame k+1's entity cull and the head of its evaluation overlap
frame k's scan / emit and the tail of its evaluation.  Prints ms per frame for F = 1, 2, 3 (same box, same process,
alternating) and checks that every slot's final draw list is byte-identical to slot 0's."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from orbit_amd import camera, layouts as L, synth
from orbit_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--entities", type=int, default=195_313)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--flights", default="1,2")
args = ap.parse_args()
dev = torch.device("cuda", 0)
spec = synth.C5Spec(entities=args.entities)
E, M = spec.entities, spec.meshlets_per_entity
draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
meshlets = synth.gen_meshlets(spec, 0, E, dev, half)
materials = synth.gen_materials(spec, dev)
ci = camera.frame_cull_info((0.0, 0.0, 1300.0))
disp_cap, draw_cap = E * spec.records_per_entity + 8, E * M // 2 + 1024
FMAX = max(int(f) for f in args.flights.split(","))
slots = []
for _ in range(FMAX):
    slots.append(dict(eng=Engine(0, max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap),
                      stream=torch.cuda.Stream(device=dev),
                      disp=torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev),
                      draw=torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev)))

if os.environ.get("STREAM", "1") == "1":  # cull from the derived meshlet streams (one stream, bound to every slot)
    ms = slots[0]["eng"].meshlet_stream(meshlets, 0, E * M)
    for s in slots:
        s["eng"].bind_meshlet_stream(ms)

for s in slots:
    s["cull_done"] = torch.cuda.Event()


def frame(s, prev, ordered):
    st = s["stream"]
    s["eng"].entity_cull(ci, draws, mesh, s["disp"], ent, E, disp_cap, stream=st)
    if ordered and prev is not s:
        st.wait_event(prev["cull_done"])  # this frame's meshlet cull after the previous frame's: only the entity cull overlaps
    s["eng"].meshlet_cull(ci, s["disp"], meshlets, s["draw"], ent, materials, disp_cap, draw_cap,
                          material_count=spec.materials, stream=st)
    if ordered:
        s["cull_done"].record(st)


def run(mode, steps):
    F, ordered = mode
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        frame(slots[i % F], slots[(i - 1) % F], ordered)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


out = {}
modes = [(int(f), o) for f in args.flights.split(",") for o in ((False, True) if int(f) > 1 else (False,))]
for m in modes:
    run(m, 6)
for rep in range(args.reps):
    for m in modes:
        out.setdefault(f"{m[0]}{' ordered' if m[1] else ''}", []).append(run(m, args.steps))
for s in slots:
    s["eng"].status()
ref = slots[0]["draw"]
n = int(ref[:4].view(torch.int32).item())
same = all(torch.equal(s["draw"][:4 + 28 * n], ref[:4 + 28 * n]) for s in slots)
print(json.dumps({"ms_per_frame": {str(k): [round(v, 4) for v in vs] for k, vs in out.items()}, "survivors": n,
                  "slots_identical": bool(same)}))
