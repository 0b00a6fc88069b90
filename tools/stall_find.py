"""When does the one long stall of a process happen?  N frames of the config-5 cull enqueued back to back, an event
after every frame (recorded, not waited for), intervals between consecutive events afterwards.
usage (GPU box): [N=800] [E=195313] [PRE_SLEEP_MS=0] [SYNC_AT=110,328] python tools/stall_find.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes as sc
from orbit_amd import layouts as L, synth
from orbit_amd.engine import Engine
dev = torch.device("cuda", 0)
spec = synth.C5Spec(entities=int(os.environ.get("E", 195_313)))
E, M = spec.entities, spec.meshlets_per_entity
draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
meshlets = synth.gen_meshlets(spec, 0, E, dev, half)
materials = synth.gen_materials(spec, dev)
cam = sc.default_camera(position=(0.0, 0.0, 1300.0))
ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=0)
disp_cap, draw_cap = E * spec.records_per_entity + 8, E * M // 2 + 1024
eng = Engine(0, max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap)
if os.environ.get("SOURCE", "stream") == "stream":
    ms = eng.meshlet_stream(meshlets, 0, E * M); ms.set_materials(materials, spec.materials); eng.bind_meshlet_stream(ms)
disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev)
N = int(os.environ.get("N", 800))
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
torch.cuda.synchronize()
time.sleep(float(os.environ.get("PRE_SLEEP_MS", 0)) / 1e3)
# PRE_LAUNCHES=n: n tiny launches (+ an event record every eighth) before the frames — does the one long stall of the
# process depend on how many submissions it has made (then this pre-pays it), or on the frames' own work?
pre = int(os.environ.get("PRE_LAUNCHES", 0))
if pre:
    tiny = torch.zeros(64, device=dev)
    pe = torch.cuda.Event()
    tp0 = time.perf_counter()
    gaps = []
    last = tp0
    for i in range(pre):
        tiny.add_(1.0)
        if i % 8 == 0:
            pe.record()
        now = time.perf_counter()
        if now - last > 2e-3:
            gaps.append((i, (now - last) * 1e3))
        last = now
    torch.cuda.synchronize()
    print(f"pre-launches {pre}: {(time.perf_counter() - tp0) * 1e3:.1f} ms; host gaps > 2 ms: {[(i, round(g, 1)) for i, g in gaps]}")
# EVENTS_PER_FRAME (default 1; 0: host time stamps only, 2: two new timing events per frame), REUSE_EVENT=1: the SAME
# timing event re-recorded every frame, PRE_EVENTS=n: n timing events created and recorded once before the frames —
# which of these moves the stall?
epf = int(os.environ.get("EVENTS_PER_FRAME", 1))
reuse = os.environ.get("REUSE_EVENT") == "1"
pre_events = [torch.cuda.Event(enable_timing=True) for _ in range(int(os.environ.get("PRE_EVENTS", 0)))]
for ev in pre_events:
    ev.record()
torch.cuda.synchronize()
extra = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)] if epf == 2 else None
one = torch.cuda.Event(enable_timing=True)
t_host = []
evs[0].record()
sync_at = {int(x) for x in os.environ.get("SYNC_AT", "").split(",") if x}
for i in range(N):
    if i in sync_at:
        torch.cuda.synchronize()
    eng.entity_cull(ci, draws, mesh, disp, ent, E, disp_cap)
    eng.meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
    if reuse:
        one.record()
    elif epf >= 1:
        evs[i + 1].record()
        if epf == 2:
            extra[i].record()
    t_host.append(time.perf_counter())
if reuse or epf == 0:
    evs[N].record()
torch.cuda.synchronize()
if reuse or epf == 0:
    th = (np.array(t_host) - t_host[0]) * 1e3
    print(f"N {N}: device total {evs[0].elapsed_time(evs[N]):.1f} ms; host enqueue total {th[-1]:.1f} ms (no per-frame events)")
    hgap = np.diff(th)
    for j in np.nonzero(hgap > 2.0)[0][:20]:
        print(f"  host gap between enqueues: {hgap[j]:.2f} ms after frame {j} (host time {th[j]:.1f} ms)")
    sys.exit(0)
dt = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(N)])
cum = np.cumsum(dt)
th = (np.array(t_host) - t_host[0]) * 1e3
big = np.nonzero(dt > 3 * np.median(dt))[0]
print(f"N {N} median {np.median(dt):.4f} ms; total {cum[-1]:.1f} ms; host enqueue total {th[-1]:.1f} ms")
for i in big[:10]:
    print(f"  long frame {i}: {dt[i]:.2f} ms, at device time {cum[i] - dt[i]:.1f} ms; host was enqueuing frame {int(np.searchsorted(th, cum[i] - dt[i]))} then")
hgap = np.diff(th)
for j in np.nonzero(hgap > 2.0)[0][:20]:
    print(f"  host gap between enqueues: {hgap[j]:.2f} ms after frame {j} (host time {th[j]:.1f} ms)")
