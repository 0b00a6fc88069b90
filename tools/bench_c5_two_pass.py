"""Secondary measurement (SURVEY.md §8d C5 "two-pass as secondary"): the 50 M-meshlet scene through the
full two-pass HiZ frame — pass-1 cull (previous frame's bits), depth_reduce (1080p), pass-2 cull — on one MI355X,
steady state (frame >= 1), HIP-event medians.  With --check the final visibility bitsets and both draw lists of the
last frame are compared with the CPU oracle on the same inputs (bit-exact)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes as sc
from orbit_amd import layouts as L, synth
from orbit_amd.engine import Engine, depth_pyramid_desc

ap = argparse.ArgumentParser()
ap.add_argument("--entities", type=int, default=195_313)
ap.add_argument("--frames", type=int, default=12)
ap.add_argument("--check", action="store_true")
ap.add_argument("--meshlet-source", choices=("stream", "buffer"), default="stream",
                help="passes 0 and 2 cull from the derived meshlet streams (default) or from the 32-B Meshlet buffer")
ap.add_argument("--wall-distance", type=float, default=950.0, help="occluder wall, metres in front of the camera")
ap.add_argument("--wall-fraction", type=float, default=0.6, help="part of the screen width the wall covers")
args = ap.parse_args()

dev = torch.device("cuda", 0)
spec = synth.C5Spec(entities=args.entities)
E, M = spec.entities, spec.meshlets_per_entity
draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
meshlets = synth.gen_meshlets(spec, 0, E, dev, half)
materials = synth.gen_materials(spec, dev)
cam = sc.default_camera(position=(0.0, 0.0, 1300.0))
W, H = 1920, 1080
# synthetic reverse-Z depth: a wall 950 m in front of the camera over the left 60 % of the screen, sky elsewhere
depth = torch.zeros(H, W, dtype=torch.float32, device=dev)
depth[:, : int(args.wall_fraction * W)] = cam.z_near / args.wall_distance
d = depth_pyramid_desc(W, H)
pyr = torch.zeros(d.total_texels, dtype=torch.float32, device=dev)
ps = (d.width, d.height)
rec = spec.records_per_entity
disp_cap, draw_cap = E * rec + 8, E * M // 2 + 1024
eng = Engine(0, max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap)
if args.meshlet_source == "stream":
    ms = eng.meshlet_stream(meshlets, 0, E * M)
    ms.set_materials(materials, spec.materials)  # the alpha classes: the evaluation reads no material index
    torch.cuda.synchronize()
    eng.bind_meshlet_stream(ms)
disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
draw1 = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev)
draw2 = torch.zeros_like(draw1)
evis = torch.zeros((E + 31) // 32, dtype=torch.int32, device=dev)
mvis = torch.zeros(E * spec.vis_words_per_entity, dtype=torch.int32, device=dev)
ci1 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=1)
ci2 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, p00=cam.p00, p11=cam.p11, z_near=cam.z_near)


def frame(evs=None):
    def mark(k):
        if evs is not None:
            evs[k].record()
    mark(0)
    eng.entity_cull(ci1, draws, mesh, disp, ent, E, disp_cap, visibility_buffer=evis)
    eng.meshlet_cull(ci1, disp, meshlets, draw1, ent, materials, disp_cap, draw_cap, meshlet_visibility_buffer=mvis,
                     material_count=spec.materials)
    mark(1)
    eng.depth_reduce(depth, W, H, pyr)
    mark(2)
    eng.entity_cull(ci2, draws, mesh, disp, ent, E, disp_cap, visibility_buffer=evis, depth_pyramid=pyr,
                    depth_pyramid_size=ps)
    eng.meshlet_cull(ci2, disp, meshlets, draw2, ent, materials, disp_cap, draw_cap, meshlet_visibility_buffer=mvis,
                     depth_pyramid=pyr, depth_pyramid_size=ps, material_count=spec.materials)
    mark(3)


prev_e = prev_m = None
ts = []
for f in range(args.frames):
    if f == args.frames - 1:  # inputs of the last frame, for the oracle
        prev_e, prev_m = evis.cpu().numpy().view(np.uint32).copy(), mvis.cpu().numpy().view(np.uint32).copy()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    frame(evs)
    torch.cuda.synchronize()
    if f >= 2:
        ts.append([evs[i].elapsed_time(evs[i + 1]) for i in range(3)] + [evs[0].elapsed_time(evs[3])])
eng.status()
ts = np.median(np.array(ts), axis=0)
n1, n2 = int(draw1[:4].view(torch.int32).item()), int(draw2[:4].view(torch.int32).item())
out = {"workload": f"config5 two-pass: {E} entities x {M} meshlets, 1080p depth (wall at {args.wall_distance:.0f} m over "
                   f"{100 * args.wall_fraction:.0f} % of the screen)",
       "meshlet_source": args.meshlet_source,
       "pass1_cull_ms": float(ts[0]), "depth_reduce_ms": float(ts[1]), "pass2_cull_ms": float(ts[2]),
       "frame_ms": float(ts[3]), "meshlets_per_s_frame": E * M / (float(ts[3]) * 1e-3),
       "pass1_draws": n1, "pass2_draws": n2,
       "visible_entities": int(np.unpackbits(evis.cpu().numpy().view(np.uint8)).sum()),
       "visible_meshlets": int(np.unpackbits(mvis.cpu().numpy().view(np.uint8)).sum())}
# mesh-shading path on the same scene: entity_cull + orbit_meshlet_task_cull, pass 0 (one 44-B record per dispatch
# record instead of the compacted draw list: no scan, no emit)
ci0 = sc.make_cull_info(cam.view, cam.planes)
task = torch.zeros(44 * disp_cap, dtype=torch.uint8, device=dev)
tt = []
for f in range(12):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    eng.entity_cull(ci0, draws, mesh, disp, ent, E, disp_cap)
    eng.meshlet_task_cull(ci0, disp, meshlets, task, ent, materials, disp_cap, material_count=spec.materials)
    b.record()
    torch.cuda.synchronize()
    if f >= 2:
        tt.append(a.elapsed_time(b))
out["task_path_pass0_ms"] = float(np.median(tt))
out["task_path_meshlets_per_s"] = E * M / (out["task_path_pass0_ms"] * 1e-3)
if args.check:
    from oracle import oracle
    t0 = time.time()
    h = lambda t: t.cpu().numpy()
    edb, mi, en, ml, mt = h(draws).view(np.uint8), h(mesh).view(np.uint8), h(ent).view(np.uint8), h(meshlets).view(np.uint8), h(materials)
    o_pyr, _ = oracle.depth_reduce(h(depth), W, H)
    assert np.array_equal(o_pyr.view(np.uint32), h(pyr).view(np.uint32)), "pyramid differs"
    od1, oe1, _ = oracle.entity_cull(ci1, edb, E, mi, en, disp_cap, prev_e)
    oc1, om1, _ = oracle.meshlet_cull(ci1, od1, ml, draw_cap, en, mt, prev_m)
    assert np.array_equal(h(draw1)[:4 + 28 * n1], oc1[:4 + 28 * n1]) and int(oc1[:4].view(np.uint32)[0]) == n1, "pass-1 draws differ"
    od2, oe2, _ = oracle.entity_cull(ci2, edb, E, mi, en, disp_cap, prev_e, o_pyr, ps)
    oc2, om2, _ = oracle.meshlet_cull(ci2, od2, ml, draw_cap, en, mt, prev_m, o_pyr, ps)
    assert int(oc2[:4].view(np.uint32)[0]) == n2 and np.array_equal(h(draw2)[:4 + 28 * n2], oc2[:4 + 28 * n2]), "pass-2 draws differ"
    assert np.array_equal(h(evis).view(np.uint32), oe2), "entity visibility differs"
    assert np.array_equal(h(mvis).view(np.uint32), om2), "meshlet visibility differs"
    od0, _, _ = oracle.entity_cull(ci0, edb, E, mi, en, disp_cap)
    ot0, _ = oracle.meshlet_task_cull(ci0, od0, ml, en, mt)
    assert np.array_equal(h(task)[:44 * len(ot0)], ot0.view(np.uint8).reshape(-1)), "task records differ"
    out["task_survivors"] = int(ot0["task_mesh_count"].sum())
    out["oracle_bit_identical"] = True
    out["oracle_seconds"] = time.time() - t0
print(json.dumps(out, indent=1))
eng.close()
