"""In-process alternating A/B of two builds of the library on the PASS-2 cull (occlusion_pass 2: HiZ test, bits
written) of the config-5 scene: the two-pass frame of tools/bench_c5_two_pass.py through an engine of each library in
turn, HIP-event medians of the late pass (entity_cull + meshlet_cull) and of the whole frame.
usage: [SOURCE=stream|buffer] python tools/ab_pass2.py tools/variants/<name>.so [...]   (the product is variant 0)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes as sc
from orbit_amd import _lib, layouts as L, synth
from orbit_amd.engine import Engine, depth_pyramid_desc

paths = [None] + sys.argv[1:]
dev = torch.device("cuda", 0)
spec = synth.C5Spec(entities=int(os.environ.get("E", 195_313)))
E, M = spec.entities, spec.meshlets_per_entity
draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
meshlets = synth.gen_meshlets(spec, 0, E, dev, half)
materials = synth.gen_materials(spec, dev)
cam = sc.default_camera(position=(0.0, 0.0, 1300.0))
W, H = 1920, 1080
depth = torch.zeros(H, W, dtype=torch.float32, device=dev)
depth[:, : int(0.6 * W)] = cam.z_near / 950.0
d = depth_pyramid_desc(W, H)
ps = (d.width, d.height)
disp_cap, draw_cap = E * spec.records_per_entity + 8, E * M // 2 + 1024
ci1 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=1)
ci2 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
stream = os.environ.get("SOURCE", "stream") == "stream"


class Side:
    def __init__(self, path):
        self.eng = Engine(0, _library=None if path is None else _lib.load_variant(os.path.abspath(path)),
                          max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap)
        if stream:
            ms = self.eng.meshlet_stream(meshlets, 0, E * M)
            ms.set_materials(materials, spec.materials)
            self.eng.bind_meshlet_stream(ms)
        self.disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
        self.draw1 = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev)
        self.draw2 = torch.zeros_like(self.draw1)
        self.evis = torch.zeros((E + 31) // 32, dtype=torch.int32, device=dev)
        self.mvis = torch.zeros(E * spec.vis_words_per_entity, dtype=torch.int32, device=dev)
        self.pyr = torch.zeros(d.total_texels, dtype=torch.float32, device=dev)

    def frame(self, evs=None):
        e = self.eng
        if evs: evs[0].record()
        e.entity_cull(ci1, draws, mesh, self.disp, ent, E, disp_cap, visibility_buffer=self.evis)
        e.meshlet_cull(ci1, self.disp, meshlets, self.draw1, ent, materials, disp_cap, draw_cap,
                       meshlet_visibility_buffer=self.mvis, material_count=spec.materials)
        e.depth_reduce(depth, W, H, self.pyr)
        if evs: evs[1].record()
        e.entity_cull(ci2, draws, mesh, self.disp, ent, E, disp_cap, visibility_buffer=self.evis, depth_pyramid=self.pyr,
                      depth_pyramid_size=ps)
        e.meshlet_cull(ci2, self.disp, meshlets, self.draw2, ent, materials, disp_cap, draw_cap,
                       meshlet_visibility_buffer=self.mvis, depth_pyramid=self.pyr, depth_pyramid_size=ps,
                       material_count=spec.materials)
        if evs: evs[2].record()


sides = [Side(p) for p in paths]
for s in sides:
    for _ in range(4):
        s.frame()
torch.cuda.synchronize()
assert all(torch.equal(s.mvis, sides[0].mvis) and torch.equal(s.evis, sides[0].evis) for s in sides), "builds disagree"
R = int(os.environ.get("REPS", 25))
t2, tf = [[] for _ in sides], [[] for _ in sides]
for r in range(R):
    for k, s in enumerate(sides):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        s.frame(evs)
        torch.cuda.synchronize()
        t2[k].append(evs[1].elapsed_time(evs[2])); tf[k].append(evs[0].elapsed_time(evs[2]))
for k, p in enumerate(paths):
    print(f"{'product' if p is None else os.path.basename(p):24s} pass-2 cull {np.median(t2[k]):.4f} ms   two-pass frame {np.median(tf[k]):.4f} ms")
