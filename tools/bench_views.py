"""A frame's independent culls (early forward pass + four shadow cascades, config-3-class scene) issued one after the
other on one stream vs side by side through orbit_cull_views; and the frame's pyramids (main view + 4 cascades)
through orbit_depth_reduce one by one vs orbit_depth_reduce_multi.  HIP-event medians, eager launches."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes as sc
from orbit_amd import layouts as L, passes
from orbit_amd.engine import Engine, depth_pyramid_desc

scene = sc.make_scene(3, 2900, n_meshes=2900, meshlets_per_mesh=(4, 38), extent=(75.0, 20.0, 75.0))
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
g = dict(draws=dev(scene.entity_draw_buffer()), mesh=dev(scene.mesh_infos), ent=dev(scene.entities),
         mlt=dev(scene.meshlets), mat=dev(scene.materials))
cam = sc.default_camera()
cis = [sc.make_cull_info(cam.view, cam.planes)]
for c in range(4):
    h, _, _ = passes.shadow_cascade(direction=(-0.45, 0.2, 0.1, 0.86), camera_position=(0.0, 2.0, 0.0),
                                    camera_orientation=(0.0, 0.0, 0.0, 1.0), camera_fov=float(cam.fov),
                                    camera_near_clip=cam.z_near, camera_aspect_ratio=cam.aspect, cascade_index=c,
                                    max_shadow_distance=80.0)
    cis.append(passes.cull_info_to_gpu(h))
eng = Engine(0)
disp_cap, draw_cap = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
views = []
for ci in cis:
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device="cuda")
    draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device="cuda")
    views.append(dict(cull_info=ci, entity_draw_buffer=g["draws"], mesh_info_buffer=g["mesh"], meshlet_dispatch_buffer=disp,
                      entity_buffer=g["ent"], entity_draw_count=scene.entity_draw_count, dispatch_capacity=disp_cap,
                      meshlet_buffer=g["mlt"], draw_commands_buffer=draw, material_buffer=g["mat"], draw_capacity=draw_cap,
                      material_count=len(scene.materials)))


def sequential():
    for v in views:
        eng.entity_cull(v["cull_info"], g["draws"], g["mesh"], v["meshlet_dispatch_buffer"], g["ent"],
                        scene.entity_draw_count, disp_cap)
        eng.meshlet_cull(v["cull_info"], v["meshlet_dispatch_buffer"], g["mlt"], v["draw_commands_buffer"], g["ent"],
                         g["mat"], disp_cap, draw_cap, material_count=len(scene.materials))


def timeit(fn, reps=40, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); z.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(z))
    return float(np.median(ts))


out = {"scene": f"{scene.entity_draw_count} entity draws, {len(scene.meshlets)} meshlets, 5 views (1 perspective + 4 cascades)"}
out["culls_one_after_the_other_ms"] = timeit(sequential)
ref = [v["draw_commands_buffer"].clone() for v in views]
arr, keep = eng.prepare_views(views)
# the library's choice for views of this size: the one-launch cull (one launch per projection type: 2 for the 5 views)
out["orbit_cull_views_ms"] = timeit(lambda: eng.cull_views_prepared(arr))
out["orbit_cull_views_fused_views"] = eng.fused_culls()
# ... and the launch chain (cull_path = 1: one entity launch, 2 evaluations, scan, emit for the 5 views)
eng_chain = Engine(0, cull_path=1)
out["orbit_cull_views_launch_chain_ms"] = timeit(lambda: eng_chain.cull_views_prepared(arr))
# the same two ways as replayed HIP graphs (no host in the loop)
def graph_of(fn):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            fn()
    return gr
try:
    g_seq, g_views = graph_of(sequential), graph_of(lambda: eng.cull_views_prepared(arr))
    out["graph_replay_one_after_the_other_ms"] = timeit(g_seq.replay)
    out["graph_replay_cull_views_ms"] = timeit(g_views.replay)
except Exception as ex:  # graph capture of library launches is optional here
    out["graph_replay"] = f"not available: {ex}"
out["identical_outputs"] = all(torch.equal(r, v["draw_commands_buffer"]) for r, v in zip(ref, views))
# pyramids: 1080p main view + four 2048^2 cascades
items = []
for (W, H) in [(1920, 1080)] + [(2048, 2048)] * 4:
    d = depth_pyramid_desc(W, H)
    items.append(dict(depth=torch.rand(H, W, device="cuda"), width=W, height=H,
                      pyramid=torch.zeros(d.total_texels, dtype=torch.float32, device="cuda")))
out["depth_reduce_x5_ms"] = timeit(lambda: [eng.depth_reduce(i["depth"], i["width"], i["height"], i["pyramid"]) for i in items])
out["depth_reduce_multi_ms"] = timeit(lambda: eng.depth_reduce_multi(items))
# the late half of a frame: the pyramids of three views + their pass-2 (HiZ) culls — view by view (3 x (2 + 5)
# launches) against one chain (orbit_depth_reduce_multi + orbit_cull_views: 2 + 6 launches)
screens = [(1920, 1080), (1024, 1024), (1024, 1024)]
ortho = dict(projection_type=1, p00=1.0 / 60, p11=1.0 / 60, z_near=0.5, z_far=200.0)
lcis = [sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, p00=cam.p00, p11=cam.p11, z_near=cam.z_near),
        sc.make_cull_info(cam.view, cam.planes[:4], occlusion_pass=2, **ortho),
        sc.make_cull_info(sc.translation(3.0, -1.0, -20.0), cam.planes[:2], occlusion_pass=2, **ortho)]
rng = np.random.default_rng(5)
late, litems = [], []
for ci, (W, H) in zip(lcis, screens):
    d = depth_pyramid_desc(W, H)
    depth = (torch.rand(H, W, device="cuda") * 0.02)
    depth[:, : W // 3] = 0.35
    pyr = torch.zeros(d.total_texels, dtype=torch.float32, device="cuda")
    litems.append(dict(depth=depth, width=W, height=H, pyramid=pyr))
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device="cuda")
    draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device="cuda")
    ev = dev(rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32))
    mv = dev(rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32))
    late.append(dict(cull_info=ci, entity_draw_buffer=g["draws"], mesh_info_buffer=g["mesh"], meshlet_dispatch_buffer=disp,
                     entity_buffer=g["ent"], entity_draw_count=scene.entity_draw_count, dispatch_capacity=disp_cap,
                     meshlet_buffer=g["mlt"], draw_commands_buffer=draw, material_buffer=g["mat"], draw_capacity=draw_cap,
                     material_count=len(scene.materials), visibility_buffer=ev, meshlet_visibility_buffer=mv,
                     depth_pyramid=pyr, depth_pyramid_size=(d.width, d.height), _bits=(ev.clone(), mv.clone())))


def reset_bits():
    for v in late:
        v["visibility_buffer"].copy_(v["_bits"][0]); v["meshlet_visibility_buffer"].copy_(v["_bits"][1])


def late_sequential():
    reset_bits()
    for v, it in zip(late, litems):
        eng.depth_reduce(it["depth"], it["width"], it["height"], it["pyramid"])
        kw = dict(depth_pyramid=v["depth_pyramid"], depth_pyramid_size=v["depth_pyramid_size"])
        eng.entity_cull(v["cull_info"], g["draws"], g["mesh"], v["meshlet_dispatch_buffer"], g["ent"],
                        scene.entity_draw_count, disp_cap, visibility_buffer=v["visibility_buffer"], **kw)
        eng.meshlet_cull(v["cull_info"], v["meshlet_dispatch_buffer"], g["mlt"], v["draw_commands_buffer"], g["ent"],
                         g["mat"], disp_cap, draw_cap, material_count=len(scene.materials),
                         meshlet_visibility_buffer=v["meshlet_visibility_buffer"], **kw)


larr, lkeep = eng.prepare_views([{k: x for k, x in v.items() if k != "_bits"} for v in late])


def late_chain():
    reset_bits()
    eng.depth_reduce_multi(litems)
    eng.cull_views_prepared(larr)


out["late_pass_3_views_one_after_the_other_ms"] = timeit(late_sequential)
lref = [v["draw_commands_buffer"].clone() for v in late]
out["late_pass_3_views_one_chain_ms"] = timeit(late_chain)
out["late_pass_identical_outputs"] = all(torch.equal(r, v["draw_commands_buffer"]) for r, v in zip(lref, late))
out["late_pass_bit_reset_ms"] = timeit(reset_bits)  # included in both of the above
eng.status()
print(json.dumps(out, indent=1))
