"""Secondary timings (not the bench.py headline): BASELINE configs 2-4 stand-ins on one MI355X.
HIP-event medians over 20 runs after 5 warm-ups; parity of each stage is covered by tests/."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import scenes as sc
from oracle import oracle
from orbit_amd import layouts as L
from orbit_amd.engine import Engine, depth_pyramid_desc

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()

def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))

def main():
    eng = Engine(0, max_entities=100_000, max_dispatches=1_000_000, max_draws=1_000_000, max_lights=12_000)
    out = {}
    cam = sc.default_camera()
    # config 2: Sponza-class, frustum + cone, pass 0
    s2 = sc.make_scene(2, 400, n_meshes=400, meshlets_per_mesh=(4, 26), extent=(15.0, 6.0, 9.0))
    # config 3/4: Bistro-class, two-pass HiZ at 1080p
    s3 = sc.make_scene(3, 2900, n_meshes=2900, meshlets_per_mesh=(4, 38), extent=(75.0, 20.0, 75.0))
    W, H = 1920, 1080
    for name, s in (("config2", s2), ("config3", s3)):
        g = [dev(a) for a in (s.entity_draw_buffer(), s.mesh_infos, s.entities, s.meshlets, s.materials)]
        cap_d, cap_c = s.max_dispatches() + 8, s.lod0_meshlets + 8
        disp = torch.zeros(12 + 16 * cap_d, dtype=torch.uint8, device="cuda"); draw = torch.zeros(4 + 28 * cap_c, dtype=torch.uint8, device="cuda")
        ci0 = sc.make_cull_info(cam.view, cam.planes)
        def cull0():
            eng.entity_cull(ci0, g[0], g[1], disp, g[2], s.entity_draw_count, cap_d)
            eng.meshlet_cull(ci0, disp, g[3], draw, g[2], g[4], cap_d, cap_c, material_count=len(s.materials))
        out[name] = {"entities": s.entity_draw_count, "meshlets_lod0": s.lod0_meshlets, "pass0_cull_ms": timeit(cull0)}
        if name == "config3":
            depth = sc.make_depth(9, W, H, cam); gdepth = dev(depth)
            d = depth_pyramid_desc(W, H); pyr = torch.zeros(d.total_texels, dtype=torch.float32, device="cuda")
            out[name]["depth_reduce_ms"] = timeit(lambda: eng.depth_reduce(gdepth, W, H, pyr))
            evis = torch.zeros((s.entity_draw_count + 31) // 32, dtype=torch.int32, device="cuda"); mvis = torch.zeros(s.vis_words, dtype=torch.int32, device="cuda")
            ci1 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=1)
            ci2 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
            ps = (d.width, d.height)
            def frame():
                eng.entity_cull(ci1, g[0], g[1], disp, g[2], s.entity_draw_count, cap_d, visibility_buffer=evis)
                eng.meshlet_cull(ci1, disp, g[3], draw, g[2], g[4], cap_d, cap_c, meshlet_visibility_buffer=mvis, material_count=len(s.materials))
                eng.depth_reduce(gdepth, W, H, pyr)
                eng.entity_cull(ci2, g[0], g[1], disp, g[2], s.entity_draw_count, cap_d, visibility_buffer=evis, depth_pyramid=pyr, depth_pyramid_size=ps)
                eng.meshlet_cull(ci2, disp, g[3], draw, g[2], g[4], cap_d, cap_c, meshlet_visibility_buffer=mvis, depth_pyramid=pyr, depth_pyramid_size=ps, material_count=len(s.materials))
            out[name]["two_pass_frame_ms"] = timeit(frame)
            # the same frame captured once into a HIP graph and replayed (12 launches, no per-call host work)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                frame()
            out[name]["two_pass_frame_graph_ms"] = timeit(graph.replay)
            # config 4: + 10k point lights, 240x135x32 clusters
            nl = 10_000
            lights = sc.make_lights(4, nl); glights = dev(lights)
            cx, cy, cz = 240, 135, 32
            zs, zb = oracle.cluster_grid_info(cam.z_near, 200.0, cz)
            push = np.zeros((), dtype=L.MARK_ACTIVE_PUSH)
            push["cluster_count"], push["tile_size_px"], push["screen_size"] = (cx, cy, cz), 8, (W, H)
            push["z_near"], push["z_far"], push["z_scale"], push["z_bias"], push["depth_buffer_sample_count"] = cam.z_near, 200.0, zs, zb, 1
            info = np.zeros((), dtype=L.CLUSTER_CULL_INFO)
            info["world_to_view_matrix"] = sc.mat4_cols(cam.view)
            info["screen_to_view_matrix"] = sc.mat4_cols(np.linalg.inv(cam.proj.astype(np.float64)).astype(np.float32))
            info["cluster_count"], info["tile_size_px"], info["screen_size"] = (cx, cy, cz), 8, (W, H)
            info["z_near"], info["z_far"], info["global_light_count"] = cam.z_near, 200.0, nl
            total = cx * cy * cz
            gm = torch.zeros(cx * cy, dtype=torch.int32, device="cuda"); gb = torch.zeros((total, 2), dtype=torch.int32, device="cuda")
            gu = torch.zeros(16 + 4 * total, dtype=torch.uint8, device="cuda")
            lcap = total * 32
            gl = torch.zeros(4 + 4 * lcap, dtype=torch.uint8, device="cuda"); gimg = torch.zeros((total, 2), dtype=torch.int32, device="cuda")
            c4 = {"lights": nl, "clusters": total}
            c4["mark_ms"] = timeit(lambda: eng.cluster_mark(push, gdepth, gm, gb))
            c4["compact_ms"] = timeit(lambda: eng.cluster_compact((cx, cy, cz), gm, gu, total))
            c4["assign_ms"] = timeit(lambda: eng.cluster_assign(info, gu, gb, glights, gl, lcap, gimg), n=10, warm=2)
            torch.cuda.synchronize()
            c4["active_clusters"] = int(gu[12:16].view(torch.int32).item()); c4["light_indices"] = int(gl[:4].view(torch.int32).item())
            # what the reference's shader would have run in that time (2 x active x lights); the tests the three-level cull
            # really executes are counted by tools/cluster_tests.py (0.26 % of these at this configuration)
            c4["reference_equivalent_sphere_tests_per_s"] = 2.0 * c4["active_clusters"] * nl / (c4["assign_ms"] * 1e-3)
            # distribution of lights per active cluster (compacted order) and per block of 8: what the hit cache sees
            na = c4["active_clusters"]
            uidx = gu[16:16 + 4 * na].view(torch.int32).long()
            cnt = gimg.view(torch.int32).view(-1, 2)[uidx, 1].cpu().numpy()
            pad = np.zeros((-len(cnt)) % 8, dtype=cnt.dtype)
            bmax = np.concatenate([cnt, pad]).reshape(-1, 8).max(axis=1)
            c4["lights_per_cluster"] = {"mean": float(cnt.mean()), "p99": int(np.percentile(cnt, 99)), "max": int(cnt.max()),
                                        "blocks_over_16": int((bmax > 16).sum()), "blocks": int(len(bmax))}
            out["config4"] = c4
    try: eng.status()
    except Exception as e: out["status"] = str(e)
    print(json.dumps(out, indent=1))

main()
