"""BASELINE.json configs 2-4 on one MI355X (secondary to bench.py's config-5 headline; GPU box): per config the GPU time
of every path the library has for it, the CPU oracle's time on the same inputs with its thread count (BASELINE.md §3:
the CPU baseline is reported next to every GPU figure), whether the GPU output equals the oracle's bit for bit, and
algorithmic bytes / time (SURVEY.md §8d's formulas).  Workloads: tests/config_scenes.py — the very scenes
tests/test_configs_gpu.py checks.  Protocol: 5 warm-ups, then >= 20 runs each inside its own HIP event
pair, the median.  Usage: python tools/bench_configs.py > profiles/r04_configs.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import config_scenes as cs
from oracle import oracle
from orbit_amd import layouts as L
from orbit_amd.engine import Engine, depth_pyramid_desc


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()


def host(t, dtype=np.uint8):
    return t.cpu().numpy().view(dtype)


def timeit(fn, n=20, warm=5, lead=0):
    """lead = calls enqueued right in front of the timed one (0: the timed call starts on an idle queue and its launch
    latency is inside the event pair — the floor model's launch_us; 2: the call as one of a flow of frames)"""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(lead):
            fn()
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def graph_of(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def cpu_time(fn, min_s=1.0, min_reps=3, max_reps=200):
    t, reps, out = 0.0, 0, None
    while (t < min_s or reps < min_reps) and reps < max_reps:
        t0 = time.perf_counter()
        out = fn()
        t += time.perf_counter() - t0
        reps += 1
    return t / reps, reps, out


class GScene:
    def __init__(self, s):
        self.s = s
        self.draws, self.mesh, self.ent, self.mlt, self.mat = (dev(a) for a in (
            s.entity_draw_buffer(), s.mesh_infos, s.entities, s.meshlets, s.materials))
        self.cap_d, self.cap_c = s.max_dispatches() + 8, s.lod0_meshlets + 8

    def view(self, ci, evis=None, mvis=None, pyr=None, psize=(0, 0)):
        s = self.s
        disp = torch.zeros(12 + 16 * self.cap_d, dtype=torch.uint8, device="cuda")
        draw = torch.zeros(4 + 28 * self.cap_c, dtype=torch.uint8, device="cuda")
        v = dict(cull_info=ci, entity_draw_buffer=self.draws, mesh_info_buffer=self.mesh, meshlet_dispatch_buffer=disp,
                 entity_buffer=self.ent, entity_draw_count=s.entity_draw_count, dispatch_capacity=self.cap_d,
                 meshlet_buffer=self.mlt, draw_commands_buffer=draw, material_buffer=self.mat, draw_capacity=self.cap_c,
                 visibility_buffer=evis, meshlet_visibility_buffer=mvis, material_count=len(s.materials))
        if pyr is not None:
            v.update(depth_pyramid=pyr, depth_pyramid_size=psize)
        return v, disp, draw


PATHS = ("launch_chain", "launch_chain_from_streams", "one_launch")


def engine_for(path, g):
    eng = Engine(0, max_entities=8192, max_dispatches=40_000, max_draws=200_000, cull_path=2 if path == "one_launch" else 1)
    if path == "launch_chain_from_streams":
        ms = eng.meshlet_stream(g.mlt, 0, len(g.s.meshlets))
        ms.set_materials(g.mat, len(g.s.materials))
        eng.bind_meshlet_stream(ms)
    return eng


def cull_bytes(n_e, n_m, n_d, n_draw, tested=0, occlusion=False):
    """entity_cull N_e (12 + 64 + 28) + 16 N_d + 4 ceil(N_e / 32); meshlet_cull 32 N_m + 16 N_d + 64 N_e +
    28 N_draw + 4 (+ 8 N_m / 32 visibility words and 4 N_tested pyramid texels in the occlusion passes)."""
    b = n_e * 104 + 16 * n_d + 4 * ((n_e + 31) // 32) + 32 * n_m + 16 * n_d + 64 * n_e + 28 * n_draw + 4
    if occlusion:
        b += 8 * n_m // 32 + 4 * tested
    return b


def counts(disp, draw):
    n_d = int(disp[:4].view(np.uint32)[0])
    n_m = int(disp[12:12 + 16 * n_d].view(np.uint32).reshape(-1, 4)[:, 2].sum())
    return n_d, n_m, int(draw[:4].view(np.uint32)[0])


def same(gdisp, gdraw, odisp, odraw):
    n_d, n = int(odisp[:4].view(np.uint32)[0]), int(odraw[:4].view(np.uint32)[0])
    return bool(np.array_equal(gdisp[:12 + 16 * n_d], odisp[:12 + 16 * n_d]) and np.array_equal(gdraw[:4 + 28 * n], odraw[:4 + 28 * n]))


def measure_hops():
    """tools/ubench/hop.hip on this box (built here when the binary did not travel): what a launch, a launch boundary, a
    dependent load, a returning atomic and an in-launch hand-over cost — the links these configs' times are made of."""
    import subprocess

    exe, src = os.path.join(ROOT, "tools", "ubench", "hop.bin"), os.path.join(ROOT, "tools", "ubench", "hop.hip")
    try:
        if not os.path.exists(exe):
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, src], check=True, timeout=600)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300, check=True)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as ex:  # noqa: BLE001
        return {"error": str(ex)}


# Dependent links of each path, counted from the code: L launches, a atomics (tickets), l loads that miss,
# h hand-overs inside a launch (write-through store -> polling load).  floor = launch + (L - 1) boundaries + the links.
CHAINS = {
    # one launch: entity ticket | draws -> mesh info + entity | count out -> look-back | records stored -> "all chunks in"
    # | records -> rows + matrices | tile count out -> look-back | commands stored (drain)
    "one_launch": dict(L=1, a=2, l=5, h=3),
    # entity (one launch below 128 chunks: ticket, 2 loads, count hand-over, records) | eval (count, records, matrices,
    # rows) | scan (counts, sums) | emit (sums, records + payload, commands)
    "launch_chain": dict(L=4, a=1, l=11, h=1),
    "launch_chain_from_streams": dict(L=3, a=1, l=11, h=2),  # the chain emit scans inside its launch
    # depth_reduce: depth -> tile levels (LDS) -> level 5 written through -> last tile workgroup (2 arrival atomics) -> tail
    "depth_reduce": dict(L=1, a=2, l=2, h=1),
    # mark + the compaction's counts (depth -> masks / bounds) | compaction emit (own count -> masks + preceding counts)
    # | AABBs + lights + group order (header -> list -> bounds) | coarse filter (union box -> lights) | count (header,
    # order, segment counts -> the item's boxes and candidates) | write (header -> counts, kept hits, sums)
    "compute_clusters": dict(L=6, a=0, l=12, h=0),
}


def floor_us(hops, name, times=1):
    if "error" in hops:
        return None
    c = CHAINS[name]
    one = hops["launch_us"] + (c["L"] - 1) * hops["boundary_us"] + c["a"] * hops["atomic_hop_us"] + c["l"] * hops["load_hop_us"] \
        + c["h"] * hops["handoff_hop_us"]
    return times * one


def main():
    threads = oracle.max_threads()
    cam = cs.camera()
    hops = measure_hops()
    out = {"protocol": "HIP-event median of 20 after 5 warm-ups (SURVEY.md §8d), each timed call on an idle queue (*_behind_two_others_ms: two further calls enqueued in front of it — a call in a flow of frames, its launch latency hidden); CPU: oracle/orbit_oracle.c through "
                       "oracle/orbit_oracle_mt.c (native build) on all host threads, mean of >= 3 runs over >= 1 s",
           "cpu_threads": threads, "host_cores": os.cpu_count(), "cpu_flags": oracle.NATIVE_CFLAGS,
           "links": hops,
           "floor": "these workloads are chains of dependent memory round trips, not streams: floor_us = launch_us + "
                    "(launches - 1) x boundary_us + atomics x atomic_hop_us + loads x load_hop_us + hand-overs x "
                    "handoff_hop_us, with the links measured on this box (tools/ubench/hop.hip) and counted per path "
                    "from the code (tools/bench_configs.py CHAINS); frac_of_floor = floor_us / measured: 1 = nothing but "
                    "the chain is left.  achieved_GBps is kept for the record and means nothing here (8 000 GB/s peak)",
           "chains": CHAINS}
    oracle.build()

    # ---------------------------------------------------------------- config 2: frustum + cone, pass 0
    s2 = cs.config2_scene()
    g2 = GScene(s2)
    ci0 = cs.pass0_cull_info(cam)
    h2 = [a for a in (s2.entity_draw_buffer(), s2.mesh_infos, s2.entities, s2.meshlets, s2.materials)]
    cpu_s, reps, (odisp, odraw, _, _, _) = cpu_time(lambda: oracle.cull_frame_mt(
        ci0, h2[0], s2.entity_draw_count, h2[1], h2[2], h2[3], h2[4], g2.cap_d, g2.cap_c, threads, native=True))
    n_d, n_m, n_draw = counts(odisp, odraw)
    c2 = {"workload": f"config 2 (Sponza-class stand-in): {s2.entity_draw_count} entities, {s2.lod0_meshlets} meshlets LOD 0, "
                      f"entity_cull + meshlet_cull, frustum + cone, occlusion_pass 0",
          "dispatch_records": n_d, "meshlets_into_meshlet_cull": n_m, "draw_commands": n_draw,
          "algorithmic_bytes": cull_bytes(s2.entity_draw_count, n_m, n_d, n_draw), "gpu": {}}
    ok = True
    for path in PATHS:
        eng = engine_for(path, g2)
        v, disp, draw = g2.view(ci0)
        arr, keep = eng.prepare_views([v])
        ms = timeit(lambda: eng.cull_views_prepared(arr))
        gr = graph_of(lambda: eng.cull_views_prepared(arr))
        fl = floor_us(hops, path)
        c2["gpu"][path] = {"cull_ms": ms, "cull_graph_replay_ms": timeit(gr.replay), "floor_us": fl,
                           "frac_of_floor": (fl / (ms * 1e3)) if fl else None,
                           "cull_behind_two_others_ms": timeit(lambda: eng.cull_views_prepared(arr), lead=2)}
        torch.cuda.synchronize()
        eng.status()
        ok = ok and same(host(disp), host(draw), odisp, odraw)
        del gr
        eng.close()
    best = min(v["cull_ms"] for v in c2["gpu"].values())
    c2["gpu_output_bit_identical"] = ok
    c2["meshlets_per_sec"] = n_m / (best * 1e-3)
    c2["achieved_GBps"] = c2["algorithmic_bytes"] / (best * 1e-3) / 1e9
    c2["cpu_baseline"] = {"ms": cpu_s * 1e3, "value": n_m / cpu_s, "unit": "meshlets/s", "cores": threads, "kind": "port",
                          "sample": f"the whole workload, {reps} runs"}
    out["config2"] = c2

    # ---------------------------------------------------------------- config 3: the two-pass frame at 1080p
    s3 = cs.config3_scene()
    g3 = GScene(s3)
    W, H = cs.SCREEN
    depth = cs.config3_depth(cam)
    ev0, mv0 = cs.frame0_bits(oracle, s3, cam, depth)
    ci1, ci2 = cs.two_pass_cull_infos(cam)
    d = depth_pyramid_desc(W, H)
    ps = (d.width, d.height)
    h3 = [a for a in (s3.entity_draw_buffer(), s3.mesh_infos, s3.entities, s3.meshlets, s3.materials)]

    def cpu_frame():
        d1, w1, _, _, _ = oracle.cull_frame_mt(ci1, h3[0], s3.entity_draw_count, h3[1], h3[2], h3[3], h3[4], g3.cap_d,
                                               g3.cap_c, threads, visibility=ev0, meshlet_visibility=mv0, native=True)
        opyr, _ = oracle.depth_reduce(depth, W, H)
        d2, w2, e2, m2, _ = oracle.cull_frame_mt(ci2, h3[0], s3.entity_draw_count, h3[1], h3[2], h3[3], h3[4], g3.cap_d,
                                                 g3.cap_c, threads, visibility=ev0, meshlet_visibility=mv0, pyramid=opyr,
                                                 pyramid_size=ps, native=True)
        return d1, w1, d2, w2, e2, m2, opyr

    cpu_s, reps, (od1, ow1, od2, ow2, oe2, om2, opyr) = cpu_time(cpu_frame)
    nd1, nm1, ndraw1 = counts(od1, ow1)
    nd2, nm2, ndraw2 = counts(od2, ow2)
    mips = sum(d.mip_width[k] * d.mip_height[k] for k in range(d.mip_levels))
    reduce_bytes = 4 * (W * H + 2 * mips - 1)
    c3 = {"workload": f"config 3 (Bistro-class stand-in): {s3.entity_draw_count} entities, {s3.lod0_meshlets} meshlets LOD 0, "
                      f"{W}x{H}, K = 64 occluders + ground; frame = pass-1 cull (bits of the oracle's frame 0) -> "
                      f"depth_reduce -> pass-2 cull",
          "pass1": {"dispatch_records": nd1, "meshlets_into_meshlet_cull": nm1, "draw_commands": ndraw1},
          "pass2": {"dispatch_records": nd2, "meshlets_into_meshlet_cull": nm2, "draw_commands": ndraw2},
          "algorithmic_bytes": cull_bytes(s3.entity_draw_count, nm1, nd1, ndraw1, occlusion=True) + reduce_bytes
                               + cull_bytes(s3.entity_draw_count, nm2, nd2, ndraw2, tested=nm2, occlusion=True),
          "depth_reduce_algorithmic_bytes": reduce_bytes, "gpu": {}}
    ok = True
    gdepth = dev(depth).view(torch.float32)
    for path in PATHS:
        eng = engine_for(path, g3)
        evis, mvis = dev(ev0), dev(mv0)
        ev_keep, mv_keep = evis.clone(), mvis.clone()
        pyr = torch.zeros(d.total_texels, dtype=torch.float32, device="cuda")
        v1, disp1, draw1 = g3.view(ci1, evis, mvis)
        v2, disp2, draw2 = g3.view(ci2, evis, mvis, pyr, ps)
        a1, k1 = eng.prepare_views([v1])
        a2, k2 = eng.prepare_views([v2])

        def frame():
            # (the bits go back to the previous frame's before every frame: pass 2 rewrites them)
            evis.copy_(ev_keep)
            mvis.copy_(mv_keep)
            eng.cull_views_prepared(a1)
            eng.depth_reduce(gdepth, W, H, pyr)
            eng.cull_views_prepared(a2)

        def reset_only():
            evis.copy_(ev_keep)
            mvis.copy_(mv_keep)

        r = {"pass1_cull_ms": timeit(lambda: eng.cull_views_prepared(a1)),
             "depth_reduce_ms": timeit(lambda: eng.depth_reduce(gdepth, W, H, pyr))}
        reset_ms = timeit(reset_only)
        r["two_pass_frame_ms"] = timeit(frame) - reset_ms
        gr = graph_of(frame)
        r["two_pass_frame_graph_replay_ms"] = timeit(gr.replay) - reset_ms
        r["bits_reset_ms_subtracted"] = reset_ms
        fl, fd = floor_us(hops, path), floor_us(hops, "depth_reduce")
        if fl:  # pass-1 cull + depth_reduce + pass-2 cull: two culls' chains and the pyramid's, two boundaries between them
            r["pass1_cull_floor_us"] = fl
            r["pass1_cull_frac_of_floor"] = fl / (r["pass1_cull_ms"] * 1e3)
            r["depth_reduce_floor_us"] = fd
            r["depth_reduce_frac_of_floor"] = fd / (r["depth_reduce_ms"] * 1e3)
            r["two_pass_frame_floor_us"] = 2 * fl + fd - 2 * hops["launch_us"] + 2 * hops["boundary_us"]
            r["two_pass_frame_frac_of_floor"] = r["two_pass_frame_floor_us"] / (r["two_pass_frame_ms"] * 1e3)
        frame()
        torch.cuda.synchronize()
        eng.status()
        ok = ok and same(host(disp1), host(draw1), od1, ow1) and same(host(disp2), host(draw2), od2, ow2)
        ok = ok and np.array_equal(host(evis, np.uint32), oe2) and np.array_equal(host(mvis, np.uint32), om2)
        ok = ok and np.array_equal(host(pyr, np.uint32), opyr.view(np.uint32).reshape(-1))
        c3["gpu"][path] = r
        del gr
        eng.close()
    best = min(v["two_pass_frame_ms"] for v in c3["gpu"].values())
    c3["gpu_output_bit_identical"] = bool(ok)
    c3["meshlets_per_sec"] = (nm1 + nm2) / (best * 1e-3)
    c3["achieved_GBps"] = c3["algorithmic_bytes"] / (best * 1e-3) / 1e9
    c3["cpu_baseline"] = {"ms": cpu_s * 1e3, "value": (nm1 + nm2) / cpu_s, "unit": "meshlets/s", "cores": threads,
                          "kind": "port", "sample": f"the whole frame (both culls on {threads} threads, depth_reduce on one), {reps} runs"}
    out["config3"] = c3

    # ---------------------------------------------------------------- config 4: + 10 000 lights, clustered binning
    push, info, lights = cs.config4_inputs(oracle, cam)
    cx, cy, cz = cs.CLUSTERS
    total = cx * cy * cz
    cap = cx * cy * max(4, cz)
    lcap = total * 32
    t0 = time.perf_counter()
    om, ob = oracle.cluster_mark(push, depth)
    t_mark = time.perf_counter() - t0
    t0 = time.perf_counter()
    ou, _ = oracle.cluster_compact((cx, cy, cz), om, cap)
    t_compact = time.perf_counter() - t0
    t0 = time.perf_counter()
    ol, oimg, _ = oracle.cluster_assign(info, ou, ob, lights, lcap, total)
    t_assign = time.perf_counter() - t0
    n_active, n_idx = int(ou[12:16].view(np.uint32)[0]), int(ol[:4].view(np.uint32)[0])
    eng = Engine(0, max_lights=12_000, max_clusters=total)
    gm = torch.zeros(cx * cy, dtype=torch.int32, device="cuda")
    gb = torch.zeros((total, 2), dtype=torch.int32, device="cuda")
    gu = torch.zeros(16 + 4 * cap, dtype=torch.uint8, device="cuda")
    gl = torch.zeros(4 + 4 * lcap, dtype=torch.uint8, device="cuda")
    gimg = torch.zeros((total, 2), dtype=torch.int32, device="cuda")
    glights = dev(lights)
    c4 = {"workload": f"config 4: config 3's {W}x{H} depth, {cs.N_LIGHTS} point lights, {cx}x{cy}x{cz} = {total} clusters "
                      f"(8-px tiles, far {cs.CLUSTER_FAR:g}); mark_active -> compaction -> light assignment",
          "active_clusters": n_active, "light_indices": n_idx, "gpu": {}}
    c4["gpu"]["mark_ms"] = timeit(lambda: eng.cluster_mark(push, gdepth, gm, gb))
    c4["gpu"]["compact_ms"] = timeit(lambda: eng.cluster_compact((cx, cy, cz), gm, gu, cap))
    c4["gpu"]["assign_ms"] = timeit(lambda: eng.cluster_assign(info, gu, gb, glights, gl, lcap, gimg), n=20, warm=3)
    whole = lambda: eng.compute_clusters(push, info, gdepth, glights, gm, gb, gu, cap, gl, lcap, gimg)  # noqa: E731
    c4["gpu"]["compute_clusters_ms"] = timeit(whole)
    c4["gpu"]["compute_clusters_behind_two_others_ms"] = timeit(whole, lead=2)
    gr = graph_of(whole)
    c4["gpu"]["compute_clusters_graph_replay_ms"] = timeit(gr.replay)
    fl = floor_us(hops, "compute_clusters")
    if fl:
        c4["gpu"]["compute_clusters_floor_us"] = fl
        c4["gpu"]["compute_clusters_frac_of_floor"] = fl / (c4["gpu"]["compute_clusters_ms"] * 1e3)
    torch.cuda.synchronize()
    eng.status()
    act = ou[16:16 + 4 * n_active].view(np.uint32)
    c4["gpu_output_bit_identical"] = bool(
        np.array_equal(host(gm, np.uint32), om) and np.array_equal(host(gb, np.uint32).reshape(-1, 2), ob)
        and np.array_equal(host(gu)[:16 + 4 * n_active], ou[:16 + 4 * n_active])
        and np.array_equal(host(gl)[:4 + 4 * n_idx], ol[:4 + 4 * n_idx])
        and np.array_equal(host(gimg, np.uint32).reshape(-1, 2)[act], oimg[act]))
    # mark_active 4 W H samples + 4 tiles + 8 clusters touched; the assignment is priced in tests
    touched = int((ob.reshape(-1, 2) != 0).any(axis=1).sum())
    c4["mark_algorithmic_bytes"] = 4 * W * H + 4 * cx * cy + 8 * touched
    c4["mark_achieved_GBps"] = c4["mark_algorithmic_bytes"] / (c4["gpu"]["mark_ms"] * 1e-3) / 1e9
    ref_tests = 2.0 * n_active * cs.N_LIGHTS  # what light_culling.comp runs (count loop + write loop, :121-147)
    c4["reference_sphere_box_tests"] = ref_tests
    c4["reference_equivalent_tests_per_sec"] = ref_tests / (c4["gpu"]["assign_ms"] * 1e-3)
    # SURVEY.md §8d prices the assignment against the vector-FP32 peak (~20 flop per sphere-box test; 157.3 TFLOP/s: 256 CUs
    # x 128 lanes x 2 flop x 2.4 GHz): what the reference's test count would need at this speed — the library executes
    # 0.26 % of those tests (three-level cull, profiles/r03_cluster_tests.json), so the stage is launches, not arithmetic
    c4["reference_equivalent_pct_of_fp32_peak"] = 100.0 * c4["reference_equivalent_tests_per_sec"] * 20.0 / 157.3e12
    c4["reference_equivalent_pct_of_fp32_peak_what"] = (
        "the reference's 2 x active clusters x lights sphere-box tests at ~20 flop each, per second of the assignment, "
        "against 157.3 TFLOP/s vector FP32 (SURVEY.md §8d's yardstick).  Above 100 % because the library EXECUTES 0.26 % of "
        "those tests (three-level cull; profiles/r03_cluster_tests.json: 0.3 % of the peak in executed tests): the stage is "
        "its launches and their dependent loads — see compute_clusters_floor_us — not arithmetic")
    c4["cpu_baseline"] = {"ms": (t_mark + t_compact + t_assign) * 1e3, "mark_ms": t_mark * 1e3, "compact_ms": t_compact * 1e3,
                          "assign_ms": t_assign * 1e3, "value": ref_tests / t_assign, "unit": "sphere-box tests/s (assign)",
                          "cores": 1, "kind": "port", "sample": "the whole workload, one run of oracle/orbit_oracle.c (scalar, one thread)"}
    out["config4"] = c4
    eng.close()

    # ---------------------------------------------------------------- the frame: configs 3 + 4 together
    # What the renderer records per frame around these passes (src/app.rs:1151-1212): the early cull, then — once the depth
    # buffer exists — the pyramid + the late cull, the four cascade culls, compute_clusters.  Serially on one stream, and
    # through orbit_frame_late (the three late chains side by side on streams of the context).
    from orbit_amd import passes

    engf = Engine(0, max_entities=8192, max_dispatches=40_000, max_draws=200_000, max_lights=12_000, max_clusters=total,
                  max_views=5)
    evis, mvis = dev(ev0), dev(mv0)
    ev_keep, mv_keep = evis.clone(), mvis.clone()
    pyr = torch.zeros(d.total_texels, dtype=torch.float32, device="cuda")
    v1, disp1, draw1 = g3.view(ci1, evis, mvis)
    v2, disp2, draw2 = g3.view(ci2, evis, mvis, pyr, ps)
    casc = []
    for k in range(4):
        hci, _, _ = passes.shadow_cascade(direction=(-0.45, 0.2, 0.1, 0.86), camera_position=(0.0, 2.0, 0.0),
                                          camera_orientation=(0.0, 0.0, 0.0, 1.0), camera_fov=float(cam.fov),
                                          camera_near_clip=cam.z_near, camera_aspect_ratio=cam.aspect, cascade_index=k,
                                          max_shadow_distance=80.0)
        casc.append(g3.view(passes.cull_info_to_gpu(hci)))
    a1, k1 = engf.prepare_views([v1])
    a2, k2 = engf.prepare_views([v2])
    ac, kc = engf.prepare_views([v for v, _, _ in casc])
    pyramids = [dict(depth=gdepth, width=W, height=H, pyramid=pyr)]
    clusters = dict(push=push, info=info, depth=gdepth, lights=glights, tile_depth_slice_mask=gm, depth_bounds=gb,
                    unique_cluster_buffer=gu, index_capacity=cap, light_index_buffer=gl, light_index_capacity=lcap,
                    cluster_offset_image=gimg)
    f_late, kf = engf.prepare_frame_late(pyramids=pyramids, late_views=[v2], clusters=clusters)
    f_all, kfa = engf.prepare_frame_late(pyramids=pyramids, late_views=[v2], cascade_views=[v for v, _, _ in casc], clusters=clusters)

    def reset():
        evis.copy_(ev_keep)
        mvis.copy_(mv_keep)

    def late_serial(cascades):
        engf.depth_reduce(gdepth, W, H, pyr)
        engf.cull_views_prepared(a2)
        if cascades:
            engf.cull_views_prepared(ac)
        engf.compute_clusters(push, info, gdepth, glights, gm, gb, gu, cap, gl, lcap, gimg)

    variants = {
        "late_serial": lambda: (reset(), late_serial(False)),
        "late_overlapped": lambda: (reset(), engf.frame_late(f_late)),
        "late_with_cascades_serial": lambda: (reset(), late_serial(True)),
        "late_with_cascades_overlapped": lambda: (reset(), engf.frame_late(f_all)),
        "frame_serial": lambda: (reset(), engf.cull_views_prepared(a1), late_serial(False)),
        "frame_overlapped": lambda: (reset(), engf.cull_views_prepared(a1), engf.frame_late(f_late)),
        "frame_with_cascades_serial": lambda: (reset(), engf.cull_views_prepared(a1), late_serial(True)),
        "frame_with_cascades_overlapped": lambda: (reset(), engf.cull_views_prepared(a1), engf.frame_late(f_all)),
    }
    if os.environ.get("ORBIT_FRAME_ANATOMY"):  # what the overlapped late half is made of: its chains alone and in pairs
        f_c, kf1 = engf.prepare_frame_late(clusters=clusters)
        f_cp, kf2 = engf.prepare_frame_late(pyramids=pyramids, clusters=clusters)
        f_cv, kf3 = engf.prepare_frame_late(late_views=[v2], clusters=clusters)
        f_pv, kf4 = engf.prepare_frame_late(pyramids=pyramids, late_views=[v2])
        variants = {"clusters": lambda: (reset(), engf.frame_late(f_c)),
                    "clusters+pyramid": lambda: (reset(), engf.frame_late(f_cp)),
                    "clusters+late_cull": lambda: (reset(), engf.frame_late(f_cv)),
                    "pyramid+late_cull": lambda: (reset(), engf.frame_late(f_pv)),
                    "all three": lambda: (reset(), engf.frame_late(f_late))}
    reset_ms = timeit(reset)
    fr = {"what": "configs 3 + 4 as one frame (src/app.rs:1151-1212): early cull, then the late half — pyramid + late cull, "
                  "[the four cascade culls,] compute_clusters — serially on one stream and through orbit_frame_late (the "
                  "chains side by side on streams of the context, forked and joined by events); HIP-event median of 20, "
                  "the bits' reset subtracted", "bits_reset_ms_subtracted": reset_ms, "ms": {}, "graph_replay_ms": {}}
    for name, fn in variants.items():
        fr["ms"][name] = timeit(fn) - reset_ms
        gr = graph_of(fn)
        fr["graph_replay_ms"][name] = timeit(gr.replay) - reset_ms
        del gr
    if os.environ.get("ORBIT_FRAME_ANATOMY"):
        print(json.dumps(fr, indent=1))
        return
    # the overlapped frame's outputs against the oracle (the serial ones were checked above, config by config)
    variants["frame_with_cascades_overlapped"]()
    torch.cuda.synchronize()
    engf.status()
    act = ou[16:16 + 4 * n_active].view(np.uint32)
    fr["gpu_output_bit_identical"] = bool(
        same(host(disp1), host(draw1), od1, ow1) and same(host(disp2), host(draw2), od2, ow2)
        and np.array_equal(host(evis, np.uint32), oe2) and np.array_equal(host(mvis, np.uint32), om2)
        and np.array_equal(host(pyr, np.uint32), opyr.view(np.uint32).reshape(-1))
        and np.array_equal(host(gm, np.uint32), om) and np.array_equal(host(gb, np.uint32).reshape(-1, 2), ob)
        and np.array_equal(host(gu)[:16 + 4 * n_active], ou[:16 + 4 * n_active])
        and np.array_equal(host(gl)[:4 + 4 * n_idx], ol[:4 + 4 * n_idx])
        and np.array_equal(host(gimg, np.uint32).reshape(-1, 2)[act], oimg[act]))
    out["frame_configs_3_and_4"] = fr
    engf.close()
    print(json.dumps(out, indent=1))


main()
