"""Config 4 (10 000 point lights, 240 x 135 x 32 clusters, 1080p depth): how many sphere-box tests the light assignment
EXECUTES, per level of its three-level cull, against the 2 x active x lights the reference's shader runs
(light_culling.comp:106-147: every light twice per active cluster) — SURVEY.md §8d prices this stage in tests per second
and per cent of the vector-FP32 peak (~20 flop per test), not in bytes.  The counts come from a -DORBIT_TRIAGE build
(bash tools/mktriage.sh: one atomic per wave); the time is the product library's, measured in the same process.
usage (GPU box): python tools/cluster_tests.py [out.json]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np, torch
import scenes as sc
from orbit_amd import layouts as L, _lib, passes
from orbit_amd.engine import Engine

FLOP_PER_TEST, PEAK_TFLOPS = 20.0, 157.3  # SURVEY.md §8d; MI355X vector FP32 (MI355X_MICROARCH.md)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
tri_path = os.path.join(ROOT, "tools", "variants", "triage.so")
if not os.path.exists(tri_path):
    sys.exit(f"{tri_path} is missing: build it here with `bash tools/mktriage.sh`")
tri = _lib.load_variant(tri_path)
tri.orbit_debug_read_cluster_tests.restype = C.c_int32
tri.orbit_debug_read_cluster_tests.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]

cam = sc.default_camera()
W, H, nl = 1920, 1080, 10_000
cx, cy, cz = 240, 135, 32
gdepth, glights = dev(sc.make_depth(9, W, H, cam)), dev(sc.make_lights(4, nl))
der = passes.cluster_settings_derive(passes.ClusterSettings(screen_resolution=(W, H), z_slice_count=cz, far_plane=200.0), cam.z_near)
zs, zb = float(der.z_scale), float(der.z_bias)  # ClusterSettings (cluster.rs:35-72), the product's host mirror
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
gimg = torch.zeros((total, 2), dtype=torch.int32, device="cuda")
outs = []
for lib in (None, tri):
    eng = Engine(0, _library=lib, max_lights=12_000)
    gl = torch.zeros(4 + 4 * lcap, dtype=torch.uint8, device="cuda")
    eng.cluster_mark(push, gdepth, gm, gb)
    eng.cluster_compact((cx, cy, cz), gm, gu, total)
    eng.cluster_assign(info, gu, gb, glights, gl, lcap, gimg)
    torch.cuda.synchronize()
    outs.append((eng, gl))
assert torch.equal(outs[0][1], outs[1][1]), "product and triage builds disagree on the light index buffer"
eng, gl = outs[0]
ts = []
for k in range(25):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); eng.cluster_assign(info, gu, gb, glights, gl, lcap, gimg); b.record(); torch.cuda.synchronize()
    if k >= 5: ts.append(a.elapsed_time(b))
assign_ms = float(np.median(ts))
teng = outs[1][0]
c = (C.c_ulonglong * 8)()
tri.orbit_debug_read_cluster_tests(teng._ctx, c)  # clear what the warm-up call counted
teng.cluster_assign(info, gu, gb, glights, outs[1][1], lcap, gimg)
assert tri.orbit_debug_read_cluster_tests(teng._ctx, c) == 0
active, indices = int(gu[12:16].view(torch.int32).item()), int(gl[:4].view(torch.int32).item())
lv = dict(coarse_vs_union_of_256_clusters=int(c[0]), count_launch_vs_union_of_8=int(c[2]), count_launch_vs_cluster=int(c[3]),
          write_launch_vs_union_of_8=int(c[4]), write_launch_vs_cluster=int(c[5]))
executed = sum(lv.values())
nominal = 2 * active * nl
out = {"scene": f"config 4: {nl} point lights, {cx}x{cy}x{cz} clusters, {active} active, {indices} light indices",
       "assign_ms": assign_ms, "sphere_box_tests_executed": lv, "executed_total": executed,
       "coarse_candidates_total": int(c[1]), "largest_candidate_list_of_a_group": int(c[7]),
       "write_launch_blocks_served_from_the_hit_cache": int(c[6]), "blocks_of_8_clusters": (active + 7) // 8,
       "reference_tests_2_x_active_x_lights": nominal, "executed_over_reference": executed / nominal,
       "executed_tests_per_s": executed / (assign_ms * 1e-3),
       "executed_tflops_at_20_flop_per_test": executed * FLOP_PER_TEST / (assign_ms * 1e-3) / 1e12,
       "fraction_of_vector_fp32_peak": executed * FLOP_PER_TEST / (assign_ms * 1e-3) / 1e12 / PEAK_TFLOPS,
       "reference_equivalent_tests_per_s": nominal / (assign_ms * 1e-3),
       "note": "the three-level cull executes a small fraction of the reference's tests, so the stage is bound by its "
               "launches and dependent loads, not by vector FP32: the roofline SURVEY §8d names does not bind it"}
# life of the count launch's blocks (wall clock, 100 MHz): when they start and end relative to the first start — from
# a build with the stamps and nothing else (bash tools/mkvariant_any.sh stamps light_cluster -DORBIT_TRIAGE_STAMPS)
st = (C.c_ulonglong * (2 * 8192))()
sp = os.path.join(ROOT, "tools", "variants", "stamps.so")
if os.path.exists(sp):
    slib = _lib.load_variant(sp)
    slib.orbit_debug_read_wave_stamps.restype = C.c_int32
    slib.orbit_debug_read_wave_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    seng = Engine(0, _library=slib, max_lights=12_000)
    for _ in range(3):
        seng.cluster_assign(info, gu, gb, glights, outs[1][1], lcap, gimg)
    assert slib.orbit_debug_read_wave_stamps(seng._ctx, st) == 0
    a = np.array(st[: 2 * 4096], dtype=np.float64).reshape(-1, 2)
    a = a[a[:, 0] > 0]
    nb = min((active + 7) // 8, 4096)  # kClustersPerBlock = 8; blocks 0..4095 are stamped
    t0 = a[:, 0].min()
    us = (a - t0) / 100.0
    w, idle = us[:nb], us[nb:]
    q = lambda x: [round(float(v), 2) for v in np.percentile(x, [0, 50, 90, 99, 100])]
    out["count_launch_block_life_us"] = {"blocks_stamped": int(len(a)), "working_blocks": int(nb),
        "working_begin_p0_50_90_99_100": q(w[:, 0]), "working_end_p0_50_90_99_100": q(w[:, 1]),
        "working_duration_p0_50_90_99_100": q(w[:, 1] - w[:, 0]),
        "slowest_working_blocks_index_duration_filtered_coarse": [[int(i), round(float(w[i, 1] - w[i, 0]), 2), int(st[2 * (4096 + i) + 1]) >> 32, int(st[2 * (4096 + i) + 1]) & 0xFFFFFFFF] for i in np.argsort(w[:, 1] - w[:, 0])[::-1][:16]],
        "blocks_over_12_us": int(((w[:, 1] - w[:, 0]) > 12.0).sum()),
        # the slowest blocks' phase stamps (us from the block's begin; stepped path: filtered / tested of step 0, then of step 1)
        "slowest_blocks_phases": [[int(i), round(float(w[i, 1] - w[i, 0]), 2)] + [((int(st[2 * (4096 + i)]) >> (16 * k)) & 0xFFFF) / 100.0 for k in range(4)] for i in np.argsort(w[:, 1] - w[:, 0])[::-1][:8]],
        "coarse_candidates_of_the_blocks_group_p10_50_90_99_100": [int(v) for v in np.percentile([int(st[2 * (4096 + i) + 1]) & 0xFFFFFFFF for i in range(nb)], [10, 50, 90, 99, 100])],
        "candidates_behind_the_blocks_filter_p10_50_90_99_100": [int(v) for v in np.percentile([int(st[2 * (4096 + i) + 1]) >> 32 for i in range(nb)], [10, 50, 90, 99, 100])],
        # thread 0's first item, us from the block's begin: setup loads consumed | boxes united | candidates filtered | tested | end
        "phases_p10_50_90": (lambda ph, dur: {name: [round(float(v), 2) for v in np.percentile(col, [10, 50, 90])] for name, col in
                             (("setup_loads_consumed", ph[:, 0]), ("boxes_united", ph[:, 1]), ("candidates_filtered", ph[:, 2]),
                              ("tested", ph[:, 3]), ("end", dur))})(
            np.array([[((int(st[2 * (4096 + i)]) >> (16 * k)) & 0xFFFF) / 100.0 for k in range(4)] for i in range(nb)
                      if (int(st[2 * (4096 + i)]) >> 48) != 0]),
            np.array([w[i, 1] - w[i, 0] for i in range(nb) if (int(st[2 * (4096 + i)]) >> 48) != 0])),
        "idle_begin_p0_50_90_99_100": q(idle[:, 0]) if len(idle) else None, "idle_end_p0_50_90_99_100": q(idle[:, 1]) if len(idle) else None}
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
