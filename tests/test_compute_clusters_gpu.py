"""orbit_compute_clusters as ONE call (src/passes/cluster.rs:368-591) beyond what tests/test_gpu_parity.py's _light_clusters
checks per scene: its mark launch leaves the compaction's chunk counts behind and the assignment clears them, its count
launch takes the groups in the order the PREVIOUS call left — state that lives in the context between calls.  So: calls
of different grids, light sets and active-cluster counts one after the other on one context, in both directions; a
refused call in between (nothing may be left behind by it); the separate entry points interleaved with the one call;
the call replayed as a graph.  Always against the oracle's five buffers."""
import numpy as np
import pytest

import scenes as sc
from orbit_amd import layouts as L
from test_gpu_parity import cluster_inputs, dev, host, torch_mod  # noqa: F401

pytestmark = pytest.mark.gpu

# (W, H, lights, tile, z slices, radius scale): small grid / few lights, the 1080p grid with a hot spot, a grid whose
# clusters saturate at 256 lights (heavy blocks: the write launch tests them again), no lights at all
SCENES = [(320, 180, 200, 8, 32, None), (1920, 1080, 3000, 8, 32, 4.0), (128, 72, 5000, 16, 16, 12.0), (200, 120, 0, 8, 8, None),
          (640, 360, 1500, 8, 32, 2.0)]


def _scene(oracle, i):
    W, H, n_lights, tile, zs, rs = SCENES[i]
    push, depth, info, lights = cluster_inputs(oracle, 11 + i, W, H, n_lights, tile, zs)
    if rs is not None:
        lights["outer_radius"] *= np.float32(rs)
    cc = [int(v) for v in push["cluster_count"]]
    total = cc[0] * cc[1] * cc[2]
    cap = cc[0] * cc[1] * max(4, cc[2])
    om, ob = oracle.cluster_mark(push, depth)
    ou, _ = oracle.cluster_compact(cc, om, cap)
    n_active = int(ou[12:16].view(np.uint32)[0])
    lcap = n_active * 256 + 16
    ol, oimg, dropped = oracle.cluster_assign(info, ou, ob, lights, lcap, total)
    assert dropped == 0
    return dict(push=push, depth=depth, info=info, lights=lights, cc=cc, total=total, cap=cap, lcap=lcap, n_active=n_active,
                n_lights=n_lights, ref=(om, ob, ou, ol, oimg))


def _buffers(torch, s):
    return dict(gm=torch.full((s["cc"][0] * s["cc"][1],), 0x5A5A5A5A, dtype=torch.int32, device="cuda"),
                gb=torch.full((s["total"], 2), 0x5A5A5A5A, dtype=torch.int32, device="cuda"),
                gu=torch.full((L.COMPACT_HEADER + 4 * s["cap"],), 0xEE, dtype=torch.uint8, device="cuda"),
                gl=torch.full((L.LIGHT_INDEX_HEADER + 4 * s["lcap"],), 0xEE, dtype=torch.uint8, device="cuda"),
                gimg=torch.zeros((s["total"], 2), dtype=torch.int32, device="cuda"),
                depth=dev(torch, s["depth"]), lights=dev(torch, s["lights"]) if s["n_lights"] else None)


def _poison(b):
    """What a call must overwrite (the image is written for active clusters only: zero, as the renderer clears it)."""
    b["gm"].fill_(0x5A5A5A5A)
    b["gb"].fill_(0x5A5A5A5A)
    b["gu"].fill_(0xEE)
    b["gl"].fill_(0xEE)
    b["gimg"].zero_()


def _call(engine, s, b):
    engine.compute_clusters(s["push"], s["info"], b["depth"], b["lights"], b["gm"], b["gb"], b["gu"], s["cap"], b["gl"], s["lcap"],
                            b["gimg"])


def _check(torch, engine, s, b, what):
    torch.cuda.synchronize()
    engine.status()
    om, ob, ou, ol, oimg = s["ref"]
    n, n_idx = s["n_active"], int(ol[:4].view(np.uint32)[0])
    assert np.array_equal(host(b["gm"], np.uint32), om), f"{what}: tile masks"
    assert np.array_equal(host(b["gb"], np.uint32).reshape(-1, 2), ob), f"{what}: depth bounds"
    assert np.array_equal(host(b["gu"])[:16 + 4 * n], ou[:16 + 4 * n]), f"{what}: compacted list"
    assert np.array_equal(host(b["gl"])[:4 + 4 * n_idx], ol[:4 + 4 * n_idx]), f"{what}: light index lists"
    assert np.array_equal(host(b["gimg"], np.uint32).reshape(-1, 2), oimg), f"{what}: (offset, count) image"


@pytest.fixture()
def engine(torch_mod):
    from orbit_amd.engine import Engine

    e = Engine(0, max_lights=6000, max_clusters=240 * 135 * 32)
    yield e
    e.close()


def test_calls_of_different_scenes_on_one_context(torch_mod, engine, oracle):
    """Every scene after every other one: more groups than the call before, fewer, none (no lights: every cluster gets
    (offset, 0)), heavy blocks after none and none after some — the group order and the heavy list of a call are what
    the one before left."""
    torch = torch_mod
    scenes = [_scene(oracle, i) for i in range(len(SCENES))]
    bufs = [_buffers(torch, s) for s in scenes]
    assert max(int(s["ref"][4][:, 1].max()) for s in scenes) == 256 and scenes[3]["n_lights"] == 0
    order = [0, 1, 2, 3, 4, 3, 2, 1, 0, 2, 0, 4, 1, 1]
    for k, i in enumerate(order):
        _poison(bufs[i])
        _call(engine, scenes[i], bufs[i])
        _check(torch, engine, scenes[i], bufs[i], f"call {k} (scene {i})")


def test_a_refused_call_leaves_nothing_behind(torch_mod, engine, oracle):
    """Arguments the later stages would refuse are refused before the mark launch (whose chunk counts only the
    assignment's first launch clears): a good call, refused ones of every kind, the good call again."""
    from orbit_amd._lib import OrbitError

    torch = torch_mod
    s = _scene(oracle, 0)
    b = _buffers(torch, s)
    _call(engine, s, b)
    _check(torch, engine, s, b, "first")
    bad_info = s["info"].copy()
    bad_info["global_light_count"] = 6001  # > caps.max_lights
    bad_grid = s["info"].copy()
    bad_grid["cluster_count"] = (s["cc"][0] + 1, s["cc"][1], s["cc"][2])  # differs from the push block's
    for info, lights, gu in ((bad_info, b["lights"], b["gu"]), (bad_grid, b["lights"], b["gu"]), (s["info"], None, b["gu"]),
                             (s["info"], b["lights"], None)):
        with pytest.raises(OrbitError):
            engine.compute_clusters(s["push"], info, b["depth"], lights, b["gm"], b["gb"], gu, s["cap"], b["gl"], s["lcap"], b["gimg"])
    _poison(b)
    _call(engine, s, b)
    _check(torch, engine, s, b, "after the refused calls")


def test_the_three_entry_points_between_calls(torch_mod, engine, oracle):
    """orbit_cluster_mark / _compact / _assign on their own (a mark launch that counts nothing, a compaction that counts
    for itself, an assignment that transforms its lights itself) between orbit_compute_clusters calls of another scene."""
    torch = torch_mod
    s0, s1 = _scene(oracle, 4), _scene(oracle, 0)
    b0, b1 = _buffers(torch, s0), _buffers(torch, s1)
    _call(engine, s0, b0)
    _check(torch, engine, s0, b0, "one call")
    engine.cluster_mark(s1["push"], b1["depth"], b1["gm"], b1["gb"])
    engine.cluster_compact(s1["cc"], b1["gm"], b1["gu"], s1["cap"])
    _call(engine, s0, b0)  # in between: the one call of the other scene
    engine.cluster_assign(s1["info"], b1["gu"], b1["gb"], b1["lights"], b1["gl"], s1["lcap"], b1["gimg"])
    _check(torch, engine, s1, b1, "three calls")
    _check(torch, engine, s0, b0, "one call again")


def test_compute_clusters_replayed_as_a_graph(torch_mod, engine, oracle):
    """Captured once, replayed: every counter the launches leave behind is back where the next replay expects it."""
    torch = torch_mod
    s = _scene(oracle, 1)
    b = _buffers(torch, s)
    _call(engine, s, b)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _call(engine, s, b)
    for k in range(4):
        _poison(b)
        g.replay()
        _check(torch, engine, s, b, f"replay {k}")


@pytest.mark.parametrize("radius_scale", [0.25, 0.5, 0.9, 2.5])
def test_one_wave_and_four_wave_items_alternate_in_one_workgroup(torch_mod, engine, oracle, radius_scale):
    """ADVICE r05 (light_cluster.hip, count launch): a group whose coarse segments all hold <= 8 candidates is ONE wave's
    item — waves 1-3 leave it without a barrier — and the next item of the same workgroup may be a four-wave one whose
    waves 1-3 write their union boxes at once.  The one-wave item's eight boxes therefore live in LDS of their own.
    1080p grid (> 65 536 active clusters: the count launch's workgroups take several items), lights in clumps so that
    neighbouring groups fall on both sides of the 8-per-segment line, many repetitions (a race is a matter of timing),
    every repetition against the oracle's five buffers."""
    torch = torch_mod
    W, H, n_lights = 1920, 1080, 2400
    push, depth, info, lights = cluster_inputs(oracle, 22, W, H, n_lights, 8, 32)
    # a tile's 64 samples spread over several z slices: > 100 000 active clusters
    depth = (depth * sc.rnd_range(22, 91, np.arange(W * H), 0.6, 1.6).reshape(H, W)).astype(np.float32)
    # clumps: every light is pulled towards one of 40 centres, half of the clumps tight, half loose
    k = np.arange(n_lights)
    centre = (k * 7) % 40
    cpos = np.stack([np.cos(centre * 0.9) * (10.0 + centre), np.full(n_lights, 2.0), -8.0 - 1.7 * centre], axis=1).astype(np.float32)
    pull = np.where(centre % 2 == 0, np.float32(0.96), np.float32(0.7))[:, None]
    lights["position"][:, :3] = (lights["position"][:, :3] * (np.float32(1.0) - pull) + cpos * pull).astype(np.float32)
    lights["outer_radius"] *= np.float32(radius_scale)
    cc = [int(v) for v in push["cluster_count"]]
    total, cap = cc[0] * cc[1] * cc[2], cc[0] * cc[1] * cc[2]
    om, ob = oracle.cluster_mark(push, depth)
    ou, _ = oracle.cluster_compact(cc, om, cap)
    n_active = int(ou[12:16].view(np.uint32)[0])
    assert n_active > 65536 + 8192, n_active  # several items per workgroup of the count launch
    lcap = n_active * 256 + 16
    ol, oimg, dropped = oracle.cluster_assign(info, ou, ob, lights, lcap, total)
    assert dropped == 0
    s = dict(push=push, depth=depth, info=info, lights=lights, cc=cc, total=total, cap=cap, lcap=lcap, n_active=n_active,
             n_lights=n_lights, ref=(om, ob, ou, ol, oimg))
    b = _buffers(torch, s)
    for rep in range(24):
        _poison(b)
        _call(engine, s, b)
        _check(torch, engine, s, b, f"repetition {rep}")


def test_hardware_log2_slice_equals_the_canonical_slice_for_every_float(torch_mod, engine, oracle):
    """cluster_mark takes a sample's z slice from v_log_f32 wherever that cannot differ from the canonical software log2
    (orbit_amd/csrc/orbit_device.h depth_slice: outside a guard band around the integers of the fused result, normal finite
    positive inputs only) and from the canonical form otherwise.  "Cannot differ" is checked, not assumed: every float bit
    pattern — all 2^32, sign, denormals, infinities and NaNs included — through both forms on the device, for the grids'
    (z_scale, z_bias) of the tests and a few hostile ones: zero mismatches, and the hardware form really is the one used
    (the canonical path decides < 1 % of the finite positive inputs)."""
    import ctypes as C

    lib = engine._lib
    lib.orbit_debug_log2_guard.restype = C.c_int32
    lib.orbit_debug_log2_guard.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_void_p]
    grids = [oracle.cluster_grid_info(0.01, 200.0, 32), oracle.cluster_grid_info(0.01, 200.0, 16),
             oracle.cluster_grid_info(0.1, 1000.0, 24), oracle.cluster_grid_info(0.01, 200.0, 8),
             (31.75, -3.0), (-2.5, 40.0), (1e-3, 0.5), (0.0, 7.0)]
    for zs, zb in grids:
        out = (C.c_uint64 * 3)()
        assert lib.orbit_debug_log2_guard(engine._ctx, 0, 0xFFFFFFFF, float(zs), float(zb), out) == 0
        assert out[0] == 0, f"z_scale {zs}, z_bias {zb}: {out[0]} floats get another slice from the hardware log2"
        worst = np.array([out[2] & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0]
        assert worst < 2.0 ** -22, f"|v_log_f32 - log2c| / (1 + |log2c|) reaches {worst}: the guard band's premise"
    # positive normal inputs only: how often the canonical path has to decide (the band + |t| >= 2^20)
    zs, zb = grids[0]
    out = (C.c_uint64 * 3)()
    assert lib.orbit_debug_log2_guard(engine._ctx, 0x00800000, 0x7F7FFFFF, float(zs), float(zb), out) == 0
    assert out[0] == 0 and out[1] < 0.01 * (0x7F7FFFFF - 0x00800000), out[1]
