"""orbit_frame_late (include/orbit_abi_ext.h; src/app.rs:1151-1212): the late half of a frame — {depth pyramid -> the
VisibilityWrite cull} || {the four shadow-cascade culls} || {compute_clusters} — as ONE call on three streams forked and
joined by events.  On BASELINE configs 3 + 4 (tests/config_scenes.py: the workloads tools/bench_configs.py times) every
output must be byte for byte what the serial calls produce and what the oracle says; also replayed as a graph, through
every cull path, with chains missing, and refused as a whole when one group's arguments are bad."""
import numpy as np
import pytest

import config_scenes as cs
from orbit_amd import layouts as L
from test_configs_gpu import view_of
from test_gpu_parity import GpuScene, assert_same, dev, host, run_oracle, torch_mod  # noqa: F401

pytestmark = pytest.mark.gpu


def _engine(path, torch, gs, total):
    from orbit_amd.engine import Engine

    eng = Engine(0, max_entities=8192, max_dispatches=40_000, max_draws=200_000, max_lights=12_000, max_clusters=total,
                 max_views=5, cull_path={"one_launch": 2, "launch_chain": 1, "library_choice": 0}[path])
    return eng


class _Frame:
    """Inputs, GPU buffers and oracle results of one late half on configs 3 + 4."""

    def __init__(self, torch, oracle):
        from orbit_amd import passes
        from orbit_amd.engine import depth_pyramid_desc

        self.torch = torch
        self.scene, self.cam = cs.config3_scene(), cs.camera()
        self.W, self.H = cs.SCREEN
        self.depth = cs.config3_depth(self.cam)
        self.ev0, self.mv0 = cs.frame0_bits(oracle, self.scene, self.cam, self.depth)
        self.gs = GpuScene(torch, self.scene)
        _, self.ci2 = cs.two_pass_cull_infos(self.cam)
        d = depth_pyramid_desc(self.W, self.H)
        self.d, self.ps = d, (d.width, d.height)
        self.cascade_cis = []
        for cascade in range(4):
            hci, _, _ = passes.shadow_cascade(direction=(-0.45, 0.2, 0.1, 0.86), camera_position=(0.0, 2.0, 0.0),
                                              camera_orientation=(0.0, 0.0, 0.0, 1.0), camera_fov=float(self.cam.fov),
                                              camera_near_clip=self.cam.z_near, camera_aspect_ratio=self.cam.aspect,
                                              cascade_index=cascade, max_shadow_distance=80.0)
            self.cascade_cis.append(passes.cull_info_to_gpu(hci))
        self.push, self.info, self.lights = cs.config4_inputs(oracle, self.cam)
        cx, cy, cz = cs.CLUSTERS
        self.total, self.cap, self.lcap = cx * cy * cz, cx * cy * max(4, cz), cx * cy * cz * 32
        # the oracle's side
        self.opyr, _ = oracle.depth_reduce(self.depth, self.W, self.H)
        self.ref_late = run_oracle(oracle, self.scene, self.ci2, self.ev0, self.mv0, self.opyr, self.ps)
        self.ref_casc = [run_oracle(oracle, self.scene, ci) for ci in self.cascade_cis]
        om, ob = oracle.cluster_mark(self.push, self.depth)
        ou, _ = oracle.cluster_compact((cx, cy, cz), om, self.cap)
        ol, oimg, dropped = oracle.cluster_assign(self.info, ou, ob, self.lights, self.lcap, self.total)
        assert dropped == 0
        self.ref_clusters = (om, ob, ou, ol, oimg)
        self.gdepth = dev(torch, self.depth).view(torch.float32)
        self.glights = dev(torch, self.lights)

    def buffers(self):
        torch, cx, cy = self.torch, cs.CLUSTERS[0], cs.CLUSTERS[1]
        b = dict(evis=dev(torch, self.ev0), mvis=dev(torch, self.mv0),
                 pyr=torch.full((self.d.total_texels,), -1.0, dtype=torch.float32, device="cuda"),
                 gm=torch.full((cx * cy,), 0x5A5A5A5A, dtype=torch.int32, device="cuda"),
                 gb=torch.full((self.total, 2), 0x5A5A5A5A, dtype=torch.int32, device="cuda"),
                 gu=torch.full((L.COMPACT_HEADER + 4 * self.cap,), 0xEE, dtype=torch.uint8, device="cuda"),
                 gl=torch.full((L.LIGHT_INDEX_HEADER + 4 * self.lcap,), 0xEE, dtype=torch.uint8, device="cuda"),
                 gimg=torch.zeros((self.total, 2), dtype=torch.int32, device="cuda"))
        b["late"] = view_of(torch, self.gs, self.ci2, b["evis"], b["mvis"], b["pyr"], self.ps)
        b["casc"] = [view_of(torch, self.gs, ci) for ci in self.cascade_cis]
        b["pyramids"] = [dict(depth=self.gdepth, width=self.W, height=self.H, pyramid=b["pyr"])]
        b["clusters"] = dict(push=self.push, info=self.info, depth=self.gdepth, lights=self.glights,
                             tile_depth_slice_mask=b["gm"], depth_bounds=b["gb"], unique_cluster_buffer=b["gu"],
                             index_capacity=self.cap, light_index_buffer=b["gl"], light_index_capacity=self.lcap,
                             cluster_offset_image=b["gimg"])
        return b

    def reset(self, b):
        b["evis"].copy_(dev(self.torch, self.ev0))
        b["mvis"].copy_(dev(self.torch, self.mv0))
        b["pyr"].fill_(-1.0)
        for t, v in ((b["gm"], 0x5A5A5A5A), (b["gb"], 0x5A5A5A5A), (b["gu"], 0xEE), (b["gl"], 0xEE)):
            t.fill_(v)
        b["gimg"].zero_()
        for _, disp, draw in [b["late"]] + b["casc"]:
            disp.fill_(0xAB)
            draw.fill_(0xCD)

    def check(self, b, what, late=True, casc=True, clusters=True):
        if late:
            assert np.array_equal(host(b["pyr"], np.uint32), self.opyr.view(np.uint32).reshape(-1)), f"{what}: pyramid"
            _, disp, draw = b["late"]
            assert_same((host(disp), host(draw), host(b["evis"], np.uint32), host(b["mvis"], np.uint32)), self.ref_late)
        if casc:
            for k, ((_, disp, draw), ref) in enumerate(zip(b["casc"], self.ref_casc)):
                _, cmds = assert_same((host(disp), host(draw), None, None), ref)
        if clusters:
            om, ob, ou, ol, oimg = self.ref_clusters
            n, n_idx = int(ou[12:16].view(np.uint32)[0]), int(ol[:4].view(np.uint32)[0])
            assert np.array_equal(host(b["gm"], np.uint32), om), f"{what}: tile masks"
            assert np.array_equal(host(b["gb"], np.uint32).reshape(-1, 2), ob), f"{what}: depth bounds"
            assert np.array_equal(host(b["gu"])[:16 + 4 * n], ou[:16 + 4 * n]), f"{what}: compacted list"
            assert np.array_equal(host(b["gl"])[:4 + 4 * n_idx], ol[:4 + 4 * n_idx]), f"{what}: light index lists"
            assert np.array_equal(host(b["gimg"], np.uint32).reshape(-1, 2), oimg), f"{what}: (offset, count) image"


@pytest.fixture(scope="module")
def frame(torch_mod, oracle):
    return _Frame(torch_mod, oracle)


@pytest.mark.parametrize("path", ["library_choice", "launch_chain", "one_launch"])
def test_frame_late_equals_the_serial_calls_and_the_oracle(torch_mod, frame, path):
    torch = torch_mod
    eng = _engine(path, torch, frame.gs, frame.total)
    b = frame.buffers()
    f, keep = eng.prepare_frame_late(pyramids=b["pyramids"], late_views=[b["late"][0]],
                                     cascade_views=[v for v, _, _ in b["casc"]], clusters=b["clusters"])
    side = torch.cuda.Stream()
    for rep in range(6):  # the side streams, events and scratch sets are reused call after call; twice on another stream
        frame.reset(b)
        torch.cuda.synchronize()
        if rep >= 4:
            side.wait_stream(torch.cuda.current_stream())
            eng.frame_late(f, stream=side)
            side.synchronize()
        else:
            eng.frame_late(f)
            torch.cuda.synchronize()
        eng.status()
        frame.check(b, f"{path}, call {rep}")
    # work enqueued behind the call sees all three chains' outputs WITHOUT a host wait: copies on the caller's stream
    frame.reset(b)
    eng.frame_late(f)
    snap = [b["gl"].clone(), b["casc"][3][2].clone(), b["late"][2].clone(), b["evis"].clone()]
    torch.cuda.synchronize()
    assert torch.equal(snap[0], b["gl"]) and torch.equal(snap[1], b["casc"][3][2]) and torch.equal(snap[2], b["late"][2])
    assert torch.equal(snap[3], b["evis"])
    frame.check(b, f"{path}, joined")
    # the serial calls on the same buffers: byte for byte the same (red zones and poison included)
    got = {k: b[k].clone() for k in ("pyr", "evis", "mvis", "gm", "gb", "gu", "gl", "gimg")}
    got_views = [(disp.clone(), draw.clone()) for _, disp, draw in [b["late"]] + b["casc"]]
    frame.reset(b)
    eng.depth_reduce_multi(b["pyramids"])
    eng.cull_views([b["late"][0]])
    eng.cull_views([v for v, _, _ in b["casc"]])
    eng.compute_clusters(frame.push, frame.info, frame.gdepth, frame.glights, b["gm"], b["gb"], b["gu"], frame.cap, b["gl"],
                         frame.lcap, b["gimg"])
    torch.cuda.synchronize()
    eng.status()
    for k, t in got.items():
        assert torch.equal(t, b[k]), f"{path}: {k} differs from the serial calls"
    for (gd, gw), (_, disp, draw) in zip(got_views, [b["late"]] + b["casc"]):
        assert torch.equal(gd, disp) and torch.equal(gw, draw), f"{path}: a view differs from the serial calls"
    del keep
    eng.close()


def test_frame_late_replayed_as_a_graph(torch_mod, frame):
    """The fork and the join are events only: the call is captured (the side streams join the capture) and replayed."""
    torch = torch_mod
    eng = _engine("library_choice", torch, frame.gs, frame.total)
    b = frame.buffers()
    f, keep = eng.prepare_frame_late(pyramids=b["pyramids"], late_views=[b["late"][0]],
                                     cascade_views=[v for v, _, _ in b["casc"]], clusters=b["clusters"])
    eng.frame_late(f)  # (creates the side streams and scratch sets outside the capture)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.frame_late(f)
    for rep in range(4):
        frame.reset(b)
        g.replay()
        torch.cuda.synchronize()
        eng.status()
        frame.check(b, f"replay {rep}")
    del keep
    eng.close()


@pytest.mark.parametrize("groups", ["late_only", "cascades_only", "clusters_only", "cascades_and_clusters", "late_and_clusters"])
def test_frame_late_with_chains_missing(torch_mod, frame, groups):
    torch = torch_mod
    eng = _engine("library_choice", torch, frame.gs, frame.total)
    b = frame.buffers()
    late, casc, clus = "late" in groups, "cascades" in groups, "clusters" in groups
    f, keep = eng.prepare_frame_late(pyramids=b["pyramids"] if late else (), late_views=[b["late"][0]] if late else (),
                                     cascade_views=[v for v, _, _ in b["casc"]] if casc else (),
                                     clusters=b["clusters"] if clus else None)
    for rep in range(2):
        frame.reset(b)
        eng.frame_late(f)
        torch.cuda.synchronize()
        eng.status()
        frame.check(b, groups, late=late, casc=casc, clusters=clus)
    # what was not asked for was not touched
    if not clus:
        assert bool((b["gl"] == 0xEE).all()) and bool((b["gm"] == 0x5A5A5A5A).all())
    if not casc:
        assert all(bool((draw == 0xCD).all()) for _, _, draw in b["casc"])
    if not late:
        assert bool((b["pyr"] == -1.0).all()) and bool((b["late"][2] == 0xCD).all())
    del keep
    eng.close()


def test_frame_late_is_refused_as_a_whole(torch_mod, frame):
    """A bad argument in ANY group is refused before anything is enqueued or forked: nothing is written, and the next
    good call is right (the mark launch's chunk counts in particular were not left behind)."""
    torch = torch_mod
    from orbit_amd._lib import OrbitError

    eng = _engine("library_choice", torch, frame.gs, frame.total)
    b = frame.buffers()
    good = dict(pyramids=b["pyramids"], late_views=[b["late"][0]], cascade_views=[v for v, _, _ in b["casc"]], clusters=b["clusters"])
    f, keep = eng.prepare_frame_late(**good)
    eng.frame_late(f)
    torch.cuda.synchronize()
    bad_info = frame.info.copy()
    bad_info["global_light_count"] = 12_001  # > caps.max_lights
    bad_view = dict(b["casc"][2][0], material_buffer=None)
    bad_pyr = [dict(b["pyramids"][0], pyramid=None)]
    nine = [v for v, _, _ in b["casc"]] * 2 + [b["casc"][0][0]]
    for kw in (dict(good, clusters=dict(b["clusters"], info=bad_info)),
               dict(good, cascade_views=[b["casc"][0][0], bad_view]),
               dict(good, pyramids=bad_pyr),
               dict(good, cascade_views=nine[:8])):  # 1 + 8 views
        frame.reset(b)
        fb, kb = eng.prepare_frame_late(**kw)
        with pytest.raises(OrbitError):
            eng.frame_late(fb)
        torch.cuda.synchronize()
        assert bool((b["gl"] == 0xEE).all()) and bool((b["pyr"] == -1.0).all()) and bool((b["late"][2] == 0xCD).all())
        assert all(bool((draw == 0xCD).all()) for _, _, draw in b["casc"])
    frame.reset(b)
    eng.frame_late(f)
    torch.cuda.synchronize()
    eng.status()
    frame.check(b, "after the refused calls")
    del keep
    eng.close()
