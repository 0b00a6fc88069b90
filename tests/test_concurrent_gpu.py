"""Two contexts culling at the same time on one device, each on its own stream (a renderer with two frames in flight):
frame after frame both produce the list either produces alone, and no in-launch wait runs out.  Launches that hand work
over INSIDE a launch must never depend on a workgroup that is not running; the case that showed what happens when one
does — two PROCESSES sharing the device — is tests/test_bench_gpu.py::test_bench_two_ranks_on_one_device_at_full_size
(two streams of one process did not reproduce it).  2 x 10 M meshlets: every launch takes its device-filling form."""
import pytest

import scenes as sc
from orbit_amd import layouts as L

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("source", ["stream", "buffer"])
def test_two_contexts_cull_concurrently(source):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from orbit_amd import synth
    from orbit_amd.engine import Engine

    dev = torch.device("cuda", 0)
    spec = synth.C5Spec(entities=40_000)
    E, M = spec.entities, spec.meshlets_per_entity
    draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
    meshlets = synth.gen_meshlets(spec, 0, E, dev, half)
    materials = synth.gen_materials(spec, dev)
    cam = sc.default_camera(position=(0.0, 0.0, 1300.0))
    ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=0)
    disp_cap, draw_cap = E * spec.records_per_entity + 8, E * M // 2 + 1024

    class Side:
        def __init__(self):
            self.eng = Engine(0, max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap)
            if source == "stream":
                ms = self.eng.meshlet_stream(meshlets, 0, E * M)
                ms.set_materials(materials, spec.materials)
                self.eng.bind_meshlet_stream(ms)
            self.stream = torch.cuda.Stream(device=dev)
            self.disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
            self.draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev)

        def frame(self):
            self.eng.entity_cull(ci, draws, mesh, self.disp, ent, E, disp_cap, stream=self.stream)
            self.eng.meshlet_cull(ci, self.disp, meshlets, self.draw, ent, materials, disp_cap, draw_cap,
                                  material_count=spec.materials, stream=self.stream)

    a, b = Side(), Side()
    a.frame()  # alone: the reference
    torch.cuda.synchronize()
    a.eng.status()
    want = a.draw.clone()
    n = int(want[:4].view(torch.int32).item())
    assert n > 0
    for _ in range(12):  # both at once, frame after frame
        a.draw.zero_(), b.draw.zero_()
        torch.cuda.synchronize()
        a.frame(), b.frame()
        a.frame(), b.frame()
        torch.cuda.synchronize()
        a.eng.status(), b.eng.status()  # no poll ran out
        used = L.DRAW_HEADER + 28 * n
        assert torch.equal(a.draw[:used], want[:used]) and torch.equal(b.draw[:used], want[:used])


def test_chain_emit_when_every_waiting_workgroup_scans_for_itself():
    """The chain emit's first workgroups scan the tiles' counts for the rest of the launch; a workgroup whose patience
    with a chunk's sum runs out scans the chunk itself (meshlet_emit.hip emit_scan_wait) — the path that otherwise only
    runs when those first workgroups are not being dispatched.  With the patience set to zero every workgroup takes it
    at once (ten chunks here, a thousand workgroups writing the same prefixes): the list must not change."""
    import ctypes as C

    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from orbit_amd import _lib, synth
    from orbit_amd.engine import Engine

    dev = torch.device("cuda", 0)
    spec = synth.C5Spec(entities=40_000)
    E, M = spec.entities, spec.meshlets_per_entity
    draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
    meshlets = synth.gen_meshlets(spec, 0, E, dev, half)
    materials = synth.gen_materials(spec, dev)
    cam = sc.default_camera(position=(0.0, 0.0, 1300.0))
    ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=0)
    disp_cap, draw_cap = E * spec.records_per_entity + 8, E * M // 2 + 1024
    eng = Engine(0, max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap)
    ms = eng.meshlet_stream(meshlets, 0, E * M)
    ms.set_materials(materials, spec.materials)
    eng.bind_meshlet_stream(ms)
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
    draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev)

    def frame():
        draw.zero_()
        eng.entity_cull(ci, draws, mesh, disp, ent, E, disp_cap)
        eng.meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
        torch.cuda.synchronize()
        eng.status()
        return draw.clone()

    want = frame()
    n = int(want[:4].view(torch.int32).item())
    assert n > 100_000 and int(disp[:4].view(torch.int32).item()) > 10 * 1024 * 16  # more than ten scan chunks of tiles
    lib = _lib.load()
    lib.orbit_debug_set_scan_patience.restype = C.c_int32
    lib.orbit_debug_set_scan_patience.argtypes = [C.c_void_p, C.c_uint32]
    assert lib.orbit_debug_set_scan_patience(eng._ctx, 0) == 0
    try:
        for _ in range(6):
            assert torch.equal(frame(), want)
    finally:
        assert lib.orbit_debug_set_scan_patience(eng._ctx, 256) == 0
    assert torch.equal(frame(), want)


@pytest.mark.parametrize("layout,survive", [("uploaded", 0.6), ("scattered", 0.6), ("uploaded", 0.02)])
def test_chain_emit_ticketed_groups_in_their_general_form(layout, survive):
    """The chain emit hands its groups out statically first and by ticket for the tail (a launch of four and more rounds:
    here 140 000 entities = 35.8 M meshlets = 17 500 groups over 4 096 waves), and a group the pipelined loop cannot
    serve — more than 128 survivors, or survivors behind a second chain break inside a record — is done again in the
    general form behind the loop, from the group index the wave kept for that iteration.  60 % survivors make EVERY
    group such a group (ticketed ones included), offsets that follow no rule break every chain; 2 % survivors leave the
    pipelined form on its own.  The Meshlet-buffer path (payload emit, its own scan launch) is the independent answer."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from orbit_amd import synth
    from orbit_amd.engine import Engine

    dev = torch.device("cuda", 0)
    spec = synth.C5Spec(entities=140_000)
    E, M = spec.entities, spec.meshlets_per_entity
    draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
    meshlets = synth.gen_meshlets(spec, 0, E, dev, half, survive_target=survive, layout=layout)
    materials = synth.gen_materials(spec, dev)
    cam = sc.default_camera(position=(0.0, 0.0, 1300.0))
    ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=0)
    disp_cap, draw_cap = E * spec.records_per_entity + 8, E * M + 1024
    eng = Engine(0, max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap)
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
    draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev)

    def frame():
        draw[:L.DRAW_HEADER].zero_()
        eng.entity_cull(ci, draws, mesh, disp, ent, E, disp_cap)
        eng.meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
        torch.cuda.synchronize()
        eng.status()
        n = int(draw[:4].view(torch.int32).item())
        return n, draw[:L.DRAW_HEADER + 28 * n].clone()

    n_ref, ref = frame()  # from the Meshlet buffer
    assert int(disp[:4].view(torch.int32).item()) >= 4 * 4096 * 32  # four rounds of groups and more: tickets are on
    assert abs(n_ref / (E * M) - survive) < 0.25 * survive
    ms = eng.meshlet_stream(meshlets, 0, E * M)
    ms.set_materials(materials, spec.materials)
    eng.bind_meshlet_stream(ms)
    for _ in range(3):
        n, got = frame()
        assert n == n_ref and torch.equal(got, ref)
