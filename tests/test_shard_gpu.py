"""orbit_cull_shard (include/orbit_abi_ext.h): a rank's whole cull — entity range, meshlet stage, record list and
(optionally) its own 28-B commands — as ONE call and, for pass 0 and up to 65 536 entity-draws, ONE launch (the entity
stage runs inside the evaluation launch: orbit_amd/csrc/meshlet_eval.hip shard_cull_kernel).  Checked against the
oracle and against the two calls it replaces, from every meshlet source, over shards of different sizes, launch after
launch on one context (the hand-over words must be left zero), with capacities that cut the outputs, as a replayed
graph, and beside another context that keeps the device busy."""
import numpy as np
import pytest

import scenes as sc
from orbit_amd import layouts as L
from test_gpu_parity import GpuScene, _expected_visible_records, dev, host, run_oracle, torch_mod  # noqa: F401

pytestmark = pytest.mark.gpu

CAPS = dict(max_entities=300_000, max_dispatches=2_000_000, max_draws=4_000_000)


@pytest.fixture(scope="module", params=["meshlet_buffer", "meshlet_stream", "meshlet_stream_classes"])
def source(request):
    return request.param


def _engine(torch, source, gs, **kw):
    """An engine culling from `source`; the stream (and its classes) of gs.meshlets is built once and bound."""
    from orbit_amd.engine import Engine

    eng = Engine(0, **dict(CAPS, **kw))
    ms = None
    if source != "meshlet_buffer":
        ms = eng.meshlet_stream(gs.meshlets, 0, gs.meshlets.numel() // 32)
        if source == "meshlet_stream_classes":
            ms.set_materials(gs.materials, gs.materials.numel() // 80)
        eng.bind_meshlet_stream(ms)
        torch.cuda.synchronize()
    return eng, ms


def _close(eng, ms):
    eng.bind_meshlet_stream(None)
    if ms is not None:
        ms.close()
    eng.close()


def _shard(torch, eng, gs, ci, b, e, cap_d, cap_c, commands, two_calls=False, stream=None, bufs=None):
    """One shard's cull; -> (dispatch buffer, record list, command buffer or None) as device tensors with red zones."""
    if bufs is None:
        disp = torch.full((L.DISPATCH_HEADER + 16 * cap_d + 64,), 0xAB, dtype=torch.uint8, device="cuda")
        rec = torch.full((L.VISIBLE_HEADER + 12 * cap_d + 64,), 0xCD, dtype=torch.uint8, device="cuda")
        cmd = torch.full((L.DRAW_HEADER + 28 * cap_c + 64,), 0xEF, dtype=torch.uint8, device="cuda") if commands else None
    else:
        disp, rec, cmd = bufs
    if two_calls:
        eng.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, e - b, cap_d, draw_first=b, stream=stream)
        if commands:
            eng.meshlet_cull_records_and_commands(ci, disp, gs.meshlets, rec, cmd, gs.entities, gs.materials, cap_d, cap_d,
                                                  cap_c, stream=stream)
        else:
            eng.meshlet_cull_visible_records(ci, disp, gs.meshlets, rec, gs.entities, gs.materials, cap_d, cap_d,
                                             stream=stream)
    else:
        eng.cull_shard(ci, gs.draws, gs.mesh_infos, disp, gs.entities, b, e - b, cap_d, gs.meshlets, gs.materials, rec,
                       cap_d, draw_commands_buffer=cmd, draw_capacity=cap_c if commands else 0, stream=stream)
    return disp, rec, cmd


def _oracle_shard(oracle, scene, ci, b, e, cap_d, cap_c):
    disp, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), e - b, scene.mesh_infos, scene.entities, cap_d,
                                    draw_first=b)
    draw, _, _ = oracle.meshlet_cull(ci, disp, scene.meshlets, cap_c, scene.entities, scene.materials)
    return disp, draw


@pytest.mark.parametrize("n_entities,world", [(300, 1), (2600, 3), (9000, 2)])
def test_shard_cull_is_the_two_calls_and_the_oracle(torch_mod, oracle, source, n_entities, world):
    """Every rank's shard through orbit_cull_shard, with and without its own commands: the dispatch buffer, the record list
    ({records, survivors} header included) and the commands are the oracle's, and byte-identical to what
    orbit_entity_cull_range + orbit_meshlet_cull_visible_records / _records_and_commands leave; one launch each time; the
    same context culls shard after shard (the hand-over words are left zero)."""
    from orbit_amd.dist import shard_ranges

    torch = torch_mod
    scene = sc.make_scene(40 + world, n_entities, n_meshes=max(40, n_entities // 6), meshlets_per_mesh=(1, 70), lods=2)
    cam = sc.default_camera(rot=(0.2, 0.4))
    ci = sc.make_cull_info(cam.view, cam.planes, alpha_mode_flag=L.ALPHA_ALL)
    gs = GpuScene(torch, scene)
    eng, ms = _engine(torch, source, gs)
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    launches = 0
    for commands in (False, True):
        for b, e in shard_ranges(scene.entity_draw_count, world):
            disp, rec, cmd = _shard(torch, eng, gs, ci, b, e, cap_d, cap_c, commands)
            disp2, rec2, cmd2 = _shard(torch, eng, gs, ci, b, e, cap_d, cap_c, commands, two_calls=True)
            torch.cuda.synchronize()
            eng.status()
            launches += 1
            assert eng.shard_culls() == launches, "orbit_cull_shard did not take the one launch"
            odisp, odraw = _oracle_shard(oracle, scene, ci, b, e, cap_d, cap_c)
            n_rec, orecs = L.dispatch_buffer_records(odisp)
            on, ocmds = L.draw_buffer_commands(odraw)
            n_rec = int(n_rec[0])
            assert n_rec > 0 and on > 0
            hd, hr = host(disp), host(rec)
            assert np.array_equal(hd[:12 + 16 * n_rec], odisp[:12 + 16 * n_rec]), "dispatch records differ from the oracle"
            assert bool((hd[12 + 16 * cap_d:] == 0xAB).all())
            want = _expected_visible_records(orecs, ocmds)
            assert list(hr[:8].view(np.uint32)) == [n_rec, on]
            assert np.array_equal(hr[8:8 + 12 * n_rec].view(np.uint32), want.view(np.uint32)), "record list differs"
            assert bool((hr[8 + 12 * n_rec:] == 0xCD).all()), "write behind the record list"
            assert torch.equal(rec, rec2) and torch.equal(disp[:12 + 16 * n_rec], disp2[:12 + 16 * n_rec])
            if commands:
                hc = host(cmd)
                assert np.array_equal(hc[:4 + 28 * on], odraw[:4 + 28 * on]), "commands differ from the oracle"
                assert bool((hc[4 + 28 * cap_c:] == 0xEF).all()) and torch.equal(cmd[:4 + 28 * on], cmd2[:4 + 28 * on])
    _close(eng, ms)


def test_shard_cull_many_chunks_and_the_chain_beyond(torch_mod, oracle, source):
    """235 chunks of 256 entity-draws in the one launch (every chunk adds up the counts of the chunks before it; the first
    workgroups take the chunks, the others wait), and a range beyond 256 chunks through the launch chain of the same
    call: both equal the two calls, the large one also the oracle; cull_path = 1 forces the chain."""
    torch = torch_mod
    scene = sc.make_scene(77, 70_000, n_meshes=3000, meshlets_per_mesh=(1, 40), lods=1)
    cam = sc.default_camera(rot=(0.1, 0.3))
    ci = sc.make_cull_info(cam.view, cam.planes, alpha_mode_flag=L.ALPHA_ALL)
    gs = GpuScene(torch, scene)
    eng, ms = _engine(torch, source, gs)
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    for (b, e), one_launch in (((0, 60_000), True), ((32, 70_000), False), ((69_984, 70_000), True)):
        before = eng.shard_culls()
        disp, rec, cmd = _shard(torch, eng, gs, ci, b, e, cap_d, cap_c, True)
        disp2, rec2, cmd2 = _shard(torch, eng, gs, ci, b, e, cap_d, cap_c, True, two_calls=True)
        torch.cuda.synchronize()
        eng.status()
        assert (eng.shard_culls() - before == 1) == one_launch
        n_rec = int(rec[:4].view(torch.int32).item())
        on = int(cmd[:4].view(torch.int32).item())
        assert n_rec > 0 and on > 0 and int(rec[4:8].view(torch.int32).item()) == on
        assert torch.equal(rec, rec2) and torch.equal(disp[:12 + 16 * n_rec], disp2[:12 + 16 * n_rec])
        assert torch.equal(cmd[:4 + 28 * on], cmd2[:4 + 28 * on])
        if b == 0:
            odisp, odraw = _oracle_shard(oracle, scene, ci, b, e, cap_d, cap_c)
            assert np.array_equal(host(disp)[:12 + 16 * n_rec], odisp[:12 + 16 * n_rec])
            assert np.array_equal(host(cmd)[:4 + 28 * on], odraw[:4 + 28 * on])
    _close(eng, ms)
    chain, ms2 = _engine(torch, source, gs, cull_path=1)
    disp3, rec3, cmd3 = _shard(torch, chain, gs, ci, 0, 60_000, cap_d, cap_c, True)
    torch.cuda.synchronize()
    chain.status()
    assert chain.shard_culls() == 0
    disp, rec, cmd = _shard(torch, chain, gs, ci, 0, 60_000, cap_d, cap_c, True, two_calls=True)
    torch.cuda.synchronize()
    assert torch.equal(rec3, rec)
    _close(chain, ms2)


def test_shard_cull_capacities_empty_views_and_other_passes(torch_mod, oracle, source):
    """Capacities that cut the dispatch records and the record list (latched, nothing written behind them), a view that
    sees nothing (empty list, header {0, 0}), and an occlusion pass other than 0 (the chain behind the same call)."""
    from orbit_amd._lib import E_CAPACITY, OrbitError

    torch = torch_mod
    scene = sc.make_scene(51, 1500, n_meshes=200, meshlets_per_mesh=(1, 70), lods=2)
    cam = sc.default_camera(rot=(0.2, 0.4))
    ci = sc.make_cull_info(cam.view, cam.planes, alpha_mode_flag=L.ALPHA_ALL)
    gs = GpuScene(torch, scene)
    eng, ms = _engine(torch, source, gs)
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    n = scene.entity_draw_count
    disp, rec, cmd = _shard(torch, eng, gs, ci, 0, n, cap_d, cap_c, True)
    torch.cuda.synchronize()
    eng.status()
    n_rec, on = int(rec[:4].view(torch.int32).item()), int(cmd[:4].view(torch.int32).item())
    # a dispatch buffer (and record list) 9 records short
    cut = n_rec - 9
    dispc = torch.full((L.DISPATCH_HEADER + 16 * cut + 64,), 0xAB, dtype=torch.uint8, device="cuda")
    recc = torch.full((L.VISIBLE_HEADER + 12 * cut + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    cmdc = torch.full((L.DRAW_HEADER + 28 * cap_c + 64,), 0xEF, dtype=torch.uint8, device="cuda")
    _shard(torch, eng, gs, ci, 0, n, cut, cap_c, True, bufs=(dispc, recc, cmdc))
    torch.cuda.synchronize()
    with pytest.raises(OrbitError) as ei:
        eng.status()
    assert ei.value.code == E_CAPACITY
    assert int(dispc[:4].view(torch.int32).item()) == cut and int(recc[:4].view(torch.int32).item()) == cut
    assert torch.equal(dispc[12:12 + 16 * cut], disp[12:12 + 16 * cut]) and bool((dispc[12 + 16 * cut:] == 0xAB).all())
    assert torch.equal(recc[8:8 + 12 * cut], rec[8:8 + 12 * cut]) and bool((recc[8 + 12 * cut:] == 0xCD).all())
    # the context is as good as new afterwards
    disp2, rec2, cmd2 = _shard(torch, eng, gs, ci, 0, n, cap_d, cap_c, True)
    torch.cuda.synchronize()
    eng.status()
    assert torch.equal(rec2, rec) and torch.equal(cmd2[:4 + 28 * on], cmd[:4 + 28 * on])
    # nothing in view
    away = sc.default_camera(position=(0.0, 0.0, -5000.0))
    ci0 = sc.make_cull_info(away.view, away.planes, alpha_mode_flag=L.ALPHA_ALL)
    disp0, rec0, cmd0 = _shard(torch, eng, gs, ci0, 0, n, cap_d, cap_c, True)
    torch.cuda.synchronize()
    eng.status()
    assert list(host(rec0)[:8].view(np.uint32)) == [0, 0] and int(cmd0[:4].view(torch.int32).item()) == 0
    assert list(host(disp0)[:12].view(np.uint32)) == [0, 1, 1]
    # pass 1 goes through the chain of the same call
    evis = np.random.default_rng(5).integers(0, 2 ** 32, (n + 31) // 32, dtype=np.uint32)
    mvis = np.random.default_rng(6).integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    ci1 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=1, noskip_alphamode=L.ALPHA_MASKED)
    before = eng.shard_culls()
    d1 = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
    r1 = torch.zeros(L.VISIBLE_HEADER + 12 * cap_d, dtype=torch.uint8, device="cuda")
    eng.cull_shard(ci1, gs.draws, gs.mesh_infos, d1, gs.entities, 0, n, cap_d, gs.meshlets, gs.materials, r1, cap_d,
                   visibility_buffer=dev(torch, evis), meshlet_visibility_buffer=dev(torch, mvis))
    torch.cuda.synchronize()
    eng.status()
    assert eng.shard_culls() == before
    ref = run_oracle(oracle, scene, ci1, evis=evis, mvis=mvis)
    _, orecs = L.dispatch_buffer_records(ref[0])
    on1, ocmds = L.draw_buffer_commands(ref[1])
    want = _expected_visible_records(orecs, ocmds)
    assert list(host(r1)[:8].view(np.uint32)) == [len(want), on1]
    assert np.array_equal(host(r1)[8:8 + 12 * len(want)].view(np.uint32), want.view(np.uint32))
    _close(eng, ms)


def test_shard_cull_replays_as_a_graph_beside_a_busy_device(torch_mod, source):
    """The one launch captured into a graph and replayed (the hand-over words are zero again after every launch) while a
    second context keeps culling on another stream — workgroups of the two launches share the CUs, so what a workgroup
    waits for may be held by one that is not resident yet: tickets and the patience rule must see it through."""
    torch = torch_mod
    scene = sc.make_scene(63, 6000, n_meshes=500, meshlets_per_mesh=(1, 70), lods=1)
    cam = sc.default_camera(rot=(0.2, 0.4))
    ci = sc.make_cull_info(cam.view, cam.planes, alpha_mode_flag=L.ALPHA_ALL)
    gs = GpuScene(torch, scene)
    eng, ms = _engine(torch, source, gs)
    other, ms_o = _engine(torch, "meshlet_buffer", gs)
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    n = scene.entity_draw_count
    ref = _shard(torch, eng, gs, ci, 0, n, cap_d, cap_c, True)
    torch.cuda.synchronize()
    eng.status()
    st, st_o = torch.cuda.Stream(), torch.cuda.Stream()
    bufs = tuple(torch.zeros_like(t) for t in ref)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        _shard(torch, eng, gs, ci, 0, n, cap_d, cap_c, True, stream=st, bufs=bufs)  # warm
        st.synchronize()
        with torch.cuda.graph(g, stream=st):
            _shard(torch, eng, gs, ci, 0, n, cap_d, cap_c, True, stream=st, bufs=bufs)
    obufs = tuple(torch.zeros_like(t) for t in ref)
    for rep in range(6):
        for t in bufs:
            t.zero_()
        torch.cuda.synchronize()
        for _ in range(4):  # the other context: the same shard cull, eagerly, on its own stream
            _shard(torch, other, gs, ci, 0, n, cap_d, cap_c, True, stream=st_o, bufs=obufs)
        g.replay()
        torch.cuda.synchronize()
        eng.status()
        other.status()
        for got, want in zip(bufs, ref):
            k = int(want[:4].view(torch.int32).item())
            stride = 16 if got is bufs[0] else 12 if got is bufs[1] else 28
            hdr = 12 if got is bufs[0] else 8 if got is bufs[1] else 4
            assert torch.equal(got[:hdr + stride * k], want[:hdr + stride * k]), f"replay {rep} differs"
        assert torch.equal(obufs[1][:8 + 12 * int(ref[1][:4].view(torch.int32).item())],
                           ref[1][:8 + 12 * int(ref[1][:4].view(torch.int32).item())])
    _close(eng, ms)
    _close(other, ms_o)
