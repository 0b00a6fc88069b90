"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle,
bit-exact on identical seeded inputs.  Run with ``-m gpu`` on an MI355X."""
import os

import numpy as np
import pytest

import scenes as sc
from orbit_amd import layouts as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


# every test that takes `engine` runs four times: from the 32-B Meshlet buffer, from the derived meshlet streams, from
# the streams with their alpha classes (no material index read; a third of the materials looked up at cull time), and
# with both cull stages as ONE launch (orbit_cull_views' one-launch path, tests/fused_engine.py)
@pytest.fixture(scope="module", params=["meshlet_buffer", "meshlet_stream", "meshlet_stream_classes", "one_launch"])
def engine(torch_mod, request):
    from orbit_amd.engine import Engine
    from stream_engine import StreamEngine

    kw = dict(max_entities=300_000, max_dispatches=2_000_000, max_draws=4_000_000, max_lights=12_000,
              max_clusters=240 * 135 * 32)
    if request.param == "one_launch":  # entity + meshlet stage as ONE launch (cull_fused.hip)
        from fused_engine import FusedEngine

        e = FusedEngine(0, **kw)
    else:
        e = Engine(0, **kw) if request.param == "meshlet_buffer" else StreamEngine(
            0, classes=request.param == "meshlet_stream_classes", **kw)
    yield e
    e.close()


def dev(torch, a):
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda()


def host(t, dtype=np.uint8):
    return t.cpu().numpy().view(dtype)


class GpuScene:
    def __init__(self, torch, scene):
        self.scene = scene
        self.draws = dev(torch, scene.entity_draw_buffer())
        self.mesh_infos = dev(torch, scene.mesh_infos)
        self.entities = dev(torch, scene.entities)
        self.meshlets = dev(torch, scene.meshlets)
        self.materials = dev(torch, scene.materials)


def run_gpu(torch, engine, gs, ci, evis=None, mvis=None, pyr=None, psize=(0, 0), disp_cap=None, draw_cap=None,
            entity_draw_count=None, material_count=0):
    s = gs.scene
    disp_cap = s.max_dispatches() + 8 if disp_cap is None else disp_cap
    n = s.entity_draw_count if entity_draw_count is None else entity_draw_count
    disp = torch.full((L.DISPATCH_HEADER + 16 * disp_cap + 256,), 0xAB, dtype=torch.uint8, device="cuda")
    evis_d = None if evis is None else dev(torch, evis)
    mvis_d = None if mvis is None else dev(torch, mvis)
    pyr_d = None if pyr is None else dev(torch, pyr)
    engine.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, n, disp_cap, visibility_buffer=evis_d,
                       depth_pyramid=pyr_d, depth_pyramid_size=psize)
    draw_cap = s.lod0_meshlets + 8 if draw_cap is None else draw_cap
    draw = torch.full((L.DRAW_HEADER + 28 * draw_cap + 256,), 0xCD, dtype=torch.uint8, device="cuda")
    engine.meshlet_cull(ci, disp, gs.meshlets, draw, gs.entities, gs.materials, disp_cap, draw_cap,
                        meshlet_visibility_buffer=mvis_d, depth_pyramid=pyr_d, depth_pyramid_size=psize,
                        material_count=material_count)  # 0: alpha modes gathered; > 0: staged in LDS by the kernels
    torch.cuda.synchronize()
    # red zones: nothing may be written past the capacities handed to the library
    assert bool((disp[L.DISPATCH_HEADER + 16 * disp_cap:] == 0xAB).all()), "write past the dispatch capacity"
    assert bool((draw[L.DRAW_HEADER + 28 * draw_cap:] == 0xCD).all()), "write past the draw capacity"
    return (host(disp), host(draw), None if evis_d is None else host(evis_d, np.uint32),
            None if mvis_d is None else host(mvis_d, np.uint32))


def run_oracle(oracle, scene, ci, evis=None, mvis=None, pyr=None, psize=(0, 0), disp_cap=None, draw_cap=None,
               entity_draw_count=None):
    disp_cap = scene.max_dispatches() + 8 if disp_cap is None else disp_cap
    n = scene.entity_draw_count if entity_draw_count is None else entity_draw_count
    disp, evis_o, d1 = oracle.entity_cull(ci, scene.entity_draw_buffer(), n, scene.mesh_infos, scene.entities,
                                          disp_cap, evis, pyr, psize)
    draw_cap = scene.lod0_meshlets + 8 if draw_cap is None else draw_cap
    draw, mvis_o, d2 = oracle.meshlet_cull(ci, disp, scene.meshlets, draw_cap, scene.entities, scene.materials, mvis,
                                           pyr, psize)
    return disp, draw, evis_o, mvis_o, d1, d2


def assert_same(gpu, ref):
    gdisp, gdraw, gevis, gmvis = gpu
    odisp, odraw, oevis, omvis = ref[:4]
    gh, grecs = L.dispatch_buffer_records(gdisp)
    oh, orecs = L.dispatch_buffer_records(odisp)
    assert list(gh) == list(oh)
    assert np.array_equal(grecs.view(np.uint32), orecs.view(np.uint32)), "dispatch records differ"
    gn, gcmds = L.draw_buffer_commands(gdraw)
    on, ocmds = L.draw_buffer_commands(odraw)
    assert gn == on
    assert np.array_equal(gcmds.view(np.uint32), ocmds.view(np.uint32)), "draw commands differ"
    if oevis is not None:
        assert np.array_equal(gevis, oevis), "entity visibility bitset differs"
    if omvis is not None:
        assert np.array_equal(gmvis, omvis), "meshlet visibility bitset differs"
    return grecs, gcmds


# ----------------------------------------------------------------------------- pass 0
@pytest.mark.parametrize("seed,n,lods", [(1, 300, 1), (2, 1000, 3), (3, 37, 1), (4, 5000, 2)])
def test_pass0(torch_mod, engine, oracle, seed, n, lods):
    scene = sc.make_scene(seed, n, n_meshes=max(1, n // 3), lods=lods)
    cam = sc.default_camera(rot=(0.6, 0.8))
    ci = sc.make_cull_info(cam.view, cam.planes, lod_base=8.0)
    gs = GpuScene(torch_mod, scene)
    recs, cmds = assert_same(run_gpu(torch_mod, engine, gs, ci), run_oracle(oracle, scene, ci))
    assert len(cmds) > 0


def test_pass0_no_planes_ragged_records(torch_mod, engine, oracle):
    scene = sc.make_scene(5, 257, meshlets_per_mesh=(1, 70))  # ragged: 1..70 meshlets -> partial records
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, np.zeros((0, 4), np.float32), alpha_mode_flag=L.ALPHA_ALL)
    gs = GpuScene(torch_mod, scene)
    recs, cmds = assert_same(run_gpu(torch_mod, engine, gs, ci), run_oracle(oracle, scene, ci))
    assert len(recs) == scene.max_dispatches()


def test_cone_bytes_and_matrix_edge_cases(torch_mod, engine, oracle):
    """Every int8 value in each cone byte (the device divides by 127 with a two-op
    fma form), plus projective (w != 1), non-finite and zero model matrices."""
    scene = sc.make_scene(11, 600, meshlets_per_mesh=(20, 64))
    rng = np.random.default_rng(11)
    m = scene.meshlets
    vals = np.arange(-128, 128, dtype=np.int8)
    for c in range(3):
        col = rng.integers(-128, 128, len(m)).astype(np.int8)
        col[:256] = rng.permutation(vals)
        m["cone_axis"][:, c] = col
    cut = rng.integers(-128, 128, len(m)).astype(np.int8)
    cut[:256] = rng.permutation(vals)
    m["cone_cutoff"] = cut
    mats = scene.entities["model_matrix"].reshape(-1, 16)
    mats[3, 3] = 0.5                      # projective rows: transform_sphere divides by w
    mats[4, 7] = 0.25
    mats[5, 15] = 2.0
    mats[6, 12] = np.inf                  # non-finite translation: NaN through the w = 0 axis term
    mats[7, 13] = np.nan
    mats[8, :] = 0.0                      # zero matrix: w = 0 -> division by zero
    mats[9, 0] = -0.0
    cam = sc.default_camera(rot=(0.3, -0.4))
    for planes in (cam.planes, np.zeros((0, 4), np.float32)):
        ci = sc.make_cull_info(cam.view, planes, alpha_mode_flag=L.ALPHA_ALL)
        gs = GpuScene(torch_mod, scene)
        recs, cmds = assert_same(run_gpu(torch_mod, engine, gs, ci), run_oracle(oracle, scene, ci))
        assert len(cmds) > 0


def test_dense_and_sparse_tiles(torch_mod, engine, oracle):
    """Wave tiles with more survivors than the payload holds (emit rebuilds them from the
    ballots) next to sparse and empty ones, in one list."""
    scene = sc.make_scene(12, 900, meshlets_per_mesh=(1, 64))
    m = scene.meshlets
    third = len(m) // 3
    m["cone_cutoff"][:third] = 127          # never cone-culled: dense tiles
    m["cone_cutoff"][2 * third:] = -128     # nearly always cone-culled: empty tiles
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, np.zeros((0, 4), np.float32), alpha_mode_flag=L.ALPHA_ALL)
    gs = GpuScene(torch_mod, scene)
    recs, cmds = assert_same(run_gpu(torch_mod, engine, gs, ci), run_oracle(oracle, scene, ci))
    assert len(cmds) > third // 2
    # capacity cut inside a dense tile: the written prefix is identical, the overflow is reported
    cap = len(cmds) - 77
    g = run_gpu(torch_mod, engine, gs, ci, draw_cap=cap)
    o = run_oracle(oracle, scene, ci, draw_cap=cap)
    assert_same(g, o)
    with pytest.raises(Exception):
        engine.status()


def test_empty_inputs(torch_mod, engine, oracle):
    scene = sc.make_scene(6, 40)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, cam.planes)
    gs = GpuScene(torch_mod, scene)
    # entity_draw_count = 0: only the {0,1,1} clear happens; meshlet cull then writes count = 0
    gpu = run_gpu(torch_mod, engine, gs, ci, entity_draw_count=0)
    assert list(gpu[0][:12].view(np.uint32)) == [0, 1, 1]
    assert int(gpu[1][:4].view(np.uint32)[0]) == 0
    # in-buffer count smaller than the dispatch size bounds the work (entity_cull.comp:106)
    gs.draws = dev(torch_mod, scene.entity_draw_buffer(count=17))
    disp, draw, _, _ = run_gpu(torch_mod, engine, gs, ci)
    od, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(count=17), scene.entity_draw_count, scene.mesh_infos,
                                  scene.entities, scene.max_dispatches() + 8)
    assert np.array_equal(disp[:12 + 16 * int(od[:4].view(np.uint32)[0])], od[:12 + 16 * int(od[:4].view(np.uint32)[0])])
    # everything behind the camera: zero survivors
    ci_none = sc.make_cull_info(sc.translation(0, 0, 1e6), cam.planes)
    gpu = run_gpu(torch_mod, engine, GpuScene(torch_mod, scene), ci_none)
    assert list(gpu[0][:12].view(np.uint32)) == [0, 1, 1] and int(gpu[1][:4].view(np.uint32)[0]) == 0


# ----------------------------------------------------------------------------- two-pass HiZ
@pytest.mark.parametrize("size", [(320, 180), (1920, 1080)])
def test_two_pass_sequence(torch_mod, engine, oracle, size):
    torch = torch_mod
    W, H = size
    scene = sc.make_scene(11, 2000, n_meshes=300, extent=(60.0, 12.0, 60.0))
    cam = sc.default_camera()
    rng = np.random.default_rng(5)
    evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    gs = GpuScene(torch, scene)
    ci1 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=1)
    assert_same(run_gpu(torch, engine, gs, ci1, evis, mvis), run_oracle(oracle, scene, ci1, evis, mvis))
    depth = sc.make_depth(9, W, H, cam)
    opyr, d = oracle.depth_reduce(depth, W, H)
    gpyr = torch.zeros(d.total_texels, dtype=torch.float32, device="cuda")
    engine.depth_reduce(dev(torch, depth), W, H, gpyr)
    torch.cuda.synchronize()
    assert np.array_equal(host(gpyr, np.uint32), opyr.view(np.uint32)), "depth pyramid differs"
    ci2 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
    ps = (d.width, d.height)
    recs, cmds = assert_same(run_gpu(torch, engine, gs, ci2, evis, mvis, opyr, ps),
                             run_oracle(oracle, scene, ci2, evis, mvis, opyr, ps))
    assert len(cmds) > 0
    # frame 0 (all-zero bitsets) and entity-only occlusion
    ze, zm = np.zeros_like(evis), np.zeros_like(mvis)
    assert_same(run_gpu(torch, engine, gs, ci2, ze, zm, opyr, ps), run_oracle(oracle, scene, ci2, ze, zm, opyr, ps))
    ci2e = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, meshlet_visibility=False, p00=cam.p00,
                             p11=cam.p11, z_near=cam.z_near)
    assert_same(run_gpu(torch, engine, gs, ci2e, evis, None, opyr, ps),
                run_oracle(oracle, scene, ci2e, evis, None, opyr, ps))
    # pass 1 is list-driven on the device: no meshlet visibility buffer (every meshlet of a drawn entity is a
    # candidate), all bits set (dense lists, several chunks per wave tile), ~3 % set and none set
    ci1e = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=1, meshlet_visibility=False)
    assert_same(run_gpu(torch, engine, gs, ci1e, evis, None), run_oracle(oracle, scene, ci1e, evis, None))
    ones_e = np.full_like(evis, 0xFFFFFFFF)
    for m in (np.full_like(mvis, 0xFFFFFFFF), mvis & rng.integers(0, 2 ** 32, len(mvis), dtype=np.uint32)
              & rng.integers(0, 2 ** 32, len(mvis), dtype=np.uint32)
              & rng.integers(0, 2 ** 32, len(mvis), dtype=np.uint32)
              & rng.integers(0, 2 ** 32, len(mvis), dtype=np.uint32), zm):
        assert_same(run_gpu(torch, engine, gs, ci1, ones_e, m), run_oracle(oracle, scene, ci1, ones_e, m))


# ... (1280, 720): whole tiles and a 32 x 16 tail; (3840, 2160): the largest in-launch tail (level 5 of 64 x 64);
# (4100, 40): level 5 of 128 x 1 — the second launch
@pytest.mark.parametrize("size", [(96, 54), (130, 40), (64, 64), (17, 9), (200, 3), (2560, 1080), (1, 1), (1280, 720),
                                  (3840, 2160), (4100, 40)])
def test_depth_reduce(torch_mod, engine, oracle, size):
    torch = torch_mod
    W, H = size
    cam = sc.default_camera(aspect=W / H)
    depth = sc.make_depth(5, W, H, cam, n_occluders=12)
    opyr, d = oracle.depth_reduce(depth, W, H)
    gpyr = torch.full((d.total_texels,), -1.0, dtype=torch.float32, device="cuda")
    engine.depth_reduce(dev(torch, depth), W, H, gpyr)
    torch.cuda.synchronize()
    assert np.array_equal(host(gpyr, np.uint32), opyr.view(np.uint32))


def test_depth_reduce_multi_mixed_sizes(torch_mod, engine, oracle):
    """update_multiple_depth_pyramids::<C> (draw_gen.rs:569-628) as ONE launch pair: the main view's pyramid and
    shadow-cascade-sized ones, different sizes in one batch (with and without a tail level) — each bit-exact."""
    torch = torch_mod
    sizes = [(1920, 1080), (2048, 2048), (2048, 2048), (96, 54), (1, 1), (333, 77), (64, 64), (17, 9)]
    items, outs = [], []
    for k, (W, H) in enumerate(sizes):
        cam = sc.default_camera(aspect=W / H)
        depth = sc.make_depth(20 + k, W, H, cam, n_occluders=10)
        opyr, d = oracle.depth_reduce(depth, W, H)
        gpyr = torch.full((d.total_texels,), -1.0, dtype=torch.float32, device="cuda")
        items.append(dict(depth=dev(torch, depth), width=W, height=H, pyramid=gpyr))
        outs.append((gpyr, opyr))
    engine.depth_reduce_multi(items)
    torch.cuda.synchronize()
    for k, (gpyr, opyr) in enumerate(outs):
        assert np.array_equal(host(gpyr, np.uint32), opyr.view(np.uint32)), f"pyramid {k} {sizes[k]} differs"
    from orbit_amd._lib import E_CAPACITY, OrbitError

    with pytest.raises(OrbitError) as ei:
        engine.depth_reduce_multi(items + items[:1])  # nine pyramids
    assert ei.value.code == E_CAPACITY


@pytest.mark.parametrize("size", [(640, 360), (130, 40)])
def test_pyramid_of_separate_per_mip_images(torch_mod, engine, oracle, size):
    """The pyramid as the reference holds it — an image with a view per mip (draw_gen.rs:456-482) — exported to HIP as
    one linear image per level with its own base address and padded row pitch (OrbitDepthPyramidLevel), and a depth
    buffer with a row pitch: orbit_depth_reduce_multi writes the levels, both cull stages sample them in pass 2 through
    the device-side level table.  Level contents == the oracle's packed chain; cull outputs == the packed run's."""
    torch = torch_mod
    W, H = size
    cam = sc.default_camera()
    depth = sc.make_depth(9, W, H, cam)
    opyr, d = oracle.depth_reduce(depth, W, H)
    dpitch = W + 13
    dbuf = torch.full((H, dpitch), float("nan"), dtype=torch.float32, device="cuda")
    dbuf[:, :W] = torch.from_numpy(depth).cuda()
    levels, table = [], np.zeros(d.mip_levels, dtype=[("texels", "<u8"), ("row_pitch", "<u4"), ("_pad", "<u4")])
    for k in range(d.mip_levels):
        pitch = d.mip_width[k] + (7 if k % 2 else 32)
        t = torch.full((d.mip_height[k], pitch), -7.0, dtype=torch.float32, device="cuda")
        levels.append((t, pitch))
        table[k] = (t.data_ptr(), pitch, 0)
    engine.depth_reduce_multi([dict(depth=dbuf, width=W, height=H, depth_row_pitch=dpitch, levels=levels)])
    torch.cuda.synchronize()
    for k, (t, pitch) in enumerate(levels):
        w, h, off = d.mip_width[k], d.mip_height[k], d.mip_offset[k]
        got = t.cpu().numpy()
        assert np.array_equal(got[:, :w].view(np.uint32), opyr[off:off + w * h].reshape(h, w).view(np.uint32)), k
        assert np.all(got[:, w:] == -7.0)  # the padding of every row is left alone
    # pass 2 of both cull stages through the level table
    scene = sc.make_scene(11, 1500, n_meshes=200, extent=(60.0, 12.0, 60.0))
    rng = np.random.default_rng(6)
    evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    gs = GpuScene(torch, scene)
    ci2 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
    ps = (d.width, d.height)
    ref = run_oracle(oracle, scene, ci2, evis, mvis, opyr, ps)
    table_d = dev(torch, table)
    disp_cap, draw_cap = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device="cuda")
    draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device="cuda")
    evis_d, mvis_d = dev(torch, evis), dev(torch, mvis)
    engine.entity_cull(ci2, gs.draws, gs.mesh_infos, disp, gs.entities, scene.entity_draw_count, disp_cap,
                       visibility_buffer=evis_d, depth_pyramid_size=ps, depth_pyramid_levels=table_d)
    engine.meshlet_cull(ci2, disp, gs.meshlets, draw, gs.entities, gs.materials, disp_cap, draw_cap,
                        meshlet_visibility_buffer=mvis_d, depth_pyramid_size=ps, depth_pyramid_levels=table_d)
    torch.cuda.synchronize()
    engine.status()
    _, cmds = assert_same((host(disp), host(draw), host(evis_d, np.uint32), host(mvis_d, np.uint32)), ref)
    assert len(cmds) > 0


@pytest.mark.parametrize("seed", range(max(16, int(os.environ.get("ORBIT_FUZZ_SEEDS", "16")) // 4)))
def test_randomized_depth_reduce(torch_mod, engine, oracle, seed):
    """Random screen sizes down to 1 x 1 and extreme aspect ratios; arbitrary float bit patterns in the depth buffer
    (NaN, inf, negatives, denormals): the ReduceMin selects must pick the same words as the oracle."""
    torch = torch_mod
    rng = np.random.default_rng(7000 + seed)
    W = int(rng.choice([1, 2, 3, 17, 64, 255, 257, int(rng.integers(1, 700))]))
    H = int(rng.choice([1, 2, 5, 33, 128, 129, int(rng.integers(1, 400))]))
    depth = rng.random((H, W), dtype=np.float32)
    if seed % 2:
        bits = rng.integers(0, 2 ** 32, (H, W), dtype=np.uint32)
        pick = rng.random((H, W)) < 0.3
        depth = np.where(pick, bits.view(np.float32), depth).astype(np.float32)
    opyr, d = oracle.depth_reduce(depth, W, H)
    gpyr = torch.full((d.total_texels,), -1.0, dtype=torch.float32, device="cuda")
    engine.depth_reduce(dev(torch, depth), W, H, gpyr)
    torch.cuda.synchronize()
    assert np.array_equal(host(gpyr, np.uint32), opyr.view(np.uint32)), (W, H)


@pytest.mark.parametrize("size", [(1920, 1080), (1500, 900), (1280, 720)])
def test_depth_reduce_of_arbitrary_bits_at_screen_sizes(torch_mod, engine, oracle, size):
    """NaN, inf, negative and denormal texels through the paths only whole tiles and a 1024^2 pyramid take: levels 1-5 by
    lane exchanges and the one-wave tail place their operands as reduce4 does — which of two NaNs (or a NaN and a number)
    a ReduceMin keeps depends on the order."""
    torch = torch_mod
    W, H = size
    rng = np.random.default_rng(W)
    depth = rng.random((H, W), dtype=np.float32)
    bits = rng.integers(0, 2 ** 32, (H, W), dtype=np.uint32)
    depth = np.where(rng.random((H, W)) < 0.3, bits.view(np.float32), depth).astype(np.float32)
    opyr, d = oracle.depth_reduce(depth, W, H)
    gpyr = torch.full((d.total_texels,), -1.0, dtype=torch.float32, device="cuda")
    engine.depth_reduce(dev(torch, depth), W, H, gpyr)
    torch.cuda.synchronize()
    assert np.array_equal(host(gpyr, np.uint32), opyr.view(np.uint32))


def test_orthographic(torch_mod, engine, oracle):
    torch = torch_mod
    scene = sc.make_scene(13, 700, extent=(20.0, 6.0, 20.0))
    proj = sc.orthographic_rh(-25, 25, -25, 25, 0.1, 80.0)
    planes = np.concatenate([sc.frustum_planes(proj, 6), sc.default_camera().planes[:3]])
    view = sc.translation(0.0, 0.0, -40.0)
    gs = GpuScene(torch, scene)
    ci = sc.make_cull_info(view, planes, projection_type=1, lod_target=(1.0, 2.0, 3.0))
    assert_same(run_gpu(torch, engine, gs, ci), run_oracle(oracle, scene, ci))
    W = H = 128
    depth = np.full((H, W), 0.5, dtype=np.float32)
    depth[:, : W // 2] = 0.9
    opyr, d = oracle.depth_reduce(depth, W, H)
    evis = np.zeros((scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = np.zeros(scene.vis_words, dtype=np.uint32)
    ci2 = sc.make_cull_info(view, planes, occlusion_pass=2, projection_type=1, p00=2.0 / 50.0, p11=2.0 / 50.0,
                            z_near=0.1, z_far=80.0)
    ps = (d.width, d.height)
    assert_same(run_gpu(torch, engine, gs, ci2, evis, mvis, opyr, ps),
                run_oracle(oracle, scene, ci2, evis, mvis, opyr, ps))


def test_capacity_overflow(torch_mod, engine, oracle):
    from orbit_amd._lib import E_CAPACITY, OrbitError

    torch = torch_mod
    scene = sc.make_scene(14, 200)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, np.zeros((0, 4), np.float32), alpha_mode_flag=L.ALPHA_ALL)
    gs = GpuScene(torch, scene)
    disp, draw, _, _ = run_gpu(torch, engine, gs, ci, disp_cap=50, draw_cap=100)
    ref = run_oracle(oracle, scene, ci, disp_cap=50, draw_cap=100)
    assert ref[4] > 0 and ref[5] > 0
    assert np.array_equal(disp[:12 + 16 * 50], ref[0][:12 + 16 * 50])
    assert np.array_equal(draw[:4 + 28 * 100], ref[1][:4 + 28 * 100])
    assert np.all(disp[12 + 16 * 50:] == 0xAB) and np.all(draw[4 + 28 * 100:] == 0xCD)  # no OOB write
    with pytest.raises(OrbitError) as ei:
        engine.status()
    assert ei.value.code == E_CAPACITY
    engine.status()  # latch cleared


def test_argument_errors(torch_mod, engine):
    from orbit_amd._lib import E_MISSING, E_PLANES, OrbitError

    torch = torch_mod
    scene = sc.make_scene(15, 10)
    cam = sc.default_camera()
    gs = GpuScene(torch, scene)
    disp = torch.zeros(12 + 16 * 64, dtype=torch.uint8, device="cuda")
    ci = sc.make_cull_info(cam.view, cam.planes)
    ci["cull_plane_count"] = 13  # assert!(planes <= 12), draw_gen.rs:334
    with pytest.raises(OrbitError) as ei:
        engine.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, 10, 64)
    assert ei.value.code == E_PLANES
    ci1 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=1)
    with pytest.raises(OrbitError) as ei:  # .unwrap() on a missing visibility buffer, draw_gen.rs:125-128
        engine.entity_cull(ci1, gs.draws, gs.mesh_infos, disp, gs.entities, 10, 64)
    assert ei.value.code == E_MISSING


def test_large_scene_properties(torch_mod, engine, oracle):
    """A scene too large for the numpy cross-check: oracle equality plus size-independent
    properties (idempotence, determinism across launches, count == popcount)."""
    torch = torch_mod
    scene = sc.make_scene(21, 60_000, n_meshes=2000, meshlets_per_mesh=(8, 64), extent=(400.0, 60.0, 400.0),
                          layout="lattice")
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, cam.planes)
    gs = GpuScene(torch, scene)
    a = run_gpu(torch, engine, gs, ci)
    b = run_gpu(torch, engine, gs, ci)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])  # deterministic sequence
    recs, cmds = assert_same(a, run_oracle(oracle, scene, ci))
    assert len(cmds) > 10_000
    # commands are in canonical (record, lane) order: meshlet_index ascending within an entity run
    key = cmds["cmd_first_instance"].astype(np.int64) * (1 << 32) + cmds["meshlet_index"]
    runs = np.flatnonzero(np.diff(cmds["cmd_first_instance"]) != 0)
    assert np.all(np.diff(key)[np.setdiff1d(np.arange(len(key) - 1), runs)] > 0)


# ----------------------------------------------------------------------------- light clusters
def cluster_inputs(oracle, seed, W, H, n_lights, tile_px=8, z_slices=32, far=200.0, samples=1, point_fraction=1.0):
    cam = sc.default_camera(aspect=W / H)
    depth = sc.make_depth(seed, W, H, cam, n_occluders=40, ground=(seed % 2 == 0))
    if samples > 1:
        jit = sc.rnd_range(seed, 90, np.arange(W * H * samples), 0.9, 1.1).reshape(H, W, samples)
        depth = (depth[:, :, None] * jit).astype(np.float32)
    cx, cy = -(-W // tile_px), -(-H // tile_px)
    zs, zb = oracle.cluster_grid_info(cam.z_near, far, z_slices)
    push = np.zeros((), dtype=L.MARK_ACTIVE_PUSH)
    push["cluster_count"] = (cx, cy, z_slices)
    push["tile_size_px"] = tile_px
    push["screen_size"] = (W, H)
    push["z_near"], push["z_far"], push["z_scale"], push["z_bias"] = cam.z_near, far, zs, zb
    push["depth_buffer_sample_count"] = samples
    info = np.zeros((), dtype=L.CLUSTER_CULL_INFO)
    info["world_to_view_matrix"] = sc.mat4_cols(cam.view)
    info["screen_to_view_matrix"] = sc.mat4_cols(np.linalg.inv(cam.proj.astype(np.float64)).astype(np.float32))
    info["cluster_count"] = (cx, cy, z_slices)
    info["tile_size_px"] = tile_px
    info["screen_size"] = (W, H)
    info["z_near"], info["z_far"] = cam.z_near, far
    info["global_light_count"] = n_lights
    lights = sc.make_lights(seed, n_lights, point_fraction=point_fraction)
    return push, depth, info, lights


@pytest.mark.parametrize("W,H,n_lights,tile,zs,samples,pf", [
    (320, 180, 200, 8, 32, 1, 1.0), (333, 187, 64, 16, 24, 1, 0.9), (256, 128, 300, 8, 32, 4, 1.0),
    (1920, 1080, 1000, 8, 32, 1, 1.0), (64, 64, 0, 8, 8, 1, 1.0), (200, 120, 40, 64, 32, 1, 1.0),
    # tile sizes that are not powers of two (the mark launch's lane geometry has a shift form and a division form), with
    # and without several samples; a tile of fewer than 64 samples and one of more than 64
    (330, 190, 100, 12, 32, 1, 1.0), (200, 100, 50, 6, 16, 2, 1.0), (150, 90, 30, 5, 32, 1, 0.9), (97, 61, 20, 3, 7, 4, 1.0)])
def test_light_clusters(torch_mod, engine, oracle, W, H, n_lights, tile, zs, samples, pf):
    _light_clusters(torch_mod, engine, oracle, W, H, n_lights, tile, zs, samples, pf, poison=False)


def test_light_clusters_with_non_finite_inputs(torch_mod, engine, oracle):
    """NaN / inf / negative depth samples and non-finite light positions and radii: cluster AABBs with NaN
    components (the union-box light filter must not apply to those blocks), lights that every comparison accepts."""
    _light_clusters(torch_mod, engine, oracle, 320, 180, 300, 8, 32, 1, 0.95, poison=True)


def test_light_clusters_many_lights_and_saturated_clusters(torch_mod, engine, oracle):
    """More lights than one filter tile holds (the device filters 4096 lights per step) and radii large enough that
    clusters exceed the 256-light cap (light_culling.comp:135): counts saturate, lists keep the first 256 in order."""
    _light_clusters(torch_mod, engine, oracle, 192, 108, 9000, 8, 32, 1, 0.98, poison=False, radius_scale=1.0)
    _light_clusters(torch_mod, engine, oracle, 128, 72, 5000, 16, 16, 1, 1.0, poison=False, radius_scale=12.0)


def test_light_clusters_config4_full_size(torch_mod, engine, oracle):
    """BASELINE.json configs[3] at its full size: 1920x1080 depth, 10 000 point lights with
    outer_r = sqrt(I / 0.25), default cluster settings => 240 x 135 x 32 = 1 036 800 clusters, 8-px tiles, far 200.
    Tile masks, depth bounds, the compacted list, every cluster's (offset, count) and the ascending light index
    lists against the oracle (mark_active.comp, active_cluster_compaction.comp, light_culling.comp:121-151)."""
    _light_clusters(torch_mod, engine, oracle, 1920, 1080, 10_000, 8, 32, 1, 1.0, poison=False)


def _light_clusters(torch_mod, engine, oracle, W, H, n_lights, tile, zs, samples, pf, poison, radius_scale=None,
                    lights_override=None):
    torch = torch_mod
    push, depth, info, lights = cluster_inputs(oracle, 4, W, H, n_lights, tile, zs, samples=samples, point_fraction=pf)
    if lights_override is not None:  # LightData[] produced elsewhere (tests/test_scene_gpu.py)
        assert len(lights_override) == n_lights
        lights = np.ascontiguousarray(lights_override, dtype=L.LIGHT)
    if radius_scale is not None:
        lights["outer_radius"] *= np.float32(radius_scale)
    if poison:
        rng = np.random.default_rng(4)
        flat = depth.reshape(-1)
        for val in (np.nan, np.inf, -np.inf, -0.25, 0.0, 1.0, 3.0):
            flat[rng.integers(0, len(flat), 200)] = val
        if len(lights) > 22:
            for k, val in enumerate((np.nan, np.inf, -np.inf)):
                lights["position"][10 + k, k % 3] = val
            lights["outer_radius"][20], lights["outer_radius"][21], lights["outer_radius"][22] = np.nan, np.inf, 0.0
    cc = [int(v) for v in push["cluster_count"]]
    total = cc[0] * cc[1] * cc[2]
    # mark
    om, ob = oracle.cluster_mark(push, depth)
    gm = torch.full((cc[0] * cc[1],), 0x7FFFFFFF, dtype=torch.int32, device="cuda")
    gb = torch.full((total, 2), 0x7FFFFFFF, dtype=torch.int32, device="cuda")
    engine.cluster_mark(push, dev(torch, depth), gm, gb)
    torch.cuda.synchronize()
    assert np.array_equal(host(gm, np.uint32), om), "tile slice masks differ"
    assert np.array_equal(host(gb, np.uint32).reshape(-1, 2), ob), "cluster depth bounds differ"
    # compact
    cap = cc[0] * cc[1] * max(4, cc[2])  # cluster.rs:53-56
    ou, dropped = oracle.cluster_compact(cc, om, cap)
    assert dropped == 0
    gu = torch.full((L.COMPACT_HEADER + 4 * cap,), 0xEE, dtype=torch.uint8, device="cuda")
    engine.cluster_compact(cc, gm, gu, cap)
    torch.cuda.synchronize()
    n_active = int(ou[12:16].view(np.uint32)[0])
    assert np.array_equal(host(gu)[:16 + 4 * n_active], ou[:16 + 4 * n_active]), "compacted cluster list differs"
    # assign
    lcap = n_active * 256 + 16
    ol, oimg, dropped = oracle.cluster_assign(info, ou, ob, lights, lcap, total)
    assert dropped == 0
    if radius_scale is not None and radius_scale > 1.0:
        assert int(oimg[:, 1].max()) == 256  # the cap is reached
    gl = torch.full((L.LIGHT_INDEX_HEADER + 4 * lcap,), 0xEE, dtype=torch.uint8, device="cuda")
    gimg = torch.zeros((total, 2), dtype=torch.int32, device="cuda")
    engine.cluster_assign(info, gu, gb, dev(torch, lights) if n_lights else None, gl, lcap, gimg)
    torch.cuda.synchronize()
    n_idx = int(ol[:4].view(np.uint32)[0])
    assert int(host(gl)[:4].view(np.uint32)[0]) == n_idx
    assert np.array_equal(host(gl)[:4 + 4 * n_idx], ol[:4 + 4 * n_idx]), "cluster light index lists differ"
    assert np.array_equal(host(gimg, np.uint32).reshape(-1, 2), oimg), "(offset,count) image differs"
    engine.status()
    if n_lights >= 200:
        assert n_idx > 0 and n_active > 0
    if n_idx > 64:  # the reference's index list can overflow (cluster.rs:546-553): cut it, nothing written past, flagged
        from orbit_amd._lib import E_CAPACITY, OrbitError

        cut = n_idx - 37
        ol3, oimg3, dropped3 = oracle.cluster_assign(info, ou, ob, lights, cut, total)
        gl3 = torch.full((L.LIGHT_INDEX_HEADER + 4 * cut + 64,), 0xEE, dtype=torch.uint8, device="cuda")
        gimg3 = torch.zeros_like(gimg)
        engine.cluster_assign(info, gu, gb, dev(torch, lights), gl3, cut, gimg3)
        torch.cuda.synchronize()
        assert dropped3 > 0
        assert np.array_equal(host(gl3)[:4 + 4 * cut], ol3[:4 + 4 * cut]), "cut light index list differs"
        assert bool((host(gl3)[4 + 4 * cut:] == 0xEE).all())
        assert np.array_equal(host(gimg3, np.uint32).reshape(-1, 2), oimg3)
        with pytest.raises(OrbitError) as ei:
            engine.status()
        assert ei.value.code == E_CAPACITY
    # compute_clusters: the same three stages behind one call produce the same five buffers
    gm2, gb2 = torch.full_like(gm, 0x55), torch.full_like(gb, 0x55)
    gu2, gl2, gimg2 = torch.full_like(gu, 0xEE), torch.full_like(gl, 0xEE), torch.zeros_like(gimg)
    engine.compute_clusters(push, info, dev(torch, depth), dev(torch, lights) if n_lights else None, gm2, gb2, gu2, cap,
                            gl2, lcap, gimg2)
    torch.cuda.synchronize()
    engine.status()
    assert torch.equal(gm2, gm) and torch.equal(gb2, gb) and torch.equal(gimg2, gimg)
    assert torch.equal(gu2[:16 + 4 * n_active], gu[:16 + 4 * n_active])
    assert torch.equal(gl2[:4 + 4 * n_idx], gl[:4 + 4 * n_idx])


# ----------------------------------------------------------------------------- mesh-shading (task) path
@pytest.mark.parametrize("occlusion_pass,ortho", [(0, False), (1, False), (2, False), (2, True), (0, True)])
def test_task_records(torch_mod, engine, oracle, occlusion_pass, ortho):
    """orbit_meshlet_task_cull against the restated task shader (forward_depth_prepass.task:115-257)."""
    torch = torch_mod
    scene = sc.make_scene(61 + occlusion_pass, 900, n_meshes=140, lods=2, meshlets_per_mesh=(1, 70),
                          extent=(20.0, 6.0, 20.0))
    rng = np.random.default_rng(61)
    evis = mvis = pyr = None
    ps = (0, 0)
    cam = sc.default_camera(rot=(0.4, 0.9))
    if ortho:  # the shadow-cascade style view of test_orthographic
        proj = sc.orthographic_rh(-25, 25, -25, 25, 0.1, 80.0)
        view, planes = sc.translation(0.0, 0.0, -40.0), sc.frustum_planes(proj, 6)
        kw = dict(projection_type=1)
        if occlusion_pass == 2:
            kw.update(p00=2.0 / 50.0, p11=2.0 / 50.0, z_near=0.1, z_far=80.0)
    else:
        view, planes = cam.view, cam.planes
        kw = dict(p00=cam.p00, p11=cam.p11, z_near=cam.z_near) if occlusion_pass == 2 else {}
    ci = sc.make_cull_info(view, planes, occlusion_pass=occlusion_pass, noskip_alphamode=L.ALPHA_MASKED, **kw)
    if occlusion_pass:
        evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
        mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    if occlusion_pass == 2:
        W, H = 320, 180
        pyr, d = oracle.depth_reduce(sc.make_depth(5, W, H, cam), W, H)
        ps = (d.width, d.height)
    gs = GpuScene(torch, scene)
    cap_d = scene.max_dispatches() + 8
    disp = torch.full((L.DISPATCH_HEADER + 16 * cap_d + 64,), 0xAB, dtype=torch.uint8, device="cuda")
    evis_d = None if evis is None else dev(torch, evis)
    mvis_d = None if mvis is None else dev(torch, mvis)
    pyr_d = None if pyr is None else dev(torch, pyr)
    engine.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, scene.entity_draw_count, cap_d,
                       visibility_buffer=evis_d, depth_pyramid=pyr_d, depth_pyramid_size=ps)
    task = torch.full((44 * cap_d + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    engine.meshlet_task_cull(ci, disp, gs.meshlets, task, gs.entities, gs.materials, cap_d,
                             meshlet_visibility_buffer=mvis_d, depth_pyramid=pyr_d, depth_pyramid_size=ps)
    torch.cuda.synchronize()
    engine.status()
    odisp, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                     scene.entities, cap_d, evis, pyr, ps)
    orecs, omv = oracle.meshlet_task_cull(ci, odisp, scene.meshlets, scene.entities, scene.materials, mvis, pyr, ps)
    n = len(orecs)
    assert n > 0 and int(orecs["task_mesh_count"].sum()) > 0
    got = host(task)[:44 * n].view(L.MESH_TASK_RECORD)
    assert np.array_equal(got.view(np.uint8), orecs.view(np.uint8)), "task records differ"
    assert bool((host(task)[44 * n:] == 0xEE).all())  # nothing written past the records
    if mvis is not None:
        assert np.array_equal(host(mvis_d, np.uint32), omv), "meshlet visibility words differ"


# ----------------------------------------------------------------------------- LOD arithmetic edge cases
@pytest.mark.parametrize("lod_base,lod_step,target", [
    (16.0, 1.0, (0.0, 0.0, 0.0)),      # log2(lod_step) == 0: division by zero in the LOD formula (entity_cull.comp:206)
    (0.0, 2.0, (0.0, 0.0, 0.0)),       # lod_base == 0: distance / 0
    (1e-30, 1.0001, (1e20, 0.0, 0.0)), # huge LOD values: uint() saturation
    (16.0, 0.5, (5.0, 5.0, 5.0)),      # lod_step < 1: negative log
    (float("nan"), 2.0, (0.0, 0.0, 0.0)), (16.0, 2.0, (float("inf"), 0.0, float("nan")))])
def test_lod_selection_edge_cases(torch_mod, engine, oracle, lod_base, lod_step, target):
    scene = sc.make_scene(81, 500, n_meshes=60, lods=5, meshlets_per_mesh=(1, 50))
    cam = sc.default_camera(rot=(0.1, 0.2))
    for lod_range in ((0, 8), (1, 3), (4, 5)):
        ci = sc.make_cull_info(cam.view, cam.planes, lod_base=lod_base, lod_step=lod_step, lod_target=target,
                               lod_range=lod_range)
        gs = GpuScene(torch_mod, scene)
        assert_same(run_gpu(torch_mod, engine, gs, ci), run_oracle(oracle, scene, ci))


# ----------------------------------------------------------------------------- shadow cascades
@pytest.mark.parametrize("cascade", range(4))
def test_shadow_cascade_cull(torch_mod, engine, oracle, cascade):
    """The renderer's own cascade math (shadow_renderer.rs:466-706 through the host mirror) produces the orthographic
    CullInfo with up to 11 planes and a LOD window per cascade; both cull stages with it, against the oracle."""
    from orbit_amd import passes

    scene = sc.make_scene(71, 1500, n_meshes=200, lods=4, meshlets_per_mesh=(1, 60), extent=(30.0, 6.0, 30.0))
    q = lambda pitch, yaw: (float(np.cos(pitch / 2) * np.sin(yaw / 2)), float(np.sin(pitch / 2) * np.cos(yaw / 2)),
                            float(-np.sin(pitch / 2) * np.sin(yaw / 2)), float(np.cos(pitch / 2) * np.cos(yaw / 2)))
    hci, _, _ = passes.shadow_cascade(q(0.4, -0.9), (2.0, 3.0, 8.0), q(0.3, -0.1), float(np.float32(np.pi / 2)), 0.01,
                                      16.0 / 9.0, cascade, lod_base=6.0)
    assert hci.plane_count >= 6
    ci = passes.cull_info_to_gpu(hci)
    assert int(ci["projection_type"]) == 1 and int(ci["cull_plane_count"]) == hci.plane_count
    gs = GpuScene(torch_mod, scene)
    recs, cmds = assert_same(run_gpu(torch_mod, engine, gs, ci), run_oracle(oracle, scene, ci))
    if cascade >= 2:
        assert len(cmds) > 0  # the far cascades cover the scene


# ----------------------------------------------------------------------------- randomized parameter sweep
# ORBIT_FUZZ_SEEDS=n widens the sweep for a soak run (default 24 seeds)
@pytest.mark.parametrize("seed", range(int(os.environ.get("ORBIT_FUZZ_SEEDS", "24"))))
def test_randomized_parameters(torch_mod, engine, oracle, seed):
    """Differential sweep over the CullInfo parameter space: every plane count 0..12 (the device shapes its plane
    batches by the count), all passes, both projections, alpha masks, LOD windows, with and without meshlet
    visibility; random planes through the scene so that plane culling actually splits it."""
    torch = torch_mod
    rng = np.random.default_rng(1000 + seed)
    W, H = 256, 144
    scene = sc.make_scene(100 + seed, int(rng.integers(40, 500)), n_meshes=int(rng.integers(5, 60)),
                          lods=int(rng.integers(1, 4)), meshlets_per_mesh=(1, int(rng.integers(2, 90))),
                          extent=(30.0, 10.0, 30.0))
    cam = sc.default_camera(rot=(float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-3.1, 3.1))))
    n_planes = seed % 13
    base = list(cam.planes) if rng.random() < 0.5 else []
    planes = []
    for i in range(n_planes):
        if i < len(base):
            planes.append(base[i])
        else:  # a random view-space plane close to the scene centre
            nrm = rng.normal(size=3)
            nrm /= np.linalg.norm(nrm)
            planes.append(np.array([*nrm, rng.uniform(10.0, 45.0)], dtype=np.float32))
    planes = np.asarray(planes, dtype=np.float32).reshape(-1, 4)
    ortho = bool(rng.random() < 0.3)
    occlusion_pass = int(rng.integers(0, 3))
    mv = bool(rng.random() < 0.7)
    kw = dict(occlusion_pass=occlusion_pass, projection_type=1 if ortho else 0,
              alpha_mode_flag=int(rng.integers(0, 8)), noskip_alphamode=int(rng.integers(0, 8)), meshlet_visibility=mv,
              lod_base=float(rng.uniform(2.0, 40.0)), lod_step=float(rng.uniform(1.3, 3.0)),
              lod_target=tuple(float(v) for v in rng.uniform(-10, 10, 3)))
    lo = int(rng.integers(0, 3))
    kw["lod_range"] = (lo, int(rng.integers(lo + 1, 9)))
    if occlusion_pass == 2:
        kw.update(p00=2.0 / 40.0, p11=2.0 / 40.0, z_near=0.1, z_far=90.0) if ortho else kw.update(
            p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
    ci = sc.make_cull_info(cam.view, planes, **kw)
    evis = mvis = pyr = None
    ps = (0, 0)
    if occlusion_pass:
        evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
        if mv:
            mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
            if rng.random() < 0.5:
                mvis &= rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    if occlusion_pass == 2:
        pyr, d = oracle.depth_reduce(sc.make_depth(seed, W, H, cam), W, H)
        ps = (d.width, d.height)
    gs = GpuScene(torch, scene)
    mc = len(scene.materials) if seed % 2 else 0  # both alpha-mode paths of the kernels
    assert_same(run_gpu(torch, engine, gs, ci, evis, mvis, pyr, ps, material_count=mc),
                run_oracle(oracle, scene, ci, evis, mvis, pyr, ps))
    engine.status()


# ----------------------------------------------------------------------------- graph capture
def test_two_pass_frame_is_graph_capturable(torch_mod, engine, oracle):
    """Every cull entry point only enqueues on the caller's stream (no allocation, no host sync), so a whole
    two-pass frame can be captured into a HIP graph and replayed: same bytes as the eager run and as the oracle."""
    torch = torch_mod
    W, H = 320, 180
    scene = sc.make_scene(31, 600, n_meshes=120, extent=(40.0, 10.0, 40.0))
    cam = sc.default_camera()
    rng = np.random.default_rng(31)
    evis0 = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis0 = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    depth = sc.make_depth(9, W, H, cam)
    d = oracle.depth_pyramid_desc(W, H)
    ps = (d.width, d.height)
    ci1 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=1)
    ci2 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
    gs = GpuScene(torch, scene)
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    n = scene.entity_draw_count
    gdepth = dev(torch, depth)
    bufs = dict(disp=torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda"),
                draw1=torch.zeros(L.DRAW_HEADER + 28 * cap_c, dtype=torch.uint8, device="cuda"),
                draw2=torch.zeros(L.DRAW_HEADER + 28 * cap_c, dtype=torch.uint8, device="cuda"),
                evis=dev(torch, evis0), mvis=dev(torch, mvis0),
                pyr=torch.zeros(d.total_texels, dtype=torch.float32, device="cuda"))

    def frame():
        b = bufs
        engine.entity_cull(ci1, gs.draws, gs.mesh_infos, b["disp"], gs.entities, n, cap_d, visibility_buffer=b["evis"])
        engine.meshlet_cull(ci1, b["disp"], gs.meshlets, b["draw1"], gs.entities, gs.materials, cap_d, cap_c,
                            meshlet_visibility_buffer=b["mvis"])
        engine.depth_reduce(gdepth, W, H, b["pyr"])
        engine.entity_cull(ci2, gs.draws, gs.mesh_infos, b["disp"], gs.entities, n, cap_d, visibility_buffer=b["evis"],
                           depth_pyramid=b["pyr"], depth_pyramid_size=ps)
        engine.meshlet_cull(ci2, b["disp"], gs.meshlets, b["draw2"], gs.entities, gs.materials, cap_d, cap_c,
                            meshlet_visibility_buffer=b["mvis"], depth_pyramid=b["pyr"], depth_pyramid_size=ps)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        frame()  # warm-up outside the capture (code objects loaded)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        frame()
    # replay from the original bitsets
    bufs["evis"].copy_(dev(torch, evis0))
    bufs["mvis"].copy_(dev(torch, mvis0))
    g.replay()
    torch.cuda.synchronize()
    engine.status()
    opyr, _ = oracle.depth_reduce(depth, W, H)
    o1 = run_oracle(oracle, scene, ci1, evis0, mvis0)
    o2 = run_oracle(oracle, scene, ci2, evis0, mvis0, opyr, ps)
    n1, c1 = L.draw_buffer_commands(host(bufs["draw1"]))
    on1, oc1 = L.draw_buffer_commands(o1[1])
    assert n1 == on1 and np.array_equal(c1.view(np.uint32), oc1.view(np.uint32))
    n2, c2 = L.draw_buffer_commands(host(bufs["draw2"]))
    on2, oc2 = L.draw_buffer_commands(o2[1])
    assert n2 == on2 and np.array_equal(c2.view(np.uint32), oc2.view(np.uint32))
    assert np.array_equal(host(bufs["evis"], np.uint32), o2[2]) and np.array_equal(host(bufs["mvis"], np.uint32), o2[3])
    # a second replay continues from the bitsets the first one left (frame n+1), like an eager frame would
    g.replay()
    torch.cuda.synchronize()
    o3 = run_oracle(oracle, scene, ci2, o2[2], o2[3], opyr, ps)
    assert np.array_equal(host(bufs["evis"], np.uint32), o3[2]) and np.array_equal(host(bufs["mvis"], np.uint32), o3[3])


@pytest.mark.parametrize("seed", range(max(12, int(os.environ.get("ORBIT_FUZZ_SEEDS", "12")) // 4)))
def test_randomized_clusters(torch_mod, engine, oracle, seed):
    """Random screen sizes (not multiples of the tile), tile sizes, slice counts, MSAA, light counts and radii through
    mark -> compact -> assign."""
    rng = np.random.default_rng(5000 + seed)
    W, H = int(rng.integers(17, 400)), int(rng.integers(9, 240))
    tile = int(2 ** rng.integers(2, 7))
    zs = int(rng.integers(1, 33))
    samples = int(rng.choice([1, 1, 2, 4]))
    n_lights = int(rng.choice([0, 1, 63, 64, 65, 300, 1100, 4100]))
    scale = float(rng.choice([0.3, 1.0, 1.0, 4.0]))
    _light_clusters(torch_mod, engine, oracle, W, H, n_lights, tile, zs, samples, float(rng.uniform(0.8, 1.0)),
                    poison=bool(rng.random() < 0.15), radius_scale=scale if scale <= 1.0 else None)


# ----------------------------------------------------------------------------- sharded ranges
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_ranges_concatenate_to_the_canonical_list(torch_mod, engine, oracle, world):
    """Each rank culls its 32-aligned entity range through orbit_entity_cull_range and holds only its
    meshlet shard (passed under global indices via a shifted base pointer, as bench.py does); the
    rank-order concatenation must be the single-GPU sequence."""
    from orbit_amd.dist import shard_ranges

    torch = torch_mod
    scene = sc.make_scene(23, 700, meshlets_per_mesh=(8, 40), shuffle=False)  # identity map: meshlets follow entities
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, cam.planes)
    gs = GpuScene(torch, scene)
    ref = run_oracle(oracle, scene, ci)
    on, ocmds = L.draw_buffer_commands(ref[1])
    lod0 = scene.mesh_infos["mesh_lods"][:, 0]
    parts = []
    for b, e in shard_ranges(scene.entity_draw_count, world):
        m0 = int(lod0[b, 0]) if b < scene.entity_draw_count else 0
        m1 = int(lod0[e - 1, 0] + lod0[e - 1, 1]) if e > b else m0
        shard = dev(torch, scene.meshlets[m0:m1]) if m1 > m0 else torch.zeros(32, dtype=torch.uint8, device="cuda")
        base = shard.data_ptr() - m0 * 32  # global meshlet indices land inside the shard
        cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
        disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
        draw = torch.zeros(L.DRAW_HEADER + 28 * cap_c, dtype=torch.uint8, device="cuda")
        engine.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, e - b, cap_d, draw_first=b)
        engine.meshlet_cull(ci, disp, base, draw, gs.entities, gs.materials, cap_d, cap_c)
        torch.cuda.synchronize()
        n, cmds = L.draw_buffer_commands(host(draw))
        parts.append(cmds)
    got = np.concatenate(parts)
    assert len(got) == on and np.array_equal(got.view(np.uint32), ocmds.view(np.uint32))


def _expected_visible_records(orecs, ocmds):
    """The record-granular visible list the oracle's outputs imply — one entry per dispatch record, in record order: the
    commands come in (record, lane) order, so every dispatch record takes the run of commands of its entity that fall
    into its meshlet range (none: an entry with mask 0)."""
    out, j = [], 0
    for r in orecs:
        ent, off, cnt = int(r["entity_index"]), int(r["meshlet_offset"]), int(r["meshlet_count"])
        mask = 0
        while j < len(ocmds) and int(ocmds["cmd_first_instance"][j]) == ent and off <= int(ocmds["meshlet_index"][j]) < off + cnt:
            bit = int(ocmds["meshlet_index"][j]) - off
            assert not (mask >> bit) & 1 and (mask >> bit) == 0, "commands of a record must come in ascending lane order"
            mask |= 1 << bit
            j += 1
        out.append((ent, off, mask))
    assert j == len(ocmds)
    return np.array(out, dtype=L.VISIBLE_RECORD) if out else np.zeros(0, dtype=L.VISIBLE_RECORD)


@pytest.mark.parametrize("world", [1, 3])
def test_visible_record_lists_expand_to_the_canonical_list(torch_mod, engine, oracle, world):
    """The 12-B-per-record exchange format: every rank's orbit_meshlet_cull_visible_records list is {entity_index,
    meshlet_offset, should-draw ballot} of each of its dispatch records, in record order (written by the evaluation launch
    itself), with {records, survivors} in the header; the rank-order concatenation, expanded by
    orbit_expand_visible_records, is the single-GPU MeshletDrawCommandBuffer — also when the capacities cut it."""
    from orbit_amd._lib import E_CAPACITY, OrbitError
    from orbit_amd.dist import shard_ranges

    torch = torch_mod
    scene = sc.make_scene(31, 2600, meshlets_per_mesh=(1, 70), lods=2)  # > 1024 records per rank: several blocks
    cam = sc.default_camera(rot=(0.2, 0.4))
    ci = sc.make_cull_info(cam.view, cam.planes, alpha_mode_flag=L.ALPHA_ALL)
    gs = GpuScene(torch, scene)
    ref = run_oracle(oracle, scene, ci)
    _, orecs = L.dispatch_buffer_records(ref[0])
    on, ocmds = L.draw_buffer_commands(ref[1])
    want = _expected_visible_records(orecs, ocmds)
    assert on > 2000 and len(want) == len(orecs) > 1100 and int((want["mask"] == 0).sum()) > 0  # some records have no survivor
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    parts = []
    for b, e in shard_ranges(scene.entity_draw_count, world):
        disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
        vis = torch.full((L.VISIBLE_HEADER + 12 * cap_d + 64,), 0xCD, dtype=torch.uint8, device="cuda")
        engine.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, e - b, cap_d, draw_first=b)
        engine.meshlet_cull_visible_records(ci, disp, gs.meshlets, vis, gs.entities, gs.materials, cap_d, cap_d)
        torch.cuda.synchronize()
        engine.status()
        hv = host(vis)
        n, surv = (int(v) for v in hv[:8].view(np.uint32))
        recs = hv[8:8 + 12 * n].view(L.VISIBLE_RECORD)
        assert surv == sum(bin(int(m)).count("1") for m in recs["mask"]) and bool((hv[8 + 12 * n:] == 0xCD).all())
        parts.append(recs)
    got = np.concatenate(parts)
    assert len(got) == len(want) and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # the gathered list, expanded on "the receiving GPU"
    allrec = np.zeros(L.VISIBLE_HEADER + 12 * len(got), dtype=np.uint8)
    allrec[:4].view(np.uint32)[0] = len(got)
    allrec[8:] = got.view(np.uint8)
    grec = dev(torch, allrec)
    draw = torch.full((L.DRAW_HEADER + 28 * cap_c + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    engine.expand_visible_records(grec, gs.meshlets, draw, cap_c)
    torch.cuda.synchronize()
    engine.status()
    n, cmds = L.draw_buffer_commands(host(draw))
    assert n == on and np.array_equal(cmds.view(np.uint32), ocmds.view(np.uint32))
    assert bool((host(draw)[4 + 28 * on:] == 0xCD).all())
    # capacity cuts: the expansion writes a prefix, the cull keeps the first records; both latch the overflow
    cut = on - 33
    draw2 = torch.full((L.DRAW_HEADER + 28 * cut + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    engine.expand_visible_records(grec, gs.meshlets, draw2, cut)
    torch.cuda.synchronize()
    n2, cmds2 = L.draw_buffer_commands(host(draw2))
    assert n2 == cut and np.array_equal(cmds2.view(np.uint32), ocmds[:cut].view(np.uint32))
    assert bool((host(draw2)[4 + 28 * cut:] == 0xCD).all())
    with pytest.raises(OrbitError) as ei:
        engine.status()
    assert ei.value.code == E_CAPACITY
    # a command capacity far below the list (fewer blocks of the expansion than the list has blocks of records)
    tiny = 700
    draw3 = torch.full((L.DRAW_HEADER + 28 * tiny + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    engine.expand_visible_records(grec, gs.meshlets, draw3, tiny)
    torch.cuda.synchronize()
    n3, cmds3 = L.draw_buffer_commands(host(draw3))
    assert n3 == tiny and np.array_equal(cmds3.view(np.uint32), ocmds[:tiny].view(np.uint32))
    assert bool((host(draw3)[4 + 28 * tiny:] == 0xCD).all())
    with pytest.raises(OrbitError) as ei:
        engine.status()
    assert ei.value.code == E_CAPACITY
    if world == 1:
        rcut = len(want) - 7
        disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
        vis = torch.full((L.VISIBLE_HEADER + 12 * rcut + 64,), 0xCD, dtype=torch.uint8, device="cuda")
        engine.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, scene.entity_draw_count, cap_d)
        engine.meshlet_cull_visible_records(ci, disp, gs.meshlets, vis, gs.entities, gs.materials, cap_d, rcut)
        torch.cuda.synchronize()
        hv = host(vis)
        assert int(hv[:4].view(np.uint32)[0]) == rcut and int(hv[4:8].view(np.uint32)[0]) == on
        assert np.array_equal(hv[8:8 + 12 * rcut].view(np.uint32), want[:rcut].view(np.uint32))
        assert bool((hv[8 + 12 * rcut:] == 0xCD).all())
        with pytest.raises(OrbitError) as ei:
            engine.status()
        assert ei.value.code == E_CAPACITY
        # an empty frame: nothing in view
        away = sc.default_camera(position=(0.0, 0.0, -5000.0))
        ci0 = sc.make_cull_info(away.view, away.planes, alpha_mode_flag=L.ALPHA_ALL)
        engine.entity_cull(ci0, gs.draws, gs.mesh_infos, disp, gs.entities, scene.entity_draw_count, cap_d)
        engine.meshlet_cull_visible_records(ci0, disp, gs.meshlets, vis, gs.entities, gs.materials, cap_d, rcut)
        engine.expand_visible_records(vis, gs.meshlets, draw2, cut)
        torch.cuda.synchronize()
        engine.status()
        assert list(host(vis)[:8].view(np.uint32)) == [0, 0] and int(host(draw2)[:4].view(np.uint32)[0]) == 0


@pytest.mark.parametrize("occlusion_pass", [1, 2])
def test_visible_records_in_the_occlusion_passes(torch_mod, engine, oracle, occlusion_pass):
    """The record list of passes 1 (list-driven evaluation) and 2 (HiZ candidates): the oracle's dispatch records and
    draw commands of the same pass, regrouped by record; the meshlet visibility words are those of the plain cull."""
    torch = torch_mod
    scene = sc.make_scene(71 + occlusion_pass, 900, n_meshes=140, lods=2, meshlets_per_mesh=(1, 70), extent=(20.0, 6.0, 20.0))
    rng = np.random.default_rng(71)
    cam = sc.default_camera(rot=(0.4, 0.9))
    kw = dict(p00=cam.p00, p11=cam.p11, z_near=cam.z_near) if occlusion_pass == 2 else {}
    ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=occlusion_pass, noskip_alphamode=L.ALPHA_MASKED, **kw)
    evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    pyr, ps = None, (0, 0)
    if occlusion_pass == 2:
        pyr, d = oracle.depth_reduce(sc.make_depth(5, 320, 180, cam), 320, 180)
        ps = (d.width, d.height)
    gs = GpuScene(torch, scene)
    cap_d = scene.max_dispatches() + 8
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
    evis_d, mvis_d = dev(torch, evis), dev(torch, mvis)
    pyr_d = None if pyr is None else dev(torch, pyr)
    engine.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, scene.entity_draw_count, cap_d,
                       visibility_buffer=evis_d, depth_pyramid=pyr_d, depth_pyramid_size=ps)
    vis = torch.full((L.VISIBLE_HEADER + 12 * cap_d + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    engine.meshlet_cull_visible_records(ci, disp, gs.meshlets, vis, gs.entities, gs.materials, cap_d, cap_d,
                                        meshlet_visibility_buffer=mvis_d, depth_pyramid=pyr_d, depth_pyramid_size=ps)
    torch.cuda.synchronize()
    engine.status()
    ref = run_oracle(oracle, scene, ci, evis=evis, mvis=mvis, pyr=pyr, psize=ps)
    _, orecs = L.dispatch_buffer_records(ref[0])
    on, ocmds = L.draw_buffer_commands(ref[1])
    want = _expected_visible_records(orecs, ocmds)
    assert on > 0 and len(want) > 0
    hv = host(vis)
    n, surv = (int(v) for v in hv[:8].view(np.uint32))
    assert n == len(want) and surv == on
    assert np.array_equal(hv[8:8 + 12 * n].view(np.uint32), want.view(np.uint32))
    assert bool((hv[8 + 12 * n:] == 0xCD).all())
    if ref[3] is not None:
        assert np.array_equal(host(mvis_d, np.uint32), ref[3]), "meshlet visibility words differ"
    # ... and both products of ONE evaluation (orbit_meshlet_cull_records_and_commands: the list a shard sends and the
    # commands it draws from): the same record list, and the oracle's MeshletDrawCommandBuffer
    mvis_d.copy_(dev(torch, mvis))
    cap_c = scene.lod0_meshlets + 8
    vis2 = torch.full((L.VISIBLE_HEADER + 12 * cap_d + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    draw = torch.full((L.DRAW_HEADER + 28 * cap_c + 64,), 0xEF, dtype=torch.uint8, device="cuda")
    engine.meshlet_cull_records_and_commands(ci, disp, gs.meshlets, vis2, draw, gs.entities, gs.materials, cap_d, cap_d,
                                             cap_c, meshlet_visibility_buffer=mvis_d, depth_pyramid=pyr_d,
                                             depth_pyramid_size=ps)
    torch.cuda.synchronize()
    engine.status()
    assert torch.equal(vis2, vis), "record list differs from orbit_meshlet_cull_visible_records'"
    hd = host(draw)
    assert np.array_equal(hd[:4 + 28 * on], ref[1][:4 + 28 * on]) and bool((hd[4 + 28 * cap_c:] == 0xEF).all())
    if ref[3] is not None:
        assert np.array_equal(host(mvis_d, np.uint32), ref[3]), "meshlet visibility words differ"


def test_null_arguments_are_status_codes_not_crashes(torch_mod):
    """Every entry point of include/orbit_abi.h called with NULL handles / buffers: a status code comes back (the
    reference's unwrap / assert panics, as values across the C boundary), nothing is launched, nothing crashes."""
    import ctypes as C

    from orbit_amd import _lib

    lib = _lib.load()
    caps = _lib.Caps()
    lib.orbit_default_caps(C.byref(caps))
    ctx = C.c_void_p()
    assert lib.orbit_ctx_create(0, C.byref(caps), C.byref(ctx)) == _lib.OK
    try:
        ci = sc.make_cull_info(sc.default_camera().view, sc.default_camera().planes)
        cip = ci.ctypes.data_as(C.c_void_p)
        eb, mb = _lib.EntityCullBufs(), _lib.MeshletCullBufs()  # all-NULL buffer sets
        bad = (_lib.E_INVALID, _lib.E_MISSING)
        some = C.c_void_p(ctx.value)  # any non-NULL pointer for arguments that are checked for NULL only
        one_ptr = (C.c_void_p * 1)(ctx.value)  # ... and a pointer array of one
        calls = [
            lambda: lib.orbit_entity_cull(None, cip, C.byref(eb), 1, None),
            lambda: lib.orbit_entity_cull(ctx, None, C.byref(eb), 1, None),
            lambda: lib.orbit_entity_cull(ctx, cip, None, 1, None),
            lambda: lib.orbit_entity_cull(ctx, cip, C.byref(eb), 1, None),
            lambda: lib.orbit_entity_cull_range(ctx, cip, C.byref(eb), 0, 1, None),
            lambda: lib.orbit_meshlet_cull(ctx, cip, None, None),
            lambda: lib.orbit_meshlet_cull(ctx, cip, C.byref(mb), None),
            lambda: lib.orbit_meshlet_cull(None, cip, C.byref(mb), None),
            lambda: lib.orbit_meshlet_task_cull(ctx, cip, C.byref(mb), None, None),
            lambda: lib.orbit_meshlet_cull_visible_records(ctx, cip, C.byref(mb), None, 16, None),
            lambda: lib.orbit_expand_visible_records(ctx, None, None, None, 16, None),
            lambda: lib.orbit_expand_visible_records(None, some, some, some, 16, None),
            lambda: lib.orbit_depth_reduce(ctx, None, 64, 64, None, None),
            lambda: lib.orbit_depth_reduce(ctx, some, 0, 64, some, None),
            lambda: lib.orbit_depth_pyramid_desc(0, 0, None),
            lambda: lib.orbit_cluster_mark(ctx, None, None, None, None, None),
            lambda: lib.orbit_cluster_compact(ctx, None, None, None, 0, None),
            lambda: lib.orbit_cluster_assign(ctx, None, None, None, None, None, 0, None, None),
            lambda: lib.orbit_compute_clusters(ctx, None, None, None, None, None, None, None, 0, None, 0, None, None),
            lambda: lib.orbit_gather_visible(ctx, None, 0, 1, None, None, 0, None),
            lambda: lib.orbit_ctx_profile_read(ctx, None, None),
            lambda: lib.orbit_ctx_status(None, None, 0),
            lambda: lib.orbit_cull_views(ctx, None, 1, None),
            lambda: lib.orbit_cull_views(None, None, 1, None),
            lambda: lib.orbit_cull_views(ctx, C.cast(C.pointer(_lib.CullView()), C.POINTER(_lib.CullView)), 1, None),
            lambda: lib.orbit_depth_reduce_multi(ctx, None, 1, None),
            lambda: lib.orbit_depth_reduce_multi(ctx, C.cast(C.pointer(_lib.DepthReduceItem()), C.POINTER(_lib.DepthReduceItem)), 1, None),
            lambda: lib.orbit_exchange_list(ctx, None, 0, 1, None, None, 0, 8, 12, None),
            lambda: lib.orbit_exchange_list(ctx, some, 3, 2, one_ptr, one_ptr, 0, 8, 12, None),
            lambda: lib.orbit_exchange_list(ctx, some, 0, 1, one_ptr, one_ptr, 0, 6, 12, None),
            lambda: lib.orbit_exchange_list(ctx, some, 0, 17, one_ptr, one_ptr, 0, 8, 12, None),
            lambda: lib.orbit_compact_segments(ctx, None, 1, 16, None, 16, 8, 12, None),
            lambda: lib.orbit_compact_segments(ctx, some, 65, 16, some, 16, 8, 12, None),
            lambda: lib.orbit_allgather_list(ctx, None, 0, 1, some, 16, some, some, 16, 8, 12, None),
            lambda: lib.orbit_p2p_alloc(ctx, 0, None, None),
            lambda: lib.orbit_p2p_open(ctx, None, None),
        ]
        for i, call in enumerate(calls):
            rc = call()
            assert rc in bad, (i, rc)
            assert lib.orbit_last_error(ctx)  # a message is always available
        assert lib.orbit_ctx_status(ctx, None, 1) == _lib.OK  # and nothing was latched on the device
    finally:
        assert lib.orbit_ctx_destroy(ctx) == _lib.OK
    assert lib.orbit_ctx_destroy(None) == _lib.OK  # destroying nothing is not an error


def test_two_contexts_overlap_on_two_streams(torch_mod, oracle):
    """The concurrency contract of include/orbit_abi.h: one context per stream, contexts independent.  Two contexts
    cull two different scenes, frame after frame, on two streams with nothing ordering them, from two host threads;
    every frame of both equals the oracle."""
    import threading

    from orbit_amd.engine import Engine

    torch = torch_mod
    cam = sc.default_camera(rot=(0.6, 0.8))
    ci = sc.make_cull_info(cam.view, cam.planes)
    jobs = []
    for seed, n in ((41, 20_000), (42, 9_000)):
        scene = sc.make_scene(seed, n, n_meshes=max(50, n // 20), meshlets_per_mesh=(4, 64), extent=(200.0, 30.0, 200.0))
        gs = GpuScene(torch, scene)
        cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
        eng = Engine(0, max_entities=n + 256, max_dispatches=cap_d, max_draws=cap_c)
        stream = torch.cuda.Stream()
        outs = [(torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda"),
                 torch.zeros(L.DRAW_HEADER + 28 * cap_c, dtype=torch.uint8, device="cuda")) for _ in range(6)]
        jobs.append((scene, gs, eng, stream, cap_d, cap_c, outs))
    torch.cuda.synchronize()

    def run(job):
        scene, gs, eng, stream, cap_d, cap_c, outs = job
        for disp, draw in outs:
            eng.entity_cull(ci, gs.draws, gs.mesh_infos, disp, gs.entities, scene.entity_draw_count, cap_d, stream=stream)
            eng.meshlet_cull(ci, disp, gs.meshlets, draw, gs.entities, gs.materials, cap_d, cap_c, stream=stream)

    threads = [threading.Thread(target=run, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    try:
        for scene, gs, eng, stream, cap_d, cap_c, outs in jobs:
            ref = run_oracle(oracle, scene, ci)
            for disp, draw in outs:
                assert_same((host(disp), host(draw), None, None), ref)
            eng.status()
    finally:
        for j in jobs:
            j[2].close()


# ----------------------------------------------------------------------------- several views in one call
def test_cull_views_early_pass_and_cascades(torch_mod, oracle):
    """orbit_cull_views: the culls one frame of the reference issues independently — the early forward pass
    (VisibilityRead, perspective) and the four shadow cascades (orthographic, up to 11 planes:
    shadow_renderer.rs:466-706 through the host mirror) — side by side in one call, plus a dispatch-only view (mesh
    shading).  Every view's dispatch records, draw commands and bits equal the oracle's for that view; a second call
    reuses the side streams and scratch."""
    torch = torch_mod
    from orbit_amd import passes
    from orbit_amd.engine import Engine

    scene = sc.make_scene(17, 2600, n_meshes=400, extent=(70.0, 15.0, 70.0))
    gs = GpuScene(torch, scene)
    cam = sc.default_camera()
    rng = np.random.default_rng(17)
    evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    cis = [sc.make_cull_info(cam.view, cam.planes, occlusion_pass=1)]
    for cascade in range(4):
        hci, _, _ = passes.shadow_cascade(direction=(-0.45, 0.2, 0.1, 0.86), camera_position=(0.0, 2.0, 0.0),
                                          camera_orientation=(0.0, 0.0, 0.0, 1.0), camera_fov=float(cam.fov),
                                          camera_near_clip=cam.z_near, camera_aspect_ratio=cam.aspect,
                                          cascade_index=cascade, max_shadow_distance=80.0)
        cis.append(passes.cull_info_to_gpu(hci))
    cis.append(sc.make_cull_info(cam.view, cam.planes[:3]))  # dispatch-only view
    eng = Engine(0, max_entities=8192, max_dispatches=40_000, max_draws=200_000, max_views=4)  # 4 up front, 2 lazily
    disp_cap, draw_cap = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    views, bufs = [], []
    for k, ci in enumerate(cis):
        disp = torch.full((L.DISPATCH_HEADER + 16 * disp_cap,), 0xAB, dtype=torch.uint8, device="cuda")
        draw = torch.full((L.DRAW_HEADER + 28 * draw_cap,), 0xCD, dtype=torch.uint8, device="cuda")
        e_d, m_d = (dev(torch, evis), dev(torch, mvis)) if k == 0 else (None, None)
        bufs.append((disp, draw, e_d, m_d))
        views.append(dict(cull_info=ci, entity_draw_buffer=gs.draws, mesh_info_buffer=gs.mesh_infos,
                          meshlet_dispatch_buffer=disp, entity_buffer=gs.entities,
                          entity_draw_count=scene.entity_draw_count, dispatch_capacity=disp_cap,
                          meshlet_buffer=gs.meshlets, draw_commands_buffer=draw, material_buffer=gs.materials,
                          draw_capacity=draw_cap, visibility_buffer=e_d, meshlet_visibility_buffer=m_d,
                          skip_meshlet_stage=(k == len(cis) - 1)))
    for rep in range(2):
        if rep == 1:  # second call: the four pass-0 cascades cull from the derived meshlet streams, pass 1 does not
            eng.bind_meshlet_stream(eng.meshlet_stream(gs.meshlets, 0, len(scene.meshlets)))
        eng.cull_views(views)
        torch.cuda.synchronize()
        eng.status()
        assert eng.meshlet_stream_culls() == (4 if rep == 1 else 0)
        drawn = []
        for k, (ci, (disp, draw, e_d, m_d)) in enumerate(zip(cis, bufs)):
            ref = run_oracle(oracle, scene, ci, evis if k == 0 else None, mvis if k == 0 else None)
            if k == len(cis) - 1:  # dispatch records only; the draw buffer was never touched
                nrec = int(ref[0][:4].view(np.uint32)[0])
                assert np.array_equal(host(disp)[:12 + 16 * nrec], ref[0][:12 + 16 * nrec])
                assert bool((draw == 0xCD).all())
                continue
            _, cmds = assert_same((host(disp), host(draw), None if e_d is None else host(e_d, np.uint32),
                                   None if m_d is None else host(m_d, np.uint32)), ref)
            drawn.append(len(cmds))
        assert drawn[0] > 0 and sum(drawn[1:]) > 0, drawn
    from orbit_amd._lib import E_CAPACITY, OrbitError

    with pytest.raises(OrbitError) as ei:
        eng.cull_views(views + views[:3])
    assert ei.value.code == E_CAPACITY
    eng.close()


def test_exchange_list_capacity_and_header(torch_mod):
    """orbit_exchange_list with a world of one (its own control block, epochs counted across calls): the list lands at
    offset 0 behind a {total, 0} header; an output that cannot hold the concatenation latches ORBIT_E_CAPACITY, the header
    is clamped and nothing is written past it."""
    torch = torch_mod
    from orbit_amd._lib import E_CAPACITY, OrbitError
    from orbit_amd.dist import device_view
    from orbit_amd.engine import Engine

    eng = Engine(0)
    n = 1000
    local = torch.zeros(L.VISIBLE_HEADER + 12 * n, dtype=torch.uint8, device="cuda")
    local[:8].view(torch.int32)[:] = torch.tensor([n, 77], dtype=torch.int32)
    local[8:].view(torch.int32)[:] = torch.arange(3 * n, dtype=torch.int32, device="cuda")
    out_bytes = L.VISIBLE_HEADER + 12 * n + 64
    out_ptr, _ = eng.p2p_alloc(out_bytes)
    ctrl_ptr, _ = eng.p2p_alloc(1024)
    out = device_view(out_ptr, out_bytes, torch.device("cuda", 0))
    for cap in (n, n, n - 100):  # three epochs on one control block; the last one cut by the capacity
        out.fill_(0x5A)
        eng.exchange_list(local, 0, 1, [out_ptr], [ctrl_ptr], cap, L.VISIBLE_HEADER, 12)
        torch.cuda.synchronize()
        assert [int(v) for v in out[:8].view(torch.int32)] == [cap, 0]
        assert torch.equal(out[8:8 + 12 * cap], local[8:8 + 12 * cap]) and bool((out[8 + 12 * cap:] == 0x5A).all())
        if cap == n:
            eng.status()
    with pytest.raises(OrbitError) as ei:
        eng.status()
    assert ei.value.code == E_CAPACITY
    eng.p2p_free(ctrl_ptr)
    eng.p2p_free(out_ptr)
    eng.close()


def test_views_and_batched_pyramids_replay_as_a_graph(torch_mod, oracle):
    """orbit_cull_views and orbit_depth_reduce_multi only enqueue (the parameter blocks travel in the kernel-argument
    segments, no staging copy, no host wait): captured once, replayed on changed inputs — every replay equals the
    oracle for the inputs it found."""
    torch = torch_mod
    from orbit_amd.engine import Engine

    scene = sc.make_scene(23, 900, n_meshes=150, extent=(50.0, 10.0, 50.0))
    gs = GpuScene(torch, scene)
    cams = [sc.default_camera(), sc.default_camera(rot=(0.6, 0.8)), sc.default_camera(position=(5.0, 3.0, 9.0))]
    rng = np.random.default_rng(23)
    n = scene.entity_draw_count
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    W, H = 200, 120
    d = oracle.depth_pyramid_desc(W, H)
    ps = (d.width, d.height)
    eng = Engine(0, max_entities=4096, max_dispatches=20_000, max_draws=100_000, max_views=3)
    depths = [torch.zeros(H, W, dtype=torch.float32, device="cuda") for _ in cams]
    pyrs = [torch.zeros(d.total_texels, dtype=torch.float32, device="cuda") for _ in cams]
    evis = [torch.zeros((n + 31) // 32, dtype=torch.int32, device="cuda") for _ in cams]
    mvis = [torch.zeros(scene.vis_words, dtype=torch.int32, device="cuda") for _ in cams]
    cis, views = [], []
    for k, cam in enumerate(cams):
        ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
        cis.append(ci)
        disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
        draw = torch.zeros(L.DRAW_HEADER + 28 * cap_c, dtype=torch.uint8, device="cuda")
        views.append(dict(cull_info=ci, entity_draw_buffer=gs.draws, mesh_info_buffer=gs.mesh_infos,
                          meshlet_dispatch_buffer=disp, entity_buffer=gs.entities, entity_draw_count=n,
                          dispatch_capacity=cap_d, meshlet_buffer=gs.meshlets, draw_commands_buffer=draw,
                          material_buffer=gs.materials, draw_capacity=cap_c, visibility_buffer=evis[k],
                          meshlet_visibility_buffer=mvis[k], depth_pyramid=pyrs[k], depth_pyramid_size=ps))
    arr, keep = eng.prepare_views(views)
    items = [dict(depth=depths[k], width=W, height=H, pyramid=pyrs[k]) for k in range(len(cams))]

    def frame():
        eng.depth_reduce_multi(items)
        eng.cull_views_prepared(arr)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        frame()  # code objects loaded, view scratch in place
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        frame()
    for rep in range(3):  # new depth buffers and new bitsets every replay
        h_depth, h_e, h_m = [], [], []
        for k, cam in enumerate(cams):
            dd = sc.make_depth(100 * rep + k, W, H, cam, n_occluders=6 + 4 * rep)
            he = rng.integers(0, 2 ** 32, (n + 31) // 32, dtype=np.uint32)
            hm = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
            depths[k].copy_(torch.from_numpy(dd))
            evis[k].copy_(torch.from_numpy(he.view(np.int32)))
            mvis[k].copy_(torch.from_numpy(hm.view(np.int32)))
            h_depth.append(dd), h_e.append(he), h_m.append(hm)
        g.replay()
        torch.cuda.synchronize()
        eng.status()
        for k in range(len(cams)):
            opyr, _ = oracle.depth_reduce(h_depth[k], W, H)
            assert np.array_equal(host(pyrs[k], np.uint32), opyr.view(np.uint32)), (rep, k)
            ref = run_oracle(oracle, scene, cis[k], h_e[k], h_m[k], opyr, ps)
            assert_same((host(views[k]["meshlet_dispatch_buffer"]), host(views[k]["draw_commands_buffer"]),
                         host(evis[k], np.uint32), host(mvis[k], np.uint32)), ref)
    eng.close()


def test_cull_views_late_pass_views_with_their_own_pyramids(torch_mod, oracle):
    """The late half of a frame as ONE chain: orbit_depth_reduce_multi builds the views' pyramids (one launch pair),
    orbit_cull_views runs their pass-2 culls (VisibilityWrite: HiZ test, bits written) side by side — the main view
    (perspective) and two orthographic views, each against its own pyramid and its own bitsets, from the Meshlet buffer
    and from the derived streams.  Every view equals the oracle's single-view result."""
    torch = torch_mod
    from orbit_amd.engine import Engine, depth_pyramid_desc

    scene = sc.make_scene(19, 1800, n_meshes=300, extent=(50.0, 10.0, 50.0))
    gs = GpuScene(torch, scene)
    cam = sc.default_camera()
    rng = np.random.default_rng(19)
    screens = [(640, 360), (256, 256), (200, 120)]
    depths = [(rng.random((h, w)) * 0.02).astype(np.float32) for (w, h) in screens]
    for dpt in depths:  # far everywhere (reverse-Z: small), a near wall over part of the screen
        dpt[:, : dpt.shape[1] // 3] = 0.35
    descs = [depth_pyramid_desc(w, h) for (w, h) in screens]
    ortho = dict(projection_type=1, p00=1.0 / 40, p11=1.0 / 40, z_near=0.5, z_far=120.0)
    cis = [sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, p00=cam.p00, p11=cam.p11, z_near=cam.z_near),
           sc.make_cull_info(cam.view, cam.planes[:4], occlusion_pass=2, **ortho),
           sc.make_cull_info(sc.translation(3.0, -1.0, -20.0), cam.planes[:2], occlusion_pass=2, **ortho)]
    evis = [rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32) for _ in cis]
    mvis = [rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32) for _ in cis]
    disp_cap, draw_cap = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    # the launch chain from the Meshlet buffer and from the derived streams (cull_path = 1), then the library's own
    # choice for views of this size: the one-launch cull (one launch per projection type among the three views)
    for source in ("meshlet_buffer", "meshlet_stream", "one_launch"):
        if source != "meshlet_stream":
            eng = Engine(0, max_entities=8192, max_dispatches=40_000, max_draws=200_000, max_views=3,
                         cull_path=0 if source == "one_launch" else 1)
        else:
            eng.bind_meshlet_stream(eng.meshlet_stream(gs.meshlets, 0, len(scene.meshlets)))
        pyrs = [torch.zeros(d.total_texels, dtype=torch.float32, device="cuda") for d in descs]
        gdepths = [dev(torch, d).view(torch.float32) for d in depths]
        views, bufs = [], []
        for k, ci in enumerate(cis):
            disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device="cuda")
            draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device="cuda")
            e_d, m_d = dev(torch, evis[k]), dev(torch, mvis[k])
            bufs.append((disp, draw, e_d, m_d))
            views.append(dict(cull_info=ci, entity_draw_buffer=gs.draws, mesh_info_buffer=gs.mesh_infos,
                              meshlet_dispatch_buffer=disp, entity_buffer=gs.entities,
                              entity_draw_count=scene.entity_draw_count, dispatch_capacity=disp_cap,
                              meshlet_buffer=gs.meshlets, draw_commands_buffer=draw, material_buffer=gs.materials,
                              draw_capacity=draw_cap, visibility_buffer=e_d, meshlet_visibility_buffer=m_d,
                              depth_pyramid=pyrs[k], depth_pyramid_size=(descs[k].width, descs[k].height)))
        # the chain: 2 launches for the three pyramids, then the views' culls
        eng.depth_reduce_multi([dict(depth=gdepths[k], width=screens[k][0], height=screens[k][1], pyramid=pyrs[k])
                                for k in range(3)])
        eng.cull_views(views)
        torch.cuda.synchronize()
        eng.status()
        drawn = []
        for k, (ci, (disp, draw, e_d, m_d)) in enumerate(zip(cis, bufs)):
            opyr, _ = oracle.depth_reduce(depths[k], *screens[k])
            assert np.array_equal(host(pyrs[k], np.uint32), opyr.view(np.uint32).reshape(-1))
            ref = run_oracle(oracle, scene, ci, evis[k], mvis[k], opyr, (descs[k].width, descs[k].height))
            _, cmds = assert_same((host(disp), host(draw), host(e_d, np.uint32), host(m_d, np.uint32)), ref)
            drawn.append(len(cmds))
        assert all(n > 0 for n in drawn), drawn
        assert eng.meshlet_stream_culls() == (3 if source == "meshlet_stream" else 0)
        assert eng.fused_culls() == (3 if source == "one_launch" else 0)
        if source != "meshlet_buffer":
            eng.close()


def test_non_affine_matrices_and_non_finite_centres_take_the_literal_path(torch_mod, engine, oracle):
    """The evaluation shortens its arithmetic where the outcome cannot change: rows whose records have affine
    view x model matrices skip w and the division, and the symmetric five-plane frustum of a perspective camera is
    tested with shared products — for FINITE centres only (a product of 0 with an infinity is a NaN in the reference's
    arithmetic).  Here the inputs that must NOT take the short cuts are mixed into one scene: projective model matrices
    (last row not 0 0 0 1), infinite / NaN translations, infinite / NaN meshlet centres and radii — and plane sets that
    only look like the frustum (a non-zero y in the left plane, six planes, a shifted right plane)."""
    torch = torch_mod
    scene = sc.make_scene(83, 900, n_meshes=150, meshlets_per_mesh=(1, 70), extent=(40.0, 10.0, 40.0))
    rng = np.random.default_rng(83)
    ents = scene.entities["model_matrix"]
    n = len(ents)
    k = rng.permutation(n)
    ents[k[:60], 3] = 0.001     # column 0, row 3: w picks up x
    ents[k[60:120], 15] = 2.0   # w scale
    ents[k[120:150], 7] = -0.002
    ents[k[150:170], 12] = np.inf   # translation x
    ents[k[170:190], 13] = np.nan
    ents[k[190:200], 14] = -np.inf
    m = scene.meshlets
    j = rng.permutation(len(m))
    m["bounding_sphere"][j[:200], 0] = np.inf
    m["bounding_sphere"][j[200:400], 1] = np.nan
    m["bounding_sphere"][j[400:500], 2] = -np.inf
    m["bounding_sphere"][j[500:600], 3] = np.inf
    m["bounding_sphere"][j[600:700], 3] = np.nan
    cam = sc.default_camera(rot=(0.1, 0.3))
    gs = GpuScene(torch, scene)
    std = np.asarray(cam.planes[:5], np.float32).copy()
    variants = [std]
    v = std.copy(); v[0, 1] = 1e-3; variants.append(v)                  # the left plane tilts
    v = std.copy(); v[1, 0] = np.nextafter(v[1, 0], np.float32(0)); variants.append(v)  # right is no mirror of left
    v = std.copy(); v[4, 0] = 0.25; variants.append(v)                   # the near plane is no z-plane
    v = std.copy(); v[2, 3] = 0.5; variants.append(v)                    # an offset on the bottom plane
    variants.append(np.asarray(cam.planes[:6], np.float32))              # six planes
    variants.append(std[[1, 0, 2, 3, 4]])                                # right before left: still the frustum
    for planes in variants:
        for occlusion_pass in (0, 1):
            ci = sc.make_cull_info(cam.view, planes, occlusion_pass=occlusion_pass)
            evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32) if occlusion_pass else None
            mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32) if occlusion_pass else None
            assert_same(run_gpu(torch, engine, gs, ci, evis, mvis, material_count=len(scene.materials)),
                        run_oracle(oracle, scene, ci, evis, mvis))
