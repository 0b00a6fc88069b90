"""Product against oracle on knife-edge scenes (the generator of tests/golden/make_spirv_vectors.py, which needs only
the oracle): quantised geometry, meshlet radii that put the cone test within an ulp of equality, and in pass 2 a constant
depth buffer bisected to the closest-depth of a group of identical spheres.  On such scenes a single operation fused or
rounded differently between kernels and oracle changes the lists — the 12 committed reference-binary cases are of this
kind; these are more seeds of it."""
import os
import sys

import numpy as np
import pytest

import scenes as sc
from orbit_amd import layouts as L
from test_gpu_parity import GpuScene, assert_same, run_gpu, run_oracle, torch_mod  # noqa: F401

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["meshlet_buffer", "meshlet_stream", "meshlet_stream_classes", "one_launch"])
def engine(torch_mod, request):
    from orbit_amd.engine import Engine
    from stream_engine import StreamEngine

    kw = dict(max_entities=8192, max_dispatches=60_000, max_draws=400_000)
    if request.param == "one_launch":  # entity + meshlet stage as ONE launch (cull_fused.hip)
        from fused_engine import FusedEngine

        e = FusedEngine(0, **kw)
    else:
        e = Engine(0, **kw) if request.param == "meshlet_buffer" else StreamEngine(
            0, classes=request.param == "meshlet_stream_classes", **kw)
    yield e
    e.close()


@pytest.mark.parametrize("ortho", [False, True], ids=["persp", "ortho"])
@pytest.mark.parametrize("occlusion_pass", [0, 1, 2])
@pytest.mark.parametrize("seed", range(31, 31 + int(os.environ.get("ORBIT_KNIFE_SEEDS", "4"))))
def test_knife_edge_scene(torch_mod, engine, oracle, seed, occlusion_pass, ortho):
    import make_spirv_vectors as gen

    rng = np.random.default_rng(seed)
    scene = sc.make_scene(seed, 300, meshlets_per_mesh=(1, 70), lods=3)
    gen.quantise(scene, rng)
    cam = sc.default_camera(position=(0.0, 0.0, float(rng.integers(4, 30))), rot=(1.0, 0.0))
    gen.cone_knife_edge(scene, cam, ortho)
    kw = {}
    if occlusion_pass == 2:
        kw = gen.ORTHO if ortho else dict(p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
    ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=occlusion_pass, projection_type=1 if ortho else 0, **kw)
    evis = mvis = pyr = None
    ps = (0, 0)
    if occlusion_pass:
        evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
        mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    if occlusion_pass == 2:
        try:
            depth = gen.depth_knife_edge(scene, cam, ci, ortho, seed, rng)
        except AssertionError:  # the group is not decided by the depth buffer in this scene: a seeded depth buffer
            depth = sc.make_depth(9 + seed, gen.W, gen.H, cam)
        evis[:], mvis[:] = 0, 0
        pyr, d = oracle.depth_reduce(depth, gen.W, gen.H)
        ps = (d.width, d.height)
    gs = GpuScene(torch_mod, scene)
    recs, cmds = assert_same(run_gpu(torch_mod, engine, gs, ci, evis, mvis, pyr, ps),
                             run_oracle(oracle, scene, ci, evis, mvis, pyr, ps))
    assert len(recs) > 50 and len(cmds) > 100
