"""examples/native_frame: a plain C++/HIP process (no Python, no torch) that loads an `.orbitscene` dump, drives
entity_cull -> meshlet_cull -> depth_reduce through the C ABI and writes what it got.  Here the dump is made from a
synthetic scene, the binary is run as a child process, and its outputs must equal the oracle's on the same dump."""
import os
import struct
import subprocess

import numpy as np
import pytest

import scenes as sc
from orbit_amd import layouts as L
from orbit_amd import orbitscene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "examples", "native_frame")

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("stream", [False, True], ids=["meshlet_buffer", "meshlet_stream"])
def test_native_process_matches_the_oracle(tmp_path, oracle, stream):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(BIN):
        subprocess.run(["make", "-C", os.path.join(ROOT, "examples"), "-s"], check=True)
    scene = sc.make_scene(17, 1500, n_meshes=200, lods=2, meshlets_per_mesh=(1, 80))
    cam = sc.default_camera(rot=(0.6, 0.8))
    ci = sc.make_cull_info(cam.view, cam.planes, lod_base=8.0)
    W, H = 333, 187
    depth = sc.make_depth(5, W, H, cam)
    dump, out = str(tmp_path / "scene.orbitscene"), str(tmp_path / "out.bin")
    orbitscene.write(dump, entity_draws=scene.entity_draw_buffer(), mesh_infos=scene.mesh_infos,
                     entities=scene.entities, meshlets=scene.meshlets, materials=scene.materials,
                     cull_info=np.asarray(ci).reshape(1), depth=depth, screen=np.array([W, H], dtype=np.uint32))
    # 20 eager + 20 graph replays; --stream: the culls read the derived meshlet streams (also inside the captured graph)
    r = subprocess.run([BIN, dump, out, "20"] + (["--stream"] if stream else []), capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr
    assert "graph replay" in r.stdout, r.stdout  # the frame was captured and replayed; outputs below are the replay's
    served = int(r.stdout.split("served from the derived streams:")[1].split()[0])
    assert (served > 0) == stream, r.stdout
    raw = open(out, "rb").read()
    n_disp, n_draw, n_pyr = struct.unpack_from("<QQQ", raw, 0)
    gdisp = np.frombuffer(raw, np.uint8, n_disp, 24)
    gdraw = np.frombuffer(raw, np.uint8, n_draw, 24 + n_disp)
    gpyr = np.frombuffer(raw, np.uint32, n_pyr // 4, 24 + n_disp + n_draw)

    # capacities as the binary derives them (largest LOD of every draw)
    most = scene.mesh_infos["mesh_lods"][scene.entity_draws["mesh_index"], :, 1].max(axis=1).astype(np.int64)
    disp_cap, draw_cap = int(((most + 31) // 32).sum()) + 8, int(most.sum()) + 8
    assert n_disp == L.DISPATCH_HEADER + 16 * disp_cap and n_draw == L.DRAW_HEADER + 28 * draw_cap
    odisp, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                     scene.entities, disp_cap)
    odraw, _, _ = oracle.meshlet_cull(ci, odisp, scene.meshlets, draw_cap, scene.entities, scene.materials)
    oh, orecs = L.dispatch_buffer_records(odisp)
    gh, grecs = L.dispatch_buffer_records(gdisp)
    assert list(gh) == list(oh) and np.array_equal(grecs.view(np.uint32), orecs.view(np.uint32))
    on, ocmds = L.draw_buffer_commands(odraw)
    gn, gcmds = L.draw_buffer_commands(gdraw)
    assert gn == on and on > 0 and np.array_equal(gcmds.view(np.uint32), ocmds.view(np.uint32))
    opyr, _ = oracle.depth_reduce(depth, W, H)
    assert np.array_equal(gpyr, opyr.view(np.uint32).reshape(-1))
    assert f"draw commands {on}" in r.stdout
