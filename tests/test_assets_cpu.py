"""Asset side of the host mirror (orbit_amd/host/orbit_assets.hpp): triangle mesh -> Meshlet[] + meshlet data as
assets::mesh::compute_meshlets lays them out (src/assets/mesh.rs:292-338), with meshoptimizer's cluster bounds
restated.  Nothing in the reference pins these numbers ("parity unpinned"), so the checks are against the geometry:
the partition is exact, the sphere holds its vertices, and — the point of the normal cone — a meshlet that the cull
path's cone test rejects (oracle = the restated shader) has no triangle facing the camera."""
import numpy as np
import pytest

import scenes as sc
from orbit_amd import assets, layouts as L


uv_sphere, torus, MESHES, _outward = sc.uv_sphere, sc.torus, sc.MESHES, sc.outward_winding


@pytest.mark.parametrize("name", list(MESHES))
def test_meshlets_partition_the_mesh_within_the_limits(name):
    pos, idx = MESHES[name]()
    idx = _outward(pos, idx)
    meshlets, data = assets.compute_meshlets(pos, idx, material=5, vertex_offset=1000, data_offset_base=77)
    assert (meshlets["vertex_count"] <= assets.MAX_MESHLET_VERTICES).all()
    assert (meshlets["triangle_count"] <= assets.MAX_MESHLET_TRIANGLES).all() and (meshlets["triangle_count"] > 0).all()
    assert (meshlets["material_index"] == 5).all() and (meshlets["vertex_offset"] == 1000).all()
    # data layout: vertex indices, then u8 corners padded to a whole word; offsets continue from the base (mesh.rs:311-317)
    at = 77
    tris = []
    shifted = np.concatenate([np.zeros(77, np.uint32), data])
    for m in meshlets:
        assert int(m["data_offset"]) == at
        at += int(m["vertex_count"]) + (3 * int(m["triangle_count"]) + 3) // 4
        tris.append(assets.meshlet_triangles(m, shifted))
    assert at == 77 + len(data)
    got = np.concatenate(tris)
    assert np.array_equal(got, idx.reshape(-1, 3))  # every triangle once, in mesh order, corners in order
    assert len(meshlets) >= len(idx) // 3 // 64


@pytest.mark.parametrize("name", list(MESHES))
def test_sphere_contains_the_meshlet_and_the_mesh_bounds_contain_the_mesh(name):
    pos, idx = MESHES[name]()
    meshlets, data = assets.compute_meshlets(pos, _outward(pos, idx))
    for m in meshlets:
        v = pos[assets.meshlet_triangles(m, data).reshape(-1)]
        c, r = m["bounding_sphere"][:3], m["bounding_sphere"][3]
        assert np.all(np.linalg.norm(v - c, axis=1) <= r * (1 + 1e-5) + 1e-6)
    mn, mx, sp = assets.compute_mesh_bounds(pos)
    assert np.array_equal(mn, pos.min(axis=0)) and np.array_equal(mx, pos.max(axis=0))
    assert np.allclose(sp[:3], (mn + mx) * np.float32(0.5))
    assert np.all(np.linalg.norm(pos - sp[:3], axis=1) <= sp[3] * (1 + 1e-6))


def _one_entity_scene(pos, idx, model):
    meshlets, data = assets.compute_meshlets(pos, idx)
    mn, mx, sp = assets.compute_mesh_bounds(pos)
    mi = np.zeros(1, dtype=L.MESH_INFO)
    mi["bounding_sphere"][0] = sp
    mi["aabb_min"][0, :3], mi["aabb_max"][0, :3] = mn, mx
    mi["lod_count"] = 1
    mi["mesh_lods"][0, 0] = (0, len(meshlets))
    ents = np.zeros(1, dtype=L.ENTITY_DATA)
    ents["model_matrix"][0] = sc.mat4_cols(model)
    ents["normal_matrix"][0] = np.eye(4, dtype=np.float32).reshape(16)
    draws = np.zeros(1, dtype=L.ENTITY_DRAW)
    mats = np.zeros(1, dtype=L.MATERIAL)
    scene = sc.Scene(draws, ents, mi, meshlets, mats, (len(meshlets) + 31) // 32, len(meshlets))
    return scene, data


def _cone_trial(oracle, pos, idx, rng, scale):
    """One random rigid placement (+ uniform scale) and camera: -> (rejected meshlets, rejected ones that still have a
    triangle facing the camera)."""
    ang = rng.uniform(0, 2 * np.pi)
    c, s = np.cos(ang), np.sin(ang)
    model = np.eye(4, dtype=np.float32)
    model[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float32) * np.float32(scale)
    model[:3, 3] = rng.uniform(-5, 5, 3)
    scene, data = _one_entity_scene(pos, idx, model)
    cam_pos = rng.uniform(-12, 12, 3).astype(np.float32)
    cam = sc.default_camera(position=tuple(cam_pos), rot=(np.cos(ang * 0.7), np.sin(ang * 0.7)))
    ci = sc.make_cull_info(cam.view, np.zeros((0, 4), np.float32), alpha_mode_flag=L.ALPHA_ALL)  # cone test only
    disp, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), 1, scene.mesh_infos, scene.entities, 64)
    draw, _, _ = oracle.meshlet_cull(ci, disp, scene.meshlets, len(scene.meshlets) + 8, scene.entities, scene.materials)
    _, cmds = L.draw_buffer_commands(draw)
    drawn = set(int(v) for v in cmds["meshlet_index"])
    world = (pos @ model[:3, :3].T + model[:3, 3]).astype(np.float64)
    rejected = wrong = 0
    for mi_, m in enumerate(scene.meshlets):
        if mi_ in drawn:
            continue
        tri = world[assets.meshlet_triangles(m, data)]
        nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        facing = np.einsum("ij,ij->i", nrm, cam_pos.astype(np.float64) - tri[:, 0]) > 1e-9 * np.abs(nrm).max()
        rejected += 1
        wrong += bool(facing.any())
    return rejected, wrong


@pytest.mark.parametrize("name", list(MESHES))
def test_cone_rejected_meshlets_have_no_triangle_facing_the_camera(oracle, name):
    """Rigid motion and uniform scale <= 1: whatever the cone test (meshlet_cull.comp:104-106,148-158, restated by the
    oracle) rejects is entirely back-facing."""
    pos, idx = MESHES[name]()
    idx = _outward(pos, idx)
    rng = np.random.default_rng(3)
    total = 0
    for trial in range(12):
        rejected, wrong = _cone_trial(oracle, pos, idx, rng, scale=rng.choice([0.5, 1.0]))
        assert wrong == 0, (name, trial, "cone test rejected a meshlet with a front-facing triangle")
        total += rejected
    assert total > 30  # the test really rejected something (a closed surface hides about half of itself)


def test_reference_cone_test_over_culls_scaled_up_entities(oracle):
    """Reference behaviour, reproduced bit for bit by the engine and surfaced by the geometry: meshlet_cull.comp:121
    transforms the cone axis by the model matrix WITHOUT renormalising it, so for an entity scaled by k > 1 the left
    side of `dot(c, axis) >= cutoff * |c| + r` grows by k while the cutoff term does not — the test then also rejects
    meshlets that face the camera.  (meshoptimizer's formula assumes a unit axis.)  The drop-in keeps the behaviour;
    this test documents it."""
    pos, idx = uv_sphere()
    idx = _outward(pos, idx)
    rng = np.random.default_rng(3)
    wrong = sum(_cone_trial(oracle, pos, idx, rng, scale=3.0)[1] for _ in range(12))
    assert wrong > 0


@pytest.mark.parametrize("name", list(MESHES))
def test_nothing_potentially_visible_is_culled(oracle, name):
    """Frustum and cone together, from the geometry's side: a meshlet with a triangle that faces the camera and has a
    vertex inside the view frustum may show up on screen, so the cull path (planes of frustum_planes_from_matrix +
    cone test, scale <= 1) must draw it."""
    pos, idx = MESHES[name]()
    idx = _outward(pos, idx)
    rng = np.random.default_rng(5)
    must_total = 0
    for trial in range(10):
        ang = rng.uniform(0, 2 * np.pi)
        c, s, k = np.cos(ang), np.sin(ang), rng.choice([0.5, 1.0])
        model = np.eye(4, dtype=np.float32)
        model[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float32) * np.float32(k)
        model[:3, 3] = rng.uniform(-3, 3, 3)
        scene, data = _one_entity_scene(pos, idx, model)
        cam_pos = rng.uniform(-6, 6, 3).astype(np.float32)
        cam = sc.default_camera(position=tuple(cam_pos), rot=(np.cos(ang * 0.3), np.sin(ang * 0.3)))
        ci = sc.make_cull_info(cam.view, cam.planes, alpha_mode_flag=L.ALPHA_ALL)
        disp, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), 1, scene.mesh_infos, scene.entities, 64)
        draw, _, _ = oracle.meshlet_cull(ci, disp, scene.meshlets, len(scene.meshlets) + 8, scene.entities, scene.materials)
        _, cmds = L.draw_buffer_commands(draw)
        drawn = set(int(v) for v in cmds["meshlet_index"])
        world = (pos @ model[:3, :3].T + model[:3, 3]).astype(np.float64)
        view = (world @ cam.view[:3, :3].astype(np.float64).T) + cam.view[:3, 3].astype(np.float64)
        planes = np.asarray(cam.planes, dtype=np.float64)
        inside = np.all(view @ planes[:, :3].T + planes[:, 3] > 1e-6, axis=1)  # strictly inside every plane
        for mi_, m in enumerate(scene.meshlets):
            t = assets.meshlet_triangles(m, data)
            tri = world[t]
            nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
            facing = np.einsum("ij,ij->i", nrm, cam_pos.astype(np.float64) - tri[:, 0]) > 1e-9 * np.abs(nrm).max()
            if (facing & inside[t].any(axis=1)).any():
                must_total += 1
                assert mi_ in drawn, (name, trial, mi_, "a potentially visible meshlet was culled")
    assert must_total > 20


@pytest.mark.parametrize("seed,entities,spread,zfar", [(5, 80, 3.0, 12.0), (2, 150, 2.0, 30.0)])
def test_hiz_never_rejects_a_meshlet_that_owns_a_pixel(oracle, seed, entities, spread, zfar):
    """The whole occlusion chain against exact depth (tools/hiz_geometry_check.py): real meshes are rasterised in
    numpy, the depth goes through depth_reduce, every meshlet through the pass-2 test (entity stage + meshlet stage,
    frame-0 bitsets).  A meshlet that is the nearest surface at some pixel centre is visible by construction and must
    be drawn; the test must also have rejected a good part of the hidden ones, or it proves nothing."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import hiz_geometry_check as hz

    r = hz.check(seed, entities, 320, 180, spread, zfar)
    assert r["visible_but_rejected"] == 0, r
    assert r["visible"] >= 60 and r["drawn"] < 0.8 * r["instances"], r


def test_two_pass_culling_does_not_change_the_picture(oracle):
    """The two-pass protocol over a moving camera, judged by the image (tools/hiz_geometry_check.py::frame_sequence):
    pass 1 redraws last frame's visible set, its rasterised depth feeds depth_reduce, pass 2 adds what became visible
    and rewrites the bits.  Rasterising only what the two passes drew gives, pixel for pixel, the depth image of
    rasterising every meshlet — in every frame — while the passes skip a part of the scene."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import hiz_geometry_check as hz

    rows = hz.frame_sequence(seed=3, frames=4)
    assert all(r["wrong_pixels"] == 0 for r in rows), rows
    assert rows[0]["pass1"] == 0 and rows[0]["pass2"] > 0          # frame 0: everything is new
    assert any(r["pass2"] > 0 for r in rows[2:])                   # the camera motion disoccludes something later
    assert any(r["pass1"] + r["pass2"] < r["instances"] for r in rows[2:])  # and occlusion really skips work
