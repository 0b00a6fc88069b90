"""glTF ingestion and the LOD chain (orbit_amd/host/orbit_gltf.*: load_gltf, GpuAssets::add_mesh — SURVEY.md §8f-3)
on a procedurally generated asset (tools/make_test_glb.py).  The third-party parts of the reference's pipeline
(crates gltf, meshopt) are replaced by own code, so nothing here compares with the renderer's bytes ("parity
unpinned"): the results are checked against the document and the geometry themselves, and the oracle runs on them."""
import importlib.util
import json
import os
import struct

import numpy as np
import pytest

import scenes as sc
from orbit_amd import assets, gltf
from orbit_amd import layouts as L
from orbit_amd.passes import Panic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def glb_tool():
    spec = importlib.util.spec_from_file_location("make_test_glb", os.path.join(ROOT, "tools", "make_test_glb.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def loaded(tmp_path_factory):
    d = tmp_path_factory.mktemp("glb")
    path = str(d / "scene.glb")
    glb_tool().write(path)
    return path, gltf.load(path)


def test_document_structure(loaded):
    path, d = loaded
    doc, _ = glb_tool().build()
    assert len(d["mesh_infos"]) == len(doc["meshes"]) == 5
    assert list(d["materials"]["alpha_mode"]) == [0, 1, 2]  # OPAQUE, MASK, BLEND (assets/mod.rs:142-150)
    assert np.isclose(d["materials"]["alpha_cutoff"][1], 0.5) and d["materials"]["alpha_cutoff"][0] == 0.0
    assert np.allclose(d["materials"]["base_color"][2], [0.2, 0.4, 1.0, 0.5])
    assert len(d["entities"]) == len(doc["nodes"])  # every node becomes an entity, mesh or not (gltf_loader.rs:656-662)
    assert (d["entities"]["mesh"] < 0).sum() == 2   # "group" and "empty"
    # depth-first order, parents before children: group, arm, leaf, empty, terrain, inst0...
    names = ["group", "arm", "leaf", "empty", "terrain"]
    by_name = {n["name"]: i for i, n in enumerate(doc["nodes"])}
    want_mesh = [doc["nodes"][by_name[n]].get("mesh", -1) for n in names]
    assert list(d["entities"]["mesh"][:5]) == want_mesh
    assert d["vertex_positions"].shape[0] == sum(a["count"] for a in doc["accessors"] if a["type"] == "VEC3")


def test_lod_chain_structure(loaded):
    _, d = loaded
    mi, ml = d["mesh_infos"], d["meshlets"]
    cursor = 0
    for m in mi:
        n = int(m["lod_count"])
        assert 1 <= n <= L.MAX_MESH_LODS
        lods = m["mesh_lods"][:n]
        assert int(lods[0][0]) == cursor  # meshes and their levels are laid out back to back (assets/mod.rs:336-402)
        tris = []
        for off, cnt in lods:
            assert int(off) == cursor
            cursor += int(cnt)
            tris.append(int(ml["triangle_count"][off:off + cnt].sum()))
        assert np.all(m["mesh_lods"][n:] == 0)
        # the chain thins out: targets are x0.8 per submesh per level (:349,:391), results never exceed them
        assert all(b <= a for a, b in zip(tris, tris[1:])), tris
        if tris[0] > 1000:
            assert tris[-1] < 0.45 * tris[0], tris
    assert cursor == len(ml)
    # the two-primitive mesh: its scale decays twice per level (the reference's placement of `*= 0.8`)
    two = mi[2]
    t0 = int(ml["triangle_count"][two["mesh_lods"][0][0]:][:two["mesh_lods"][0][1]].sum())
    t1 = int(ml["triangle_count"][two["mesh_lods"][1][0]:][:two["mesh_lods"][1][1]].sum())
    assert t1 <= 0.8 * t0


def test_every_meshlet_of_every_lod_is_sound(loaded):
    _, d = loaded
    pos, data, ml, mi = d["vertex_positions"], d["meshlet_data"], d["meshlets"], d["mesh_infos"]
    vertex_end = list(mi["vertex_offset"][1:]) + [len(pos)]
    for m, vend in zip(mi, vertex_end):
        for off, cnt in m["mesh_lods"][:int(m["lod_count"])]:
            for k in range(int(off), int(off + cnt)):
                q = ml[k]
                nv, nt = int(q["vertex_count"]), int(q["triangle_count"])
                assert 3 <= nv <= assets.MAX_MESHLET_VERTICES and 1 <= nt <= assets.MAX_MESHLET_TRIANGLES
                tri = assets.meshlet_triangles(q, data)           # local indices through the u8 corners
                gv = tri.astype(np.int64) + int(q["vertex_offset"])  # mesh-relative -> global vertex
                assert gv.min() >= int(m["vertex_offset"]) and gv.max() < vend
                p = pos[gv.reshape(-1)]
                c, r = q["bounding_sphere"][:3], q["bounding_sphere"][3]
                assert np.all(np.linalg.norm(p - c, axis=1) <= r * 1.0001 + 1e-5)
                # no triangle with a repeated corner (the source sphere has zero-AREA triangles at its poles — distinct
                # indices on one point — which is the asset's business, not the chain's)
                assert np.all((gv[:, 0] != gv[:, 1]) & (gv[:, 1] != gv[:, 2]) & (gv[:, 0] != gv[:, 2]))
        # mesh bounds (mesh.rs:192-215): the sphere holds every vertex of the mesh
        mp = pos[int(m["vertex_offset"]):vend]
        assert np.all(np.linalg.norm(mp - m["bounding_sphere"][:3], axis=1) <= m["bounding_sphere"][3] * 1.00001)
        assert np.allclose(m["aabb_min"][:3], mp.min(0)) and np.allclose(m["aabb_max"][:3], mp.max(0))


def _trs(t, q, s):
    x, y, z, w = [float(v) for v in q]
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    m = np.eye(4)
    m[:3, :3] = r * np.asarray(s, dtype=np.float64)[None, :]
    m[:3, 3] = t
    return m


def test_node_hierarchy_and_transform_decomposition(loaded):
    _, d = loaded
    doc, _ = glb_tool().build()
    world = {}

    def walk(i, parent):
        n = doc["nodes"][i]
        if "matrix" in n:
            local = np.array(n["matrix"], dtype=np.float64).reshape(4, 4).T
        else:
            local = _trs(n.get("translation", [0, 0, 0]), n.get("rotation", [0, 0, 0, 1]), n.get("scale", [1, 1, 1]))
        m = parent @ local
        order.append((i, m))
        for c in n.get("children", []):
            walk(c, m)

    order = []
    for r in doc["scenes"][0]["nodes"]:
        walk(r, np.eye(4))
    assert len(order) == len(d["entities"])
    for (i, m), e in zip(order, d["entities"]):
        back = _trs(e["position"], e["orientation"], e["scale"])  # Transform::compute_matrix (scene.rs:54-56)
        assert np.allclose(back, m, atol=2e-4 * max(1.0, np.abs(m).max())), doc["nodes"][i].get("name")
        assert abs(np.linalg.norm(e["orientation"]) - 1.0) < 1e-5
    # the mirrored leaf keeps its handedness: one negative scale (glam puts the sign on x)
    leaf = d["entities"][2]
    assert leaf["scale"][0] < 0 < leaf["scale"][1]
    # decomposition of a plain TRS matrix gives the parts back
    p, q, s = gltf.transform_from_mat4(_trs([1, 2, 3], [0, 0.6, 0, 0.8], [2, 3, 4]).T.reshape(-1))
    assert np.allclose(p, [1, 2, 3]) and np.allclose(s, [2, 3, 4]) and np.allclose(q, [0, 0.6, 0, 0.8], atol=1e-6)


def test_gltf_with_external_buffer_and_malformed_input(loaded, tmp_path):
    path, d = loaded
    split = str(tmp_path / "split.gltf")
    glb_tool().write(split, split=True)
    d2 = gltf.load(split)
    for k in d:
        assert np.array_equal(d[k], d2[k]), k
    raw = open(path, "rb").read()
    (tmp_path / "cut.glb").write_bytes(raw[:len(raw) // 2])
    with pytest.raises(Panic):
        gltf.load(str(tmp_path / "cut.glb"))
    with pytest.raises(Panic):
        gltf.load(str(tmp_path / "missing.glb"))
    # a primitive that is not a triangle list: assert_eq!(primitive.mode(), Triangles) (gltf_loader.rs:441)
    doc, blob = glb_tool().build()
    doc["meshes"][0]["primitives"][0]["mode"] = 1
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    with open(tmp_path / "lines.glb", "wb") as f:
        f.write(b"glTF" + struct.pack("<II", 2, 12 + 8 + len(js) + 8 + len(blob)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(blob), 0x004E4942) + blob)
    with pytest.raises(Panic, match="TRIANGLES"):
        gltf.load(str(tmp_path / "lines.glb"))


def test_simplify_clustered_meets_its_target():
    tool = glb_tool()
    pos, idx = tool.torus()
    for frac in (1.5, 0.8, 0.33, 0.05, 0.001):
        target = int(len(idx) * frac)
        out = gltf.simplify_clustered(pos, idx, target)
        assert len(out) % 3 == 0 and len(out) <= max(target, 0) or frac >= 1.0
        if frac >= 1.0:
            assert np.array_equal(out, idx)  # nothing to do: the input comes back (like meshopt_simplify)
            continue
        assert set(out.tolist()) <= set(idx.tolist())  # existing vertices only
        t = out.reshape(-1, 3)
        assert np.all((t[:, 0] != t[:, 1]) & (t[:, 1] != t[:, 2]) & (t[:, 0] != t[:, 2]))
        if frac >= 0.05:
            assert len(out) >= 0.4 * target  # the finest grid that fits: not a collapse to nothing
            # the simplified surface stays near the original one: every kept vertex IS an original vertex, and the
            # triangles' centroids are within a cell of the torus surface (R = 2, r = 0.6)
            c = pos[t].mean(axis=1)
            dist = np.abs(np.hypot(np.hypot(c[:, 0], c[:, 2]) - 2.0, c[:, 1]) - 0.6)
            assert dist.max() < 0.6


def test_loaded_scene_through_the_oracle_picks_several_lods(loaded, oracle):
    _, d = loaded
    b = gltf.to_scene_buffers(d)
    n = int(b["entity_draws"][:4].view(np.uint32)[0])
    assert n == int((d["entities"]["mesh"] >= 0).sum())
    cam = sc.default_camera(position=(0.0, 1.0, 6.0))
    ci = sc.make_cull_info(cam.view, cam.planes, alpha_mode_flag=L.ALPHA_ALL)
    cap_d, cap_c = 60_000, len(d["meshlets"]) * 8
    disp, _, dropped = oracle.entity_cull(ci, b["entity_draws"], n, b["mesh_infos"], b["entities"], cap_d)
    assert dropped == 0
    _, recs = L.dispatch_buffer_records(disp)
    # which LOD each record's meshlet range belongs to
    lod_of = {}
    for m in d["mesh_infos"]:
        for k, (off, cnt) in enumerate(m["mesh_lods"][:int(m["lod_count"])]):
            for x in range(int(off), int(off + cnt)):
                lod_of[x] = k
    lods = {lod_of[int(r["meshlet_offset"])] for r in recs}
    assert len(lods) >= 4, lods  # instances from 4 m to 420 m: entity_cull.comp:204-209 selects across the chain
    draw, _, dropped = oracle.meshlet_cull(ci, disp, b["meshlets"], cap_c, b["entities"], b["materials"])
    assert dropped == 0 and int(draw[:4].view(np.uint32)[0]) > 100


def test_loader_survives_damaged_files(loaded, tmp_path):
    """Truncations and byte flips of the container, the JSON and the binary chunk: the loader either loads or reports
    an error (the reference unwraps and panics), it never reads outside its buffers (run under ASan in
    tools/README_sanitizers.txt)."""
    path, _ = loaded
    raw = bytearray(open(path, "rb").read())
    json_len = struct.unpack_from("<I", raw, 12)[0]
    rng = np.random.default_rng(99)
    outcomes = {"ok": 0, "error": 0}
    for k in range(120):
        b = bytearray(raw)
        mode = k % 4
        if mode == 0:    # truncate anywhere
            b = b[:int(rng.integers(0, len(b)))]
        elif mode == 1:  # flip bytes inside the JSON chunk (numbers, brackets, names)
            for _ in range(int(rng.integers(1, 6))):
                b[20 + int(rng.integers(0, json_len))] = int(rng.integers(32, 127))
        elif mode == 2:  # flip bytes of the headers / chunk table
            b[int(rng.integers(0, 28))] = int(rng.integers(0, 256))
        else:            # index data: out-of-range indices must be caught
            off = 20 + json_len + 8 + int(rng.integers(0, len(b) - (20 + json_len + 8)))
            b[off:off + 4] = struct.pack("<I", 0xFFFFFFF0)
        f = tmp_path / "damaged.glb"
        f.write_bytes(bytes(b))
        try:
            gltf.load(str(f))
            outcomes["ok"] += 1
        except Panic:
            outcomes["error"] += 1
    assert outcomes["error"] > 30 and outcomes["ok"] + outcomes["error"] == 120, outcomes


def _write_glb(path, doc, blob):
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    with open(path, "wb") as f:
        f.write(b"glTF" + struct.pack("<II", 2, 12 + 8 + len(js) + 8 + len(blob)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(blob), 0x004E4942) + blob)


@pytest.mark.parametrize("where,field,value", [
    ("bufferViews", "byteStride", 2 ** 63),       # (count - 1) * stride wraps: the round-2 crash (ADVICE)
    ("bufferViews", "byteStride", 2 ** 62),
    ("bufferViews", "byteStride", 256),           # above glTF's 252
    ("bufferViews", "byteStride", 6),             # not a multiple of 4
    ("bufferViews", "byteStride", -4),
    ("bufferViews", "byteOffset", -4),            # offset + ... wraps back into range
    ("bufferViews", "byteOffset", 2 ** 63 - 8),
    ("bufferViews", "byteLength", 8),             # a view too short for its accessor
    ("bufferViews", "byteLength", 2 ** 40),       # a view longer than the buffer
    ("accessors", "byteOffset", -16),
    ("accessors", "byteOffset", 2 ** 63 - 1),
    ("accessors", "count", -1),
    ("accessors", "count", 2 ** 62),
    ("accessors", "count", 2 ** 63 - 1),
])
def test_hostile_accessor_numbers_are_errors_not_wild_reads(tmp_path, where, field, value):
    """Offsets, strides and counts of an accessor come from the file, signed and unchecked: each of these made the
    extent test wrap (or would read through a wild pointer) before the checks in orbit_gltf.cpp Document::view."""
    doc, blob = glb_tool().build()
    hit = 0
    for k, item in enumerate(doc[where]):
        # the POSITION accessor of the first primitive and its view; and an index accessor / view
        d = json.loads(json.dumps(doc))
        d[where][k][field] = value
        p = tmp_path / f"hostile_{where}_{field}_{k}.glb"
        _write_glb(p, d, blob)
        try:
            gltf.load(str(p))
        except Panic:
            hit += 1
    assert hit >= 1, "no variant was rejected"
