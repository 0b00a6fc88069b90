#!/usr/bin/env python3
"""Procedural glTF 2.0 test asset (no network, no third-party files): writes a .glb (or .gltf + .bin with --split)
that exercises what the reference's loader touches (src/gltf_loader.rs:423-676):

  meshes      torus (u32 indices, 15 k triangles), uv-sphere (u16), a two-primitive mesh (two materials -> two
              submeshes: the LOD chain's x0.8-per-submesh quirk), a terrain grid (interleaved vertex buffer with a
              byteStride, 32 k triangles), a quad (u8 indices)
  materials   OPAQUE, MASK (alphaCutoff 0.5), BLEND
  nodes       TRS and `matrix` forms, rotations, non-uniform and mirrored scales, a three-level hierarchy, instances
              of every mesh out to a few hundred metres (the cull's LOD pick sees several levels), a mesh-less node

usage: python tools/make_test_glb.py out.glb [--split] [--instances N]
"""
import argparse
import json
import os
import struct

import numpy as np


def torus(n_major=160, n_minor=48, R=2.0, r=0.6):
    u = np.linspace(0, 2 * np.pi, n_major, endpoint=False)
    v = np.linspace(0, 2 * np.pi, n_minor, endpoint=False)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    pos = np.stack([(R + r * np.cos(vv)) * np.cos(uu), r * np.sin(vv), (R + r * np.cos(vv)) * np.sin(uu)], -1)
    i, j = np.meshgrid(np.arange(n_major), np.arange(n_minor), indexing="ij")
    a = i * n_minor + j
    b = ((i + 1) % n_major) * n_minor + j
    c = ((i + 1) % n_major) * n_minor + (j + 1) % n_minor
    d = i * n_minor + (j + 1) % n_minor
    idx = np.stack([a, b, c, a, c, d], -1).reshape(-1)
    return pos.reshape(-1, 3).astype(np.float32), idx.astype(np.uint32)


def uv_sphere(n_lat=24, n_lon=48, radius=1.0, lat_range=(0.0, np.pi)):
    th = np.linspace(lat_range[0], lat_range[1], n_lat + 1)
    ph = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    tt, pp = np.meshgrid(th, ph, indexing="ij")
    pos = radius * np.stack([np.sin(tt) * np.cos(pp), np.cos(tt), np.sin(tt) * np.sin(pp)], -1)
    i, j = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
    a = i * n_lon + j
    b = (i + 1) * n_lon + j
    c = (i + 1) * n_lon + (j + 1) % n_lon
    d = i * n_lon + (j + 1) % n_lon
    idx = np.stack([a, c, b, a, d, c], -1).reshape(-1)
    return pos.reshape(-1, 3).astype(np.float32), idx.astype(np.uint32)


def grid(n=128, size=40.0, amp=1.5):
    x = np.linspace(-size / 2, size / 2, n + 1)
    xx, zz = np.meshgrid(x, x, indexing="ij")
    yy = amp * np.sin(xx * 0.35) * np.cos(zz * 0.27) + 0.3 * amp * np.sin(xx * 1.3 + zz * 0.9)
    pos = np.stack([xx, yy, zz], -1).reshape(-1, 3)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a = i * (n + 1) + j
    b = (i + 1) * (n + 1) + j
    c = (i + 1) * (n + 1) + j + 1
    d = i * (n + 1) + j + 1
    idx = np.stack([a, d, c, a, c, b], -1).reshape(-1)
    return pos.astype(np.float32), idx.astype(np.uint32)


def quat_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    s = np.sin(angle / 2)
    return [float(axis[0] * s), float(axis[1] * s), float(axis[2] * s), float(np.cos(angle / 2))]


class Builder:
    def __init__(self):
        self.bin = bytearray()
        self.views, self.accessors, self.meshes, self.nodes = [], [], [], []

    def _view(self, data: bytes, stride=None, target=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        v = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        if target:
            v["target"] = target
        self.bin += data
        self.views.append(v)
        return len(self.views) - 1

    def positions(self, pos, interleave_pad=0):
        pos = np.ascontiguousarray(pos, dtype=np.float32)
        if interleave_pad:  # xyz followed by `interleave_pad` floats of something else (a normal, say): byteStride
            buf = np.zeros((len(pos), 3 + interleave_pad), dtype=np.float32)
            buf[:, :3] = pos
            buf[:, 3:] = 0.5
            view = self._view(buf.tobytes(), stride=4 * (3 + interleave_pad), target=34962)
        else:
            view = self._view(pos.tobytes(), target=34962)
        self.accessors.append({"bufferView": view, "componentType": 5126, "count": len(pos), "type": "VEC3",
                               "min": [float(v) for v in pos.min(0)], "max": [float(v) for v in pos.max(0)]})
        return len(self.accessors) - 1

    def indices(self, idx, component):
        dt, ct = {1: (np.uint8, 5121), 2: (np.uint16, 5123), 4: (np.uint32, 5125)}[component]
        idx = np.ascontiguousarray(idx, dtype=dt)
        view = self._view(idx.tobytes(), target=34963)
        self.accessors.append({"bufferView": view, "componentType": ct, "count": len(idx), "type": "SCALAR"})
        return len(self.accessors) - 1

    def mesh(self, name, primitives):
        """primitives: list of (positions, indices, material, index_component_bytes, interleave_pad)"""
        prims = []
        for pos, idx, mat, comp, pad in primitives:
            prims.append({"attributes": {"POSITION": self.positions(pos, pad)}, "indices": self.indices(idx, comp),
                          "material": mat, "mode": 4})
        self.meshes.append({"name": name, "primitives": prims})
        return len(self.meshes) - 1

    def node(self, **kw):
        self.nodes.append(kw)
        return len(self.nodes) - 1


def build(instances=60, seed=7):
    b = Builder()
    t_pos, t_idx = torus()
    s_pos, s_idx = uv_sphere()
    cap_pos, cap_idx = uv_sphere(12, 32, 1.2, (0.0, np.pi / 2))
    plate_pos, plate_idx = grid(24, 4.0, 0.0)
    g_pos, g_idx = grid()
    q_pos = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], dtype=np.float32)
    q_idx = np.array([0, 2, 1, 0, 3, 2], dtype=np.uint32)
    m_torus = b.mesh("torus", [(t_pos, t_idx, 0, 4, 0)])
    m_sphere = b.mesh("sphere", [(s_pos, s_idx, 1, 2, 0)])
    m_two = b.mesh("plate_and_cap", [(plate_pos, plate_idx, 0, 2, 0), (cap_pos + np.float32([0, 0.05, 0]), cap_idx, 2, 2, 0)])
    m_grid = b.mesh("terrain", [(g_pos, g_idx, 0, 4, 3)])
    m_quad = b.mesh("quad", [(q_pos, q_idx, 1, 1, 0)])
    meshes = [m_torus, m_sphere, m_two, m_grid, m_quad]
    rng = np.random.default_rng(seed)
    roots = []
    # a three-level hierarchy: group (TRS) -> arm (matrix form, rotated + non-uniform scale) -> leaf (mirrored scale)
    leaf = b.node(name="leaf", mesh=m_sphere, translation=[0.0, 1.5, 0.0], scale=[-0.5, 0.5, 0.5])
    c, s = np.cos(0.7), np.sin(0.7)
    arm_m = np.array([[1.5 * c, 0, -1.5 * s, 0], [0, 0.75, 0, 0], [s, 0, c, 0], [3.0, 0.5, 0.0, 1]], dtype=np.float32)  # columns
    arm = b.node(name="arm", mesh=m_two, matrix=[float(v) for v in arm_m.reshape(-1)], children=[leaf])
    empty = b.node(name="empty")  # an entity without a mesh (gltf_loader.rs:656-662 adds it all the same)
    group = b.node(name="group", translation=[0.0, 0.0, -12.0], rotation=quat_axis_angle((0, 1, 0), 0.4),
                   children=[arm, empty])
    roots.append(group)
    roots.append(b.node(name="terrain", mesh=m_grid, translation=[0.0, -3.0, -30.0]))
    for i in range(instances):
        m = meshes[int(rng.integers(0, len(meshes)))]
        dist = float(rng.uniform(4.0, 420.0))
        ang = float(rng.uniform(-0.9, 0.9))
        pos = [dist * np.sin(ang), float(rng.uniform(-6, 6)), -dist * np.cos(ang)]
        sc = float(rng.uniform(0.6, 2.5))
        kw = dict(name=f"inst{i}", mesh=m, translation=[float(v) for v in pos],
                  rotation=quat_axis_angle(rng.normal(size=3), float(rng.uniform(0, 6.28))))
        kw["scale"] = [sc, sc, sc] if i % 3 else [sc, sc * 0.5, sc * 1.5]
        roots.append(b.node(**kw))
    doc = {
        "asset": {"version": "2.0", "generator": "tools/make_test_glb.py"},
        "scene": 0, "scenes": [{"nodes": roots}], "nodes": b.nodes, "meshes": b.meshes,
        "materials": [
            {"name": "opaque", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.7, 0.6, 1.0], "metallicFactor": 0.1}},
            {"name": "masked", "alphaMode": "MASK", "alphaCutoff": 0.5, "emissiveFactor": [0.1, 0.0, 0.0]},
            {"name": "blended", "alphaMode": "BLEND", "pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.4, 1.0, 0.5]}},
        ],
        "accessors": b.accessors, "bufferViews": b.views, "buffers": [{"byteLength": len(b.bin)}],
    }
    return doc, bytes(b.bin)


def write(path, split=False, instances=60, seed=7):
    doc, blob = build(instances, seed)
    if split:
        bin_name = os.path.splitext(os.path.basename(path))[0] + ".bin"
        doc["buffers"][0]["uri"] = bin_name
        with open(os.path.join(os.path.dirname(path) or ".", bin_name), "wb") as f:
            f.write(blob)
        with open(path, "w") as f:
            json.dump(doc, f)
        return
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * (-len(js) % 4)
    blob += b"\0" * (-len(blob) % 4)
    total = 12 + 8 + len(js) + 8 + len(blob)
    with open(path, "wb") as f:
        f.write(b"glTF" + struct.pack("<II", 2, total))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(blob), 0x004E4942) + blob)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--split", action="store_true", help="write .gltf + external .bin instead of a .glb")
    ap.add_argument("--instances", type=int, default=60)
    a = ap.parse_args()
    write(a.out, a.split, a.instances)
    print(a.out, os.path.getsize(a.out), "bytes")
