"""`.orbitscene`: the cull path's input buffers, verbatim, in one file (SURVEY.md §8d "real-asset path").

The renderer's glTF -> meshlet pipeline (src/gltf_loader.rs, src/assets/mesh.rs, meshopt) is out of scope; a machine
that runs the Rust renderer can dump the buffers it uploads — they are plain `bytemuck` byte slices there — and this
engine consumes the dump unchanged.  Layout (little-endian):

    0   8   magic  b"ORBSCN01"
    8   4   u32    section count n
    12  4   u32    0
    16  n x { char name[24] (NUL padded); u64 offset; u64 size }   section table
    ...            section payloads, each 16-B aligned

Sections (std430 layouts of include/orbit_abi.h; the Rust side names the source buffer):
    entity_draws   EntityDrawBuffer: u32 count + EntityDraw[]            (scene.rs:127-133)
    mesh_infos     MeshInfo[] 128 B                                       (assets/mod.rs:18-28)
    entities       EntityData[] 128 B                                     (scene.rs:120-125)
    meshlets       Meshlet[] 32 B                                         (assets/mod.rs:111-122)
    materials      MaterialData[] 80 B                                    (assets/mod.rs:171-191)
    lights         LightData[] 64 B (optional)                            (scene.rs:278-291)
    cull_info      GpuCullInfo 400 B of the main view (optional)          (draw_gen.rs:208-237)
    depth          f32[h][w] reverse-Z depth of the frame (optional), with
    screen         u32 width, height
"""
import struct

import numpy as np

from . import layouts as L

MAGIC = b"ORBSCN01"
_DTYPES = {"mesh_infos": L.MESH_INFO, "entities": L.ENTITY_DATA, "meshlets": L.MESHLET, "materials": L.MATERIAL,
           "lights": L.LIGHT}


def write(path, **sections):
    """sections: name -> numpy array (any dtype; written as bytes)."""
    names = list(sections)
    table = 16 + 40 * len(names)
    off = (table + 15) // 16 * 16
    entries, blobs = [], []
    for n in names:
        b = np.ascontiguousarray(sections[n]).view(np.uint8).reshape(-1).tobytes()
        entries.append((n.encode()[:23], off, len(b)))
        blobs.append((off, b))
        off = (off + len(b) + 15) // 16 * 16
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<II", len(names), 0))
        for n, o, s in entries:
            f.write(n.ljust(24, b"\0") + struct.pack("<QQ", o, s))
        for o, b in blobs:
            f.seek(o)
            f.write(b)


def read(path):
    """-> dict name -> numpy array (typed where the layout is known, raw uint8 otherwise)."""
    raw = np.fromfile(path, dtype=np.uint8)
    if raw[:8].tobytes() != MAGIC:
        raise ValueError(f"{path}: not an .orbitscene file")
    n = struct.unpack_from("<I", raw, 8)[0]
    out = {}
    for i in range(n):
        ent = raw[16 + 40 * i:16 + 40 * (i + 1)].tobytes()
        name = ent[:24].rstrip(b"\0").decode()
        off, size = struct.unpack("<QQ", ent[24:])
        if off + size > len(raw):
            raise ValueError(f"{path}: section {name} runs past the end of the file")
        blob = raw[off:off + size]
        if name in _DTYPES:
            if size % _DTYPES[name].itemsize:
                raise ValueError(f"{path}: section {name} is not a whole number of {_DTYPES[name].itemsize}-B records")
            out[name] = blob.view(_DTYPES[name]).copy()
        elif name == "cull_info":
            out[name] = blob[:400].view(L.GPU_CULL_INFO)[0].copy()
        elif name == "depth":
            out[name] = blob.view(np.float32).copy()
        elif name == "screen":
            out[name] = blob.view(np.uint32).copy()
        else:
            out[name] = blob.copy()
    if "depth" in out and "screen" in out:
        w, h = (int(v) for v in out["screen"][:2])
        out["depth"] = out["depth"].reshape(h, w)
    return out
