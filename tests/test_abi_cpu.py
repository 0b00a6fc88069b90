"""CPU checks of the drop-in boundary: the shared library loads, exports every symbol
include/orbit_abi.h declares, its ctypes mirrors have the header's layouts, and — with no
GPU — context creation fails loudly instead of falling back to a host path."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from orbit_amd import _lib, layouts as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "orbit_abi.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(orbit_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/orbit_abi.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes signature in orbit_amd/_lib.py"
    assert lib.orbit_abi_version() == 4


def test_header_compiles_as_c_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "orbit_abi.h"\nint main(void){return sizeof(OrbitGpuCullInfo)==400?0:1;}\n')
    for cc, std in (("gcc", "-std=c11"), ("g++", "-std=c++17")):
        exe = tmp_path / f"t_{cc}"
        subprocess.run([cc, std, "-x", "c" if cc == "gcc" else "c++", "-I", os.path.join(ROOT, "include"),
                        str(src), "-o", str(exe)], check=True)
        assert subprocess.run([str(exe)]).returncode == 0


def test_ctypes_mirrors_match_header_sizes():
    # the sizes below are the ORBIT_STATIC_ASSERTs of the header
    assert C.sizeof(_lib.Caps) == 36
    assert C.sizeof(_lib.EntityCullBufs) == 6 * 8 + 16 + 8   # + depth_pyramid_levels (ABI version 2)
    assert C.sizeof(_lib.MeshletCullBufs) == 7 * 8 + 24 + 8
    assert C.sizeof(_lib.DepthPyramidLevel) == 16 and C.sizeof(_lib.DepthReduceItem) == 40
    assert C.sizeof(_lib.DepthPyramidDesc) == 16 + 3 * 16 * 4
    assert L.GPU_CULL_INFO.itemsize == 400 and L.MESHLET.itemsize == 32 and L.MESHLET_DRAW_COMMAND.itemsize == 28


def test_default_caps_are_the_reference_constants():
    lib = _lib.load()
    c = _lib.Caps()
    lib.orbit_default_caps(C.byref(c))
    # scene.rs:303-304, draw_gen.rs:15-16, device.rs:369-372
    assert (c.max_entities, c.max_dispatches, c.max_draws, c.max_lights, c.dispatch_size) == (
        100_000, 1_000_000, 1_000_000, 2_000, 32)


@pytest.mark.parametrize("size,expect", [((1920, 1080), (1024, 1024, 11)), ((2560, 1080), (2048, 1024, 12)),
                                         ((1, 1), (1, 1, 1))])
def test_pyramid_desc_matches_oracle(oracle, size, expect):
    from orbit_amd.engine import depth_pyramid_desc

    d, o = depth_pyramid_desc(*size), oracle.depth_pyramid_desc(*size)
    assert (d.width, d.height, d.mip_levels) == expect == (o.width, o.height, o.mip_levels)
    assert d.total_texels == o.total_texels and list(d.mip_offset) == list(o.mip_offset)


def test_shard_ranges_are_word_aligned_and_cover():
    from orbit_amd.dist import shard_ranges
    from orbit_amd.engine import shard_range

    for n in (0, 1, 31, 32, 33, 195_313, 1_000_003):
        for world in (1, 2, 3, 8):
            rs = shard_ranges(n, world)
            assert rs == [shard_range(n, r, world) for r in range(world)]  # python == C ABI
            assert rs[0][0] == 0 and rs[-1][1] == n
            for (b0, e0), (b1, e1) in zip(rs, rs[1:]):
                assert e0 == b1 and b1 % 32 == 0


def test_no_device_is_a_loud_error_not_a_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _lib.load()
    ctx = C.c_void_p()
    rc = lib.orbit_ctx_create(0, None, C.byref(ctx))
    assert rc == _lib.E_NO_DEVICE and not ctx.value
    assert b"no CPU fallback" in lib.orbit_last_error(None)
    with pytest.raises(_lib.OrbitError):
        from orbit_amd.engine import Engine

        Engine(0)


def test_meshlet_stream_entry_points_reject_null_without_a_device():
    """The derived-meshlet-stream surface (include/orbit_abi.h) validates before it touches a device."""
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.orbit_meshlet_stream_create(None, 0, 16, C.byref(h)) == _lib.E_INVALID and not h.value
    assert lib.orbit_meshlet_stream_update(None, None, None, 0, 0, None) == _lib.E_INVALID
    assert lib.orbit_ctx_bind_meshlet_stream(None, None) == _lib.E_INVALID
    assert lib.orbit_meshlet_stream_destroy(None) == _lib.OK
    assert lib.orbit_ctx_meshlet_stream_culls(None) == 0


def test_product_never_touches_the_oracle():
    """Nothing under orbit_amd/ may import, link or call oracle/ (task statement, section 3)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "orbit_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for line in text.splitlines():
                    code = line.split("//")[0].split("#")[0] if not f.endswith(".py") else line.split("#")[0]
                    assert not re.search(r"(import|from|include|dlopen|CDLL).*\boracle\b", code), (f, line)


def test_orbitscene_round_trip(tmp_path):
    """The `.orbitscene` dump format (real-asset path): sections come back typed and byte-identical."""
    import scenes as sc
    from orbit_amd import orbitscene

    s = sc.make_scene(5, 60)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, cam.planes)
    depth = sc.make_depth(1, 64, 36, cam)
    p = tmp_path / "s.orbitscene"
    orbitscene.write(p, entity_draws=s.entity_draw_buffer(), mesh_infos=s.mesh_infos, entities=s.entities,
                     meshlets=s.meshlets, materials=s.materials, cull_info=ci, depth=depth,
                     screen=np.array([64, 36], np.uint32), vendor_note=np.frombuffer(b"hello", np.uint8))
    d = orbitscene.read(p)
    assert d["meshlets"].dtype == L.MESHLET and np.array_equal(d["meshlets"], s.meshlets)
    assert np.array_equal(d["mesh_infos"], s.mesh_infos) and np.array_equal(d["entities"], s.entities)
    assert np.array_equal(d["materials"], s.materials)
    assert np.array_equal(d["entity_draws"], np.ascontiguousarray(s.entity_draw_buffer()).view(np.uint8).reshape(-1))
    assert d["cull_info"].tobytes() == ci.tobytes() and d["depth"].shape == (36, 64) and np.array_equal(d["depth"], depth)
    assert d["vendor_note"].tobytes() == b"hello"
    (tmp_path / "bad").write_bytes(b"not a scene file at all")
    with pytest.raises(ValueError):
        orbitscene.read(tmp_path / "bad")


def test_rust_binding_declares_every_entry_point():
    """bindings/rust/orbit_hip.rs is the text a maintainer adds on the reference side (no Rust toolchain here, so it
    is not compiled): at least its `extern "C"` block must name every function of include/orbit_abi.h."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "orbit_abi.h")).read()
    rust = open(os.path.join(root, "bindings", "rust", "orbit_hip.rs")).read()
    declared = set(re.findall(r"\b(orbit_[a-z0-9_]+)\s*\(", header))
    bound = set(re.findall(r"\bfn (orbit_[a-z0-9_]+)", rust))
    assert declared and declared <= bound, sorted(declared - bound)
