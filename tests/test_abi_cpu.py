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
HEADER = os.path.join(ROOT, "include", "orbit_abi.h")          # the drop-in core
HEADER_EXT = os.path.join(ROOT, "include", "orbit_abi_ext.h")  # everything beyond it
HEADERS = (HEADER, HEADER_EXT)


def declared_symbols(headers=HEADERS):
    names = set()
    for h in headers:
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(orbit_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 44
    # the core header alone is the drop-in of SURVEY.md §8b: the context, the four passes, the shard range, the gather
    core = declared_symbols((HEADER,))
    assert len(core) <= 20 and {"orbit_entity_cull", "orbit_meshlet_cull", "orbit_depth_reduce", "orbit_compute_clusters",
                                "orbit_gather_visible", "orbit_ctx_create", "orbit_last_error"} <= set(core)
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/orbit_abi*.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes signature in orbit_amd/_lib.py"
    assert lib.orbit_abi_version() == 6


def test_header_compiles_as_c_and_cxx(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "orbit_abi.h"\nint main(void){return sizeof(OrbitGpuCullInfo)==400?0:1;}\n')
    ext = tmp_path / "e.c"
    ext.write_text('#include "orbit_abi_ext.h"\nint main(void){return sizeof(OrbitCullView)>0&&sizeof(OrbitVisibleRecord)==12?0:1;}\n')
    for cc, std in (("gcc", "-std=c11"), ("g++", "-std=c++17")):
        exe = tmp_path / f"t_{cc}"
        for unit in (src, ext):
            subprocess.run([cc, std, "-x", "c" if cc == "gcc" else "c++", "-I", os.path.join(ROOT, "include"),
                            str(unit), "-o", str(exe)], check=True)
            assert subprocess.run([str(exe)]).returncode == 0


def test_ctypes_mirrors_match_header_sizes():
    # the sizes below are the ORBIT_STATIC_ASSERTs of the header
    assert C.sizeof(_lib.Caps) == 40
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


def c_prototypes():
    """{name: (return, [param, ...])} of every function both headers declare; a scalar is its fixed-width type
    ("i32", "u32", "u64", "f32"), a pointer is ("ptr", depth, pointee is const) — `T name[N]` parameters are pointers."""
    scalars = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "float": "f32", "void": "void"}

    def classify(decl, is_param):
        decl = decl.strip()
        depth = decl.count("*") + (1 if "[" in decl else 0)
        # is what the (outermost) pointer points to const?  `void *const *p`: the text in front of the last `*` ends in
        # const; `const T *p` / `const T p[N]`: the base type is const and there is one level
        before_last = decl.rsplit("*", 1)[0].rstrip() if "*" in decl else ""
        decl = re.sub(r"\[[^\]]*\]", "", decl).replace("*", " ")
        words = decl.split()
        if is_param and len(words) > 1 and words[-1] not in scalars and not words[-1].startswith("Orbit"):
            words = words[:-1]  # the parameter's name
        const = before_last.endswith("const") or (depth == 1 and words[0] == "const")
        base = [w for w in words if w != "const"][0]
        if depth:
            return ("ptr", depth, const)
        assert base in scalars, decl
        return scalars[base]

    out = {}
    for h in HEADERS:
        text = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        for ret, name, params in re.findall(r"([A-Za-z_][\w \t\*]*?)\b(orbit_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
            ps = [] if params.strip() in ("", "void") else [classify(q, True) for q in params.split(",")]
            out[name] = (classify(ret, False), ps)
    return out


def rust_prototypes():
    """The same out of the `extern "C"` block of bindings/rust/orbit_hip.rs."""
    rust = open(os.path.join(ROOT, "bindings", "rust", "orbit_hip.rs")).read()
    rust = re.sub(r"//[^\n]*", "", rust)

    def classify(ty):
        ty = ty.strip()
        depth, const = 0, False
        while ty.startswith("*"):
            kind, ty = ty.split(None, 1)
            if depth == 0:
                const = kind == "*const"
            depth += 1
        return ("ptr", depth, const) if depth else ty

    out = {}
    for name, params, ret in re.findall(r"\bfn (orbit_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;{]+))?;", rust):
        ps = [classify(q.split(":", 1)[1]) for q in params.split(",") if q.strip()]
        out[name] = (classify(ret) if ret.strip() else "void", ps)
    return out


def test_rust_binding_matches_the_headers_signature_by_signature():
    """bindings/rust/orbit_hip.rs is the text a maintainer pastes on the reference side; no Rust toolchain compiles it
    here, so this is what keeps it from drifting: every function of include/orbit_abi.h and orbit_abi_ext.h is declared
    with the same number of parameters, the same fixed-width scalars in the same places, pointers of the same depth
    (and the same constness of what they point to), and the same return type."""
    c, r = c_prototypes(), rust_prototypes()
    assert len(c) >= 44 and set(c) == set(declared_symbols())
    assert set(c) <= set(r), sorted(set(c) - set(r))
    assert set(r) <= set(c), f"bound but not declared in the headers: {sorted(set(r) - set(c))}"
    for name, (ret, params) in sorted(c.items()):
        rret, rparams = r[name]
        assert len(params) == len(rparams), f"{name}: {len(params)} parameters in C, {len(rparams)} in Rust"
        for k, (a, b) in enumerate(zip(params, rparams)):
            assert a == b, f"{name}, parameter {k}: C {a} vs Rust {b}"
        assert ret == rret or (ret[0] == "ptr" and rret[0] == "ptr" and ret[1] == rret[1]), f"{name}: returns {ret} vs {rret}"


def test_ctypes_table_matches_the_headers_signature_by_signature():
    """orbit_amd/_lib.py SYMBOLS against the same prototypes: parameter count, scalar width, pointer-ness."""
    widths = {C.c_int32: "i32", C.c_uint32: "u32", C.c_uint64: "u64", C.c_float: "f32", None: "void"}
    for name, (ret, params) in sorted(c_prototypes().items()):
        restype, argtypes = _lib.SYMBOLS[name]
        assert len(argtypes) == len(params), f"{name}: {len(params)} parameters in C, {len(argtypes)} in ctypes"
        for k, (a, ty) in enumerate(zip(params, argtypes)):
            is_ptr = ty in (C.c_void_p, C.c_char_p) or hasattr(ty, "contents")
            assert (a[0] == "ptr") == is_ptr if isinstance(a, tuple) else (not is_ptr and widths[ty] == a), (name, k, a, ty)
        if isinstance(ret, tuple):
            assert restype in (C.c_void_p, C.c_char_p), name
        else:
            assert widths[restype] == ret, (name, ret, restype)
