"""Triage of the one-launch cull (GPU box): one small scene through orbit_cull_views on the one-launch path, every step
printed as it happens (run under `timeout`), outputs compared with the launch chain's."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import scenes as sc
from orbit_amd import layouts as L
from orbit_amd.engine import Engine

VARIANT = os.path.join(ROOT, "tools", "variants", "fused_dbg.so")  # bash tools/mkvariant_any.sh fused_dbg cull_fused,abi -DORBIT_FUSED_DEBUG
DBG = None
if os.path.exists(VARIANT) and os.environ.get("ORBIT_FUSED_MARKS") == "1":
    import ctypes as C

    from orbit_amd import _lib

    DBG = _lib.load_variant(VARIANT)
    DBG.orbit_debug_read_fused_marks.restype = C.c_int32
    DBG.orbit_debug_read_fused_marks.argtypes = [C.c_void_p, C.c_void_p]
    DBG.orbit_ctx_fused_culls.restype = C.c_uint64
    DBG.orbit_ctx_fused_culls.argtypes = [C.c_void_p]


def marks(eng):
    out = np.zeros(8192, dtype=np.uint64)
    rc = DBG.orbit_debug_read_fused_marks(eng._ctx, out.ctypes.data)
    rows = out.reshape(-1, 8)
    names = ["ent ticket", "chunk total", "base", "records out", "nrec", "tile ticket", "tile count", "left"]
    if os.environ.get("ORBIT_FUSED_STAMPS") == "1":  # a -DORBIT_FUSED_DEBUG=2 build: times, us from the first mark
        vals = rows & np.uint64(0x7FFFFFFFFFFFFFFF)
        t0 = int(vals[rows >> np.uint64(63) != 0].min())
        for w, r in enumerate(rows):
            if r.any():
                print(f"  block {w // 4:3d} wave {w % 4}: " + ", ".join(
                    f"{n}={(int(v & 0x7FFFFFFFFFFFFFFF) - t0) / 100.0:6.2f}" for n, v in zip(names, r) if v >> 63), flush=True)
        return rc
    for w, r in enumerate(rows):
        if r.any():
            print(f"  block {w // 4} wave {w % 4}: " + ", ".join(
                f"{n}={int(v & 0x7FFFFFFFFFFFFFFF)}" for n, v in zip(names, r) if v >> 63), flush=True)
    return rc


def say(*a):
    print(f"[{time.time() - T0:7.2f}]", *a, flush=True)


T0 = time.time()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
occ = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
scene = sc.make_scene(5, n, n_meshes=max(1, n // 3), meshlets_per_mesh=(1, 70))
cam = sc.default_camera(rot=(0.6, 0.8))
ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=occ)
say("scene", n, "entities", scene.lod0_meshlets, "meshlets")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()


bufs = dict(draws=dev(scene.entity_draw_buffer()), mesh=dev(scene.mesh_infos), ent=dev(scene.entities),
            mlt=dev(scene.meshlets), mat=dev(scene.materials))
cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
rng = np.random.default_rng(1)
evis = rng.integers(0, 2 ** 32, (n + 31) // 32, dtype=np.uint32)
mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
out = {}
for name, path in (("chain", 1), ("fused", 2)):
    eng = Engine(0, _library=DBG, max_entities=max(n, 256) + 256, max_dispatches=cap_d + 64, max_draws=cap_c + 64,
                 cull_path=path)
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
    draw = torch.zeros(L.DRAW_HEADER + 28 * cap_c, dtype=torch.uint8, device="cuda")
    v = dict(cull_info=ci, entity_draw_buffer=bufs["draws"], mesh_info_buffer=bufs["mesh"], meshlet_dispatch_buffer=disp,
             entity_buffer=bufs["ent"], entity_draw_count=n, dispatch_capacity=cap_d, meshlet_buffer=bufs["mlt"],
             draw_commands_buffer=draw, material_buffer=bufs["mat"], draw_capacity=cap_c,
             visibility_buffer=dev(evis) if occ else None, meshlet_visibility_buffer=dev(mvis) if occ else None)
    for r in range(reps):
        say(name, "launch", r)
        t = time.time()
        eng.cull_views([v])
        if DBG is not None and name == "fused" and r == reps - 1:
            time.sleep(1.0)
            say("marks 1 s after the launch (rc", marks(eng), ")")
        torch.cuda.synchronize()
        say(name, "done in", f"{(time.time() - t) * 1e3:.2f} ms")
        try:
            eng.status()
        except Exception as ex:  # noqa: BLE001
            say(name, "status:", ex)
    out[name] = (disp.cpu().numpy(), draw.cpu().numpy())
    say(name, "records", int(out[name][0][:4].view(np.uint32)[0]), "commands", int(out[name][1][:4].view(np.uint32)[0]),
        "fused culls", eng.fused_culls())
    eng.close()
nr, nc = int(out["chain"][0][:4].view(np.uint32)[0]), int(out["chain"][1][:4].view(np.uint32)[0])
say("records equal:", np.array_equal(out["chain"][0][:12 + 16 * nr], out["fused"][0][:12 + 16 * nr]),
    "commands equal:", np.array_equal(out["chain"][1][:4 + 28 * nc], out["fused"][1][:4 + 28 * nc]))
