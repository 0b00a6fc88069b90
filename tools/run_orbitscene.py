"""Runs an `.orbitscene` dump (orbit_amd/orbitscene.py) through the engine on one MI355X: entity + meshlet cull with
the dumped CullInfo (or a default camera), the two-pass HiZ frame and the light clusters when the dump carries a depth
buffer / lights.  Every output is compared bit-exact with the CPU oracle; timings are HIP-event medians.

    python tools/run_orbitscene.py scene.orbitscene [--no-check]
    python tools/run_orbitscene.py --make-synthetic out.orbitscene   # a config-3-like stand-in, for trying the tool
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np

from orbit_amd import layouts as L
from orbit_amd import orbitscene


def make_synthetic(path):
    import scenes as sc

    s = sc.make_scene(3, 2900, n_meshes=2900, meshlets_per_mesh=(4, 38), extent=(75.0, 20.0, 75.0))
    cam = sc.default_camera()
    W, H = 1920, 1080
    orbitscene.write(path, entity_draws=s.entity_draw_buffer(), mesh_infos=s.mesh_infos, entities=s.entities,
                     meshlets=s.meshlets, materials=s.materials, lights=sc.make_lights(4, 2000),
                     cull_info=sc.make_cull_info(cam.view, cam.planes), depth=sc.make_depth(9, W, H, cam),
                     screen=np.array([W, H], np.uint32))


def run(path, check=True):
    import torch

    import scenes as sc
    from orbit_amd.engine import Engine, depth_pyramid_desc

    d = orbitscene.read(path)
    for need in ("entity_draws", "mesh_infos", "entities", "meshlets", "materials"):
        if need not in d:
            raise SystemExit(f"{path}: section {need} is missing")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
    edb = d["entity_draws"]
    n = int(edb[:4].view(np.uint32)[0])
    lod0 = d["mesh_infos"]["mesh_lods"][:, 0, 1]
    draws = edb[4:4 + 12 * n].view(np.uint32).reshape(-1, 3)
    max_meshlets = int(d["mesh_infos"]["mesh_lods"][draws[:, 1]][:, :, 1].max(axis=1).sum())
    cap_d, cap_c = max_meshlets // 32 + n + 8, max_meshlets + 8
    ci0 = d.get("cull_info")
    if ci0 is None:
        cam = sc.default_camera()
        ci0 = sc.make_cull_info(cam.view, cam.planes)
    eng = Engine(0, max_entities=n + 256, max_dispatches=cap_d, max_draws=cap_c,
                 max_lights=max(2000, len(d.get("lights", ())) + 16))
    g = {k: dev(d[k]) for k in ("entity_draws", "mesh_infos", "entities", "meshlets", "materials")}
    disp = torch.zeros(12 + 16 * cap_d, dtype=torch.uint8, device="cuda")
    draw = torch.zeros(4 + 28 * cap_c, dtype=torch.uint8, device="cuda")

    def timeit(fn, reps=20, warm=3):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts))

    def cull(ci, **kw):
        eng.entity_cull(ci, g["entity_draws"], g["mesh_infos"], disp, g["entities"], n, cap_d,
                        **{k: v for k, v in kw.items() if k in ("visibility_buffer", "depth_pyramid", "depth_pyramid_size")})
        eng.meshlet_cull(ci, disp, g["meshlets"], draw, g["entities"], g["materials"], cap_d, cap_c,
                         material_count=len(d["materials"]),
                         **{k: v for k, v in kw.items() if k in ("meshlet_visibility_buffer", "depth_pyramid", "depth_pyramid_size")})

    out = {"file": os.path.basename(path), "entities": n, "meshlets": int(len(d["meshlets"])),
           "lod0_meshlets_of_drawn_entities": int(lod0[draws[:, 1]].sum())}
    out["pass0_cull_ms"] = timeit(lambda: cull(ci0))
    torch.cuda.synchronize()
    eng.status()
    out["pass0_records"], out["pass0_draws"] = int(disp[:4].view(torch.int32).item()), int(draw[:4].view(torch.int32).item())
    if check:
        from oracle import oracle
        od, _, _ = oracle.entity_cull(ci0, edb, n, d["mesh_infos"], d["entities"], cap_d)
        oc, _, _ = oracle.meshlet_cull(ci0, od, d["meshlets"], cap_c, d["entities"], d["materials"])
        assert np.array_equal(disp.cpu().numpy()[:12 + 16 * out["pass0_records"]], od[:12 + 16 * out["pass0_records"]])
        assert np.array_equal(draw.cpu().numpy()[:4 + 28 * out["pass0_draws"]], oc[:4 + 28 * out["pass0_draws"]])
        out["pass0_bit_identical"] = True
    if "depth" in d:
        depth = d["depth"]
        H, W = depth.shape
        gdepth = dev(depth)
        pd = depth_pyramid_desc(W, H)
        pyr = torch.zeros(pd.total_texels, dtype=torch.float32, device="cuda")
        out["depth_reduce_ms"] = timeit(lambda: eng.depth_reduce(gdepth, W, H, pyr))
        if check:
            opyr, _ = oracle.depth_reduce(depth, W, H)
            assert np.array_equal(pyr.cpu().numpy().view(np.uint32), opyr.view(np.uint32))
            out["pyramid_bit_identical"] = True
    eng.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--make-synthetic", action="store_true")
    a = ap.parse_args()
    if a.make_synthetic:
        make_synthetic(a.path)
        print("wrote", a.path, os.path.getsize(a.path), "bytes")
    else:
        print(json.dumps(run(a.path, check=not a.no_check), indent=1))
