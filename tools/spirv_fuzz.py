"""Differential fuzz of the oracle against the reference's compiled cull shaders (build container only: reads
/root/reference/shaders/*.spv through oracle/spirv_vm.py).  Every seed runs occlusion passes 0 / 1 / 2 x perspective /
orthographic on a knife-edge scene (tests/golden/make_spirv_vectors.py: quantised geometry, cone radii within an ulp
of equality, pass 2 with a constant depth buffer bisected to the closest-depth of a group of identical spheres) and
compares dispatch records, draw commands and visibility words.  The meshlet stage runs on the ORACLE's records so that
one stage's difference does not hide the other's.
usage: python tools/spirv_fuzz.py FIRST_SEED LAST_SEED [--plain]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_spirv_vectors as gen  # noqa: E402  (puts the repo root and tests/ on the path)
import scenes as sc  # noqa: E402
from oracle import oracle, spirv_vm as vm  # noqa: E402
from orbit_amd import layouts as L  # noqa: E402

F = np.float32


def run(seed, occlusion_pass, ortho, knife, n=400):
    rng = np.random.default_rng(seed)
    scene = sc.make_scene(seed, n, meshlets_per_mesh=(1, 70), lods=3)
    cam = sc.default_camera(rot=(0.2, 0.4))
    if knife:
        gen.quantise(scene, rng)
        cam = sc.default_camera(position=(0.0, 0.0, float(rng.integers(4, 30))), rot=(1.0, 0.0))
        gen.cone_knife_edge(scene, cam, ortho)
    kw = {}
    if occlusion_pass == 2:
        kw = gen.ORTHO if ortho else dict(p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
    ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=occlusion_pass, projection_type=1 if ortho else 0, **kw)
    depth = sc.make_depth(9 + seed, gen.W, gen.H, cam)
    evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    if knife and occlusion_pass == 2:
        depth = gen.depth_knife_edge(scene, cam, ci, ortho, seed, rng)
        evis[:], mvis[:] = 0, 0
    pyr, dsc = oracle.depth_reduce(depth, gen.W, gen.H)
    ps = (dsc.width, dsc.height)
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    pk = (pyr, ps) if occlusion_pass == 2 else ()
    od, oev, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                    scene.entities, cap_d, evis if occlusion_pass else None, *pk)
    oc, omv, _ = oracle.meshlet_cull(ci, od, scene.meshlets, cap_c, scene.entities, scene.materials,
                                     mvis if occlusion_pass else None, *pk)
    B = {}

    def buf(i, a):
        B[i] = np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()
    civ = np.array(ci).copy().reshape(())
    if occlusion_pass:
        civ["visibility_buffer"], civ["meshlet_visibility_buffer"], civ["depth_pyramid"] = 6, 7, 11
    buf(1, scene.entity_draw_buffer()), buf(2, scene.mesh_infos), buf(4, scene.entities), buf(5, civ.reshape(1))
    B[3] = np.zeros(L.DISPATCH_HEADER + 16 * cap_d, np.uint8)
    B[3][4:12] = np.array([1, 1], np.uint32).view(np.uint8)
    buf(6, evis.copy()), buf(7, mvis.copy()), buf(8, scene.meshlets), buf(10, scene.materials)
    B[9] = np.zeros(L.DRAW_HEADER + 28 * cap_c, np.uint8)
    kwm = dict(images={11: {"size": lambda lod: (max(ps[0] >> lod, 1), max(ps[1] >> lod, 1))}},
               samplers={k: {"id": k} for k in range(8)}, subgroup=32,
               sample=lambda img, smp, u, v, lod: np.array(
                   [oracle.hiz_sample(pyr, ps[0], ps[1], float(u), float(v), float(lod)), 0, 0, 0], F))
    vm.Machine(gen.module("entity_cull.comp.spv"), B, np.array([1, 2, 3, 4, 5], np.uint32).tobytes(), gen.log2,
               **kwm).run((scene.entity_draw_count + 255) // 256, batch=8)
    nrec, onrec = int(B[3][:4].view(np.uint32)[0]), int(od[:4].view(np.uint32)[0])
    ok = [nrec == onrec and np.array_equal(B[3][:12 + 16 * nrec], od[:12 + 16 * onrec]),
          (not occlusion_pass) or np.array_equal(B[6].view(np.uint32), oev)]
    B[3] = od.copy()
    vm.Machine(gen.module("meshlet_cull.comp.spv"), B, np.array([3, 8, 9, 4, 5, 10], np.uint32).tobytes(), gen.log2,
               **kwm).run(onrec, batch=256)
    nc, onc = int(B[9][:4].view(np.uint32)[0]), int(oc[:4].view(np.uint32)[0])
    ok += [nc == onc and np.array_equal(B[9][:4 + 28 * nc], oc[:4 + 28 * onc]),
           (not occlusion_pass) or np.array_equal(B[7].view(np.uint32), omv)]
    return ok, onc


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    knife = "--plain" not in sys.argv
    bad = draws = 0
    t0 = time.time()
    for seed in range(first, last):
        for p in (0, 1, 2):
            for ortho in (False, True):
                try:
                    ok, n = run(seed, p, ortho, knife)
                except AssertionError as e:  # a knife-edge scene whose group is not decided by the depth buffer
                    print("skipped seed", seed, "pass", p, "ortho", ortho, "-", e)
                    continue
                draws += n
                if not all(ok):
                    bad += 1
                    print("MISMATCH seed", seed, "pass", p, "ortho", ortho, dict(zip(("dispatch", "evis", "draws", "mvis"), ok)))
    print("runs", (last - first) * 6, "mismatching", bad, "draw commands compared", draws, "seconds", round(time.time() - t0, 1))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
