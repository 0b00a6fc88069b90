"""How much of the reference binaries' output depends on what SPIR-V leaves to the driver?  Re-runs the 12 committed
cull cases (tests/golden/spirv_cull.npz) through oracle/spirv_vm.py with OpDot / OpMatrixTimesVector / Length evaluated
as fma chains — what GPU drivers' compilers typically emit — instead of the canonical separately rounded sums, and
counts the dispatch records, draw commands and visibility words that change.  Build container only (reads
/root/reference/shaders).  usage: python tools/spirv_sensitivity.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_spirv_vectors as gen  # noqa: E402
from oracle import oracle, spirv_vm as vm  # noqa: E402
from orbit_amd import layouts as L  # noqa: E402

F = np.float32
st = np.load(os.path.join(ROOT, "tests", "golden", "spirv_cull.npz"))
names = sorted({k.split("/")[0] for k in st.files})
tot = dict(records=0, records_changed=0, draws=0, draws_changed=0)
for name in names:
    g = lambda k: st[f"{name}/{k}"]
    ci = g("cull_info").view(L.GPU_CULL_INFO).reshape(()).copy()
    p = int(ci["occlusion_pass"])
    if p:
        ci["visibility_buffer"], ci["meshlet_visibility_buffer"], ci["depth_pyramid"] = 6, 7, 11
    pyr, ps = g("pyramid"), tuple(int(v) for v in g("pyramid_size"))
    cap_d, cap_c = (int(v) for v in g("caps"))
    B = {1: g("entity_draws").copy(), 2: g("mesh_infos").view(np.uint8).reshape(-1).copy(),
         4: g("entities").view(np.uint8).reshape(-1).copy(), 5: np.array(ci).reshape(1).view(np.uint8).reshape(-1).copy(),
         6: g("evis_in").view(np.uint8).reshape(-1).copy(), 7: g("mvis_in").view(np.uint8).reshape(-1).copy(),
         8: g("meshlets").view(np.uint8).reshape(-1).copy(), 10: g("materials").view(np.uint8).reshape(-1).copy(),
         3: np.zeros(L.DISPATCH_HEADER + 16 * cap_d, np.uint8), 9: np.zeros(L.DRAW_HEADER + 28 * cap_c, np.uint8)}
    B[3][4:12] = np.array([1, 1], np.uint32).view(np.uint8)
    kw = dict(images={11: {"size": lambda lod: (max(ps[0] >> lod, 1), max(ps[1] >> lod, 1))}},
              samplers={k: {"id": k} for k in range(8)}, subgroup=32, fused_dot=True,
              sample=lambda img, smp, u, v, lod: np.array(
                  [oracle.hiz_sample(pyr, ps[0], ps[1], float(u), float(v), float(lod)), 0, 0, 0], F))
    n_draws = int(np.frombuffer(B[1][:4].tobytes(), np.uint32)[0])
    vm.Machine(gen.module("entity_cull.comp.spv"), B, np.array([1, 2, 3, 4, 5], np.uint32).tobytes(), gen.log2, **kw).run(
        (n_draws + 255) // 256, batch=8)
    want_d = g("spv_dispatch")
    nrec, wrec = int(B[3][:4].view(np.uint32)[0]), int(want_d[:4].view(np.uint32)[0])
    a = {bytes(r) for r in B[3][12:12 + 16 * nrec].reshape(-1, 16)}
    b = {bytes(r) for r in want_d[12:12 + 16 * wrec].reshape(-1, 16)}
    B[3][:len(want_d)] = want_d  # the meshlet stage on the canonical records: one stage's change at a time
    B[3][len(want_d):] = 0
    vm.Machine(gen.module("meshlet_cull.comp.spv"), B, np.array([3, 8, 9, 4, 5, 10], np.uint32).tobytes(), gen.log2,
               **kw).run(wrec, batch=256)
    want_c = g("spv_draw")
    nc, wc = int(B[9][:4].view(np.uint32)[0]), int(want_c[:4].view(np.uint32)[0])
    c = {bytes(r) for r in B[9][4:4 + 28 * nc].reshape(-1, 28)}
    d = {bytes(r) for r in want_c[4:4 + 28 * wc].reshape(-1, 28)}
    print("%-16s records %4d (%d differ)   draws %5d (%d differ)" % (name, wrec, len(a ^ b), wc, len(c ^ d)))
    tot["records"] += wrec; tot["records_changed"] += len(a ^ b); tot["draws"] += wc; tot["draws_changed"] += len(c ^ d)
print(tot)
