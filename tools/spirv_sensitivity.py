"""How much of the reference binaries' output depends on what SPIR-V leaves to the driver?  Re-runs the 12 committed
cull cases (tests/golden/spirv_cull.npz) through oracle/spirv_vm.py once per driver-defined point, each varied alone
against the canonical choice, and counts the dispatch records and draw commands that change:

  dot_fma               OpDot / OpMatrixTimesVector / Length as fma chains (what GPU compilers typically emit)
  fma_unfused           GLSL.std.450 Fma as a rounded product + a rounded sum (legal: the binaries' Fma carry no
                        NoContraction decoration)
  log2_plus_ulp / _minus_ulp   Log2 one ulp above / below the shared software log2 (HiZ mip choice, LOD pick)
  footprint_nonzero     the ReduceMin sampler takes only texels with a non-zero bilinear weight (the canonical model
                        takes the whole clamped 2x2 footprint)

The append order of atomics / the subgroup size change no SET of records (the lists are compared as sets here, and the
product's order is the canonical one by construction).  Build container only (reads /root/reference/shaders).
usage: python tools/spirv_sensitivity.py [out.json]   (default profiles/r03_driver_envelope.json)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_spirv_vectors as gen  # noqa: E402
import np_restatement as npr  # noqa: E402
from oracle import oracle, spirv_vm as vm  # noqa: E402
from orbit_amd import layouts as L  # noqa: E402

F = np.float32


def footprint_nonzero(pyr, w0, h0, u, v, lod):
    """The canonical sampler (oracle_hiz_sample: nearest mip = clamp(ceil(lod + 0.5) - 1), clamped 2x2 footprint)
    restricted to the texels whose bilinear weight is not zero."""
    mips, offs, ws, hs, _ = npr.pyramid_levels(w0, h0)
    with np.errstate(all="ignore"):
        lf = F(np.ceil(F(F(lod) + F(0.5)))) - F(1.0)
    level = int(npr.f2i_clamp(np.array([lf], F), mips - 1)[0])
    w, h, off = int(ws[level]), int(hs[level]), int(offs[level])
    x, y = F(F(u) * F(w)) - F(0.5), F(F(v) * F(h)) - F(0.5)
    fx, fy = np.floor(x), np.floor(y)
    tx, ty = F(x - fx), F(y - fy)  # weights: (1 - t) for the lower texel, t for the upper one
    cl = lambda f, hi: int(npr.f2i_clamp(np.array([f], F), hi)[0])
    xs = [cl(fx, w - 1)] + ([cl(F(fx + F(1)), w - 1)] if not (tx == 0) else [])  # NaN: both (as the canonical model)
    ys = [cl(fy, h - 1)] + ([cl(F(fy + F(1)), h - 1)] if not (ty == 0) else [])
    vals = [pyr[off + yy * w + xx] for yy in ys for xx in xs]
    m = vals[0]
    for t in vals[1:]:
        m = t if t < m else m
    return m


def ulp_shift(x, up):
    x = np.asarray(x, F)
    return np.nextafter(x, F(np.inf) if up else F(-np.inf)).astype(F)


VARIANTS = {
    "dot_fma": dict(fused_dot=True),
    "fma_unfused": dict(unfused_fma=True),
    "log2_plus_ulp": dict(log2=lambda x: ulp_shift(gen.log2(x), True)),
    "log2_minus_ulp": dict(log2=lambda x: ulp_shift(gen.log2(x), False)),
    "footprint_nonzero": dict(sampler=footprint_nonzero),
}


def run_case(st, name, fused_dot=False, unfused_fma=False, log2=None, sampler=None):
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
    smp = sampler or (lambda pyr, w, h, u, v, lod: oracle.hiz_sample(pyr, w, h, float(u), float(v), float(lod)))
    kw = dict(images={11: {"size": lambda lod: (max(ps[0] >> lod, 1), max(ps[1] >> lod, 1))}},
              samplers={k: {"id": k} for k in range(8)}, subgroup=32, fused_dot=fused_dot, unfused_fma=unfused_fma,
              sample=lambda img, s, u, v, lod: np.array([smp(pyr, ps[0], ps[1], u, v, lod), 0, 0, 0], F))
    lg = log2 or gen.log2
    n_draws = int(np.frombuffer(B[1][:4].tobytes(), np.uint32)[0])
    vm.Machine(gen.module("entity_cull.comp.spv"), B, np.array([1, 2, 3, 4, 5], np.uint32).tobytes(), lg, **kw).run(
        (n_draws + 255) // 256, batch=8)
    want_d = g("spv_dispatch")
    nrec, wrec = int(B[3][:4].view(np.uint32)[0]), int(want_d[:4].view(np.uint32)[0])
    a = {bytes(r) for r in B[3][12:12 + 16 * nrec].reshape(-1, 16)}
    b = {bytes(r) for r in want_d[12:12 + 16 * wrec].reshape(-1, 16)}
    B[3][:len(want_d)] = want_d  # the meshlet stage on the canonical records: one stage's change at a time
    B[3][len(want_d):] = 0
    vm.Machine(gen.module("meshlet_cull.comp.spv"), B, np.array([3, 8, 9, 4, 5, 10], np.uint32).tobytes(), lg, **kw).run(
        wrec, batch=256)
    want_c = g("spv_draw")
    nc, wc = int(B[9][:4].view(np.uint32)[0]), int(want_c[:4].view(np.uint32)[0])
    c = {bytes(r) for r in B[9][4:4 + 28 * nc].reshape(-1, 28)}
    d = {bytes(r) for r in want_c[4:4 + 28 * wc].reshape(-1, 28)}
    return dict(records=wrec, records_differ=len(a ^ b), draws=wc, draws_differ=len(c ^ d))


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_driver_envelope.json")
    st = np.load(os.path.join(ROOT, "tests", "golden", "spirv_cull.npz"))
    names = sorted({k.split("/")[0] for k in st.files})
    base = {n: run_case(st, n) for n in names}
    assert all(v["records_differ"] == 0 and v["draws_differ"] == 0 for v in base.values()), "canonical run != vectors"
    out = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "vectors": "tests/golden/spirv_cull.npz", "variants": {}}
    for vname, kw in VARIANTS.items():
        per, tot = {}, dict(records=0, records_differ=0, draws=0, draws_differ=0)
        for n in names:
            if vname == "footprint_nonzero" and not n.startswith("p2"):
                per[n] = dict(base[n])  # only pass 2 samples the pyramid
            else:
                per[n] = run_case(st, n, **kw)
            for k in tot:
                tot[k] += per[n][k]
            print("%-20s %-16s records %4d (%d differ)   draws %5d (%d differ)" % (
                vname, n, per[n]["records"], per[n]["records_differ"], per[n]["draws"], per[n]["draws_differ"]))
        out["variants"][vname] = {"total": tot, "per_case": per}
        print(vname, tot)
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print(out_path)


if __name__ == "__main__":
    main()
