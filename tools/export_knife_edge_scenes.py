"""Writes the knife-edge cull cases of tests/golden/spirv_cull.npz as `.orbitscene` dumps (orbit_amd/orbitscene.py) plus,
beside each, what BOTH arithmetic profiles expect (`<case>.expected.npz`: dispatch records, draw commands and — in the
occlusion passes — the two visibility bitsets, canonical and contracted): the inputs on which a real Vulkan driver's result
tells which profile it implements (docs/vulkan_pin.md).  No GPU, no reference checkout needed.
usage: python tools/export_knife_edge_scenes.py out_dir"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orbit_amd import orbitscene  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "knife_edge_scenes"
    os.makedirs(out, exist_ok=True)
    g = os.path.join(ROOT, "tests", "golden")
    canon, contr = np.load(os.path.join(g, "spirv_cull.npz")), np.load(os.path.join(g, "spirv_cull_contracted.npz"))
    for p in (0, 1, 2):
        for proj in ("persp", "ortho"):
            name = f"p{p}_{proj}_knife"
            k = lambda s: canon[f"{name}/{s}"]
            sections = dict(entity_draws=k("entity_draws"), mesh_infos=k("mesh_infos"), entities=k("entities"),
                            meshlets=k("meshlets"), materials=k("materials"), cull_info=k("cull_info"))
            if p:  # last frame's bits and (pass 2) the pyramid the binaries sampled, as extra sections
                sections.update(entity_visibility=k("evis_in"), meshlet_visibility=k("mvis_in"))
            if p == 2:
                sections.update(depth_pyramid=k("pyramid"), depth_pyramid_size=k("pyramid_size"))
            orbitscene.write(os.path.join(out, name + ".orbitscene"), **sections)
            exp = {}
            for s in ("spv_dispatch", "spv_draw", "spv_evis", "spv_mvis"):
                exp["canonical_" + s[4:]], exp["contracted_" + s[4:]] = canon[f"{name}/{s}"], contr[f"{name}/{s}"]
            np.savez_compressed(os.path.join(out, name + ".expected.npz"), **exp)
            d = sum(int(not np.array_equal(exp["canonical_" + s], exp["contracted_" + s])) for s in ("dispatch", "draw", "evis", "mvis"))
            print(f"{name}: {int(exp['canonical_draw'][:4].view(np.uint32)[0])} canonical / "
                  f"{int(exp['contracted_draw'][:4].view(np.uint32)[0])} contracted draw commands; {d} of 4 outputs differ between the profiles")


main()
