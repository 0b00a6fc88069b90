"""The PRODUCT against outputs of the reference's own compiled shaders (tests/golden/spirv_cull.npz; see
tests/test_spirv_vectors_cpu.py): entity_cull + meshlet_cull through the C ABI, from the Meshlet buffer and from the
derived meshlet streams, must reproduce the binaries' dispatch records, draw commands and visibility words."""
import numpy as np
import pytest

from orbit_amd import layouts as L
from test_gpu_parity import dev, host, torch_mod  # noqa: F401
from test_spirv_vectors_cpu import CASES, load_case, vectors  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("source", ["meshlet_buffer", "meshlet_stream"])
@pytest.mark.parametrize("name", CASES)
def test_product_equals_the_reference_binaries(torch_mod, vectors, name, source):
    from orbit_amd.engine import Engine

    torch = torch_mod
    c = load_case(vectors, name)
    p = int(c["ci"]["occlusion_pass"])
    n_draws = int(np.frombuffer(c["draws"][:4].tobytes(), np.uint32)[0])
    cap_d, cap_c = c["caps"]
    eng = Engine(0, max_entities=4096, max_dispatches=cap_d + 64, max_draws=cap_c + 64)
    g = {k: dev(torch, c[k]) for k in ("draws", "mesh_infos", "entities", "meshlets", "materials")}
    if source == "meshlet_stream":
        eng.bind_meshlet_stream(eng.meshlet_stream(g["meshlets"], 0, len(c["meshlets"])))
    evis = dev(torch, c["evis"]) if p else None
    mvis = dev(torch, c["mvis"]) if p else None
    pyr = dev(torch, c["pyr"]) if p == 2 else None
    ps = c["ps"] if p == 2 else (0, 0)
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * cap_d, dtype=torch.uint8, device="cuda")
    draw = torch.zeros(L.DRAW_HEADER + 28 * cap_c, dtype=torch.uint8, device="cuda")
    eng.entity_cull(c["ci"], g["draws"], g["mesh_infos"], disp, g["entities"], n_draws, cap_d, visibility_buffer=evis,
                    depth_pyramid=pyr, depth_pyramid_size=ps)
    eng.meshlet_cull(c["ci"], disp, g["meshlets"], draw, g["entities"], g["materials"], cap_d, cap_c,
                     meshlet_visibility_buffer=mvis, depth_pyramid=pyr, depth_pyramid_size=ps,
                     material_count=len(c["materials"]))
    torch.cuda.synchronize()
    eng.status()
    nrec, ndraw = int(c["spv_dispatch"][:4].view(np.uint32)[0]), int(c["spv_draw"][:4].view(np.uint32)[0])
    assert np.array_equal(host(disp)[:L.DISPATCH_HEADER + 16 * nrec], c["spv_dispatch"]), "dispatch records differ"
    assert np.array_equal(host(draw)[:L.DRAW_HEADER + 28 * ndraw], c["spv_draw"]), "draw commands differ"
    if p:
        assert np.array_equal(host(evis, np.uint32), c["spv_evis"]) and np.array_equal(host(mvis, np.uint32), c["spv_mvis"])
    assert (eng.meshlet_stream_culls() > 0) == (source == "meshlet_stream" and p != 1)
    eng.close()
