"""orbit_gather_visible (C ABI, RCCL) on one GPU: a communicator of one rank.  The multi-rank
fan-out of the same entry point needs more GPUs than a test box has; its rank-order/prefix logic is
the one tests/test_dist_cpu.py exercises with two gloo ranks through orbit_amd/dist.py."""
import ctypes as C
import os

import numpy as np
import pytest

import scenes as sc
from orbit_amd import layouts as L

pytestmark = pytest.mark.gpu


class NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


@pytest.fixture(scope="module")
def comm_of_one():
    import torch

    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")  # HIP context before RCCL
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
    uid = NcclUniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    yield comm
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)


def test_gather_visible_world_of_one(comm_of_one):
    import torch

    from orbit_amd._lib import E_CAPACITY, E_INVALID, OrbitError
    from orbit_amd.engine import Engine

    scene = sc.make_scene(21, 400)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, np.zeros((0, 4), np.float32), alpha_mode_flag=L.ALPHA_ALL)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
    eng = Engine(0)
    disp_cap, draw_cap = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device="cuda")
    draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device="cuda")
    ents, mats = dev(scene.entities), dev(scene.materials)
    eng.entity_cull(ci, dev(scene.entity_draw_buffer()), dev(scene.mesh_infos), disp, ents, scene.entity_draw_count,
                    disp_cap)
    eng.meshlet_cull(ci, disp, dev(scene.meshlets), draw, ents, mats, disp_cap, draw_cap)
    torch.cuda.synchronize()
    n = int(draw[:4].view(torch.int32).item())
    assert n > 100
    out = torch.full((L.DRAW_HEADER + 28 * draw_cap + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    eng.gather_visible(comm_of_one, 0, 1, draw, out, draw_cap)
    torch.cuda.synchronize()
    assert torch.equal(out[:4 + 28 * n], draw[:4 + 28 * n])
    assert bool((out[4 + 28 * n:] == 0xCD).all())  # nothing written past the list
    with pytest.raises(OrbitError) as ei:  # the concatenation must fit
        eng.gather_visible(comm_of_one, 0, 1, draw, out, n - 1)
    assert ei.value.code == E_CAPACITY
    with pytest.raises(OrbitError) as ei:
        eng.gather_visible(comm_of_one, 1, 1, draw, out, draw_cap)
    assert ei.value.code == E_INVALID
    eng.close()
