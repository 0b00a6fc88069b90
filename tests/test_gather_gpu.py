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


def _record_list(torch, eng, scene, ci, cap_rec):
    """This scene's record list through the product (orbit_meshlet_cull_visible_records) in a buffer of cap_rec entries."""
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
    disp_cap = scene.max_dispatches() + 8
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device="cuda")
    rec = torch.full((L.VISIBLE_HEADER + 12 * cap_rec,), 0x33, dtype=torch.uint8, device="cuda")
    ents, mats = dev(scene.entities), dev(scene.materials)
    eng.entity_cull(ci, dev(scene.entity_draw_buffer()), dev(scene.mesh_infos), disp, ents, scene.entity_draw_count,
                    disp_cap)
    eng.meshlet_cull_visible_records(ci, disp, dev(scene.meshlets), rec, ents, mats, disp_cap, cap_rec)
    torch.cuda.synchronize()
    eng.status()
    return rec


def test_allgather_list_world_of_one(comm_of_one):
    """north_star's transport with no host in it (orbit_allgather_list): ONE ncclAllGather of the rank's whole list buffer
    as a fixed-capacity segment, then the compaction launch — through a real communicator (of one rank: what a test box
    has); the gathered list is the local one byte for byte, nothing is written behind it, and the call never waits."""
    import torch

    from orbit_amd._lib import E_CAPACITY, OrbitError
    from orbit_amd.engine import Engine

    scene = sc.make_scene(23, 500)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, cam.planes, alpha_mode_flag=L.ALPHA_ALL)
    eng = Engine(0)
    cap = scene.max_dispatches() + 8
    rec = _record_list(torch, eng, scene, ci, cap)
    n = int(rec[:4].view(torch.int32).item())
    assert 100 < n <= cap
    seg_bytes = L.VISIBLE_HEADER + 12 * cap
    segments = torch.full((seg_bytes,), 0x44, dtype=torch.uint8, device="cuda")
    out = torch.full((L.VISIBLE_HEADER + 12 * cap + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    eng.allgather_list(comm_of_one, 0, 1, rec, cap, segments, out, cap, L.VISIBLE_HEADER, 12)
    torch.cuda.synchronize()
    eng.status()
    assert torch.equal(segments, rec)  # the whole segment travelled, slack included
    assert int(out[:4].view(torch.int32).item()) == n and int(out[4:8].view(torch.int32).item()) == 0
    assert torch.equal(out[8:8 + 12 * n], rec[8:8 + 12 * n]) and bool((out[8 + 12 * n:] == 0xCD).all())
    # an output that cannot hold the list: a prefix, and the overflow is latched (no host check could know the count)
    out2 = torch.full((L.VISIBLE_HEADER + 12 * (n - 5) + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    eng.allgather_list(comm_of_one, 0, 1, rec, cap, segments, out2, n - 5, L.VISIBLE_HEADER, 12)
    torch.cuda.synchronize()
    assert int(out2[:4].view(torch.int32).item()) == n - 5
    assert torch.equal(out2[8:8 + 12 * (n - 5)], rec[8:8 + 12 * (n - 5)]) and bool((out2[8 + 12 * (n - 5):] == 0xCD).all())
    with pytest.raises(OrbitError) as ei:
        eng.status()
    assert ei.value.code == E_CAPACITY
    eng.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_compact_segments_is_the_rank_ordered_concatenation(world):
    """orbit_compact_segments on `world` segments as an all-gather would have left them (every rank's list buffer whole,
    counts inside, slack behind the items): the contiguous list is the concatenation in rank order — what
    orbit_exchange_list delivers — for ragged counts, an empty rank and a full one; eager and as a replayed graph."""
    import torch

    from orbit_amd.engine import Engine

    eng = Engine(0)
    rng = np.random.default_rng(world)
    cap = 5000
    counts = [int(c) for c in rng.integers(0, cap, world)]
    counts[0], counts[-1] = cap, 0  # a full segment and an empty one
    seg_bytes = L.VISIBLE_HEADER + 12 * cap
    host = rng.integers(0, 256, world * seg_bytes, dtype=np.uint8)  # slack = garbage, as in a reused buffer
    for r in range(world):
        host[seg_bytes * r:seg_bytes * r + 8].view(np.uint32)[:] = (counts[r], 0xDEAD)
    want = np.concatenate([host[seg_bytes * r + 8:seg_bytes * r + 8 + 12 * counts[r]] for r in range(world)])
    total = sum(counts)
    segments = torch.from_numpy(host).cuda()
    out = torch.full((L.VISIBLE_HEADER + 12 * total + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    eng.compact_segments(segments, world, cap, out, total, L.VISIBLE_HEADER, 12)
    torch.cuda.synchronize()
    eng.status()
    got = out.cpu().numpy()
    assert list(got[:8].view(np.uint32)) == [total, 0]
    assert np.array_equal(got[8:8 + 12 * total], want) and bool((got[8 + 12 * total:] == 0xCD).all())
    # captured once, replayed on other counts: the counts are read on the device
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        eng.compact_segments(segments, world, cap, out, total, L.VISIBLE_HEADER, 12, stream=st)  # warm
        st.synchronize()
        with torch.cuda.graph(g, stream=st):
            eng.compact_segments(segments, world, cap, out, total, L.VISIBLE_HEADER, 12, stream=st)
    counts2 = [c // 2 for c in counts]
    for r in range(world):
        host[seg_bytes * r:seg_bytes * r + 4].view(np.uint32)[0] = counts2[r]
    segments.copy_(torch.from_numpy(host))
    out.fill_(0xCD)
    g.replay()
    torch.cuda.synchronize()
    eng.status()
    want2 = np.concatenate([host[seg_bytes * r + 8:seg_bytes * r + 8 + 12 * counts2[r]] for r in range(world)])
    got = out.cpu().numpy()
    assert int(got[:4].view(np.uint32)[0]) == sum(counts2) and np.array_equal(got[8:8 + 12 * sum(counts2)], want2)
    assert bool((got[8 + 12 * sum(counts2):] == 0xCD).all())
    eng.close()
