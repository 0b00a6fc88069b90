"""The exchange without a host round trip (orbit_p2p_* + orbit_scatter_list, dist.P2PExchange) with its REAL
transport: two / three processes that share the test box's one GPU map each other's exchange buffers through HIP IPC
and scatter their shard's list into every peer's buffer with the HIP kernel.  The fixed-size collectives run over gloo
here (RCCL refuses two ranks on one device); on a multi-GPU node they are RCCL's and nothing waits for the host."""
import os
import socket

import numpy as np
import pytest

import scenes as sc
from orbit_amd import layouts as L

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, seed, n_entities, out_dir, signalling="collectives"):
    import torch
    import torch.distributed as dist

    from orbit_amd.dist import P2PExchange, shard_ranges
    from orbit_amd.engine import Engine

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = sc.make_scene(seed, n_entities, n_meshes=60)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, cam.planes)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    eng = Engine(0)
    b, e = shard_ranges(scene.entity_draw_count, world)[rank]
    disp_cap, cap = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
    draw = torch.zeros(L.DRAW_HEADER + 28 * cap, dtype=torch.uint8, device=dev)
    vis = torch.zeros(L.VISIBLE_HEADER + 8 * cap, dtype=torch.uint8, device=dev)
    ents, mats, mlt = up(scene.entities), up(scene.materials), up(scene.meshlets)
    edb, mi = up(scene.entity_draw_buffer()), up(scene.mesh_infos)
    eng.entity_cull(ci, edb, mi, disp, ents, e - b, disp_cap, draw_first=b)
    eng.meshlet_cull(ci, disp, mlt, draw, ents, mats, disp_cap, cap)
    eng.meshlet_cull_visible(ci, disp, mlt, vis, ents, mats, disp_cap, cap)
    ex_c = P2PExchange(eng, cap, L.DRAW_HEADER, 28, dev, signalling=signalling)
    ex_v = P2PExchange(eng, cap, L.VISIBLE_HEADER, 8, dev, signalling=signalling)
    assert ex_c.signalling == signalling
    expanded = torch.zeros(L.DRAW_HEADER + 28 * cap, dtype=torch.uint8, device=dev)
    steps = 3 if signalling == "collectives" else 12
    for step in range(steps):  # buffers are reused step after step
        out_c = ex_c.exchange(draw)
        out_v = ex_v.exchange(vis)
        eng.expand_visible(out_v, mlt, expanded, cap)
    torch.cuda.synchronize()
    if signalling == "device":
        # the same step — cull of the shard, both exchanges, the expansion — captured into ONE graph and replayed: no
        # collective and no host call inside (the epochs live in the control blocks)
        expanded.zero_()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                eng.entity_cull(ci, edb, mi, disp, ents, e - b, disp_cap, draw_first=b, stream=s)
                eng.meshlet_cull(ci, disp, mlt, draw, ents, mats, disp_cap, cap, stream=s)
                eng.meshlet_cull_visible(ci, disp, mlt, vis, ents, mats, disp_cap, cap, stream=s)
                ex_c.exchange(draw, stream=s)
                ex_v.exchange(vis, stream=s)
                eng.expand_visible(ex_v.out, mlt, expanded, cap, stream=s)
        dist.barrier()  # every rank has captured (capture itself enqueues nothing)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
    eng.status()
    dist.barrier()
    np.save(os.path.join(out_dir, f"cmd_{rank}.npy"), out_c.cpu().numpy())
    np.save(os.path.join(out_dir, f"exp_{rank}.npy"), expanded.cpu().numpy())
    dist.barrier()
    ex_c.close()
    ex_v.close()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("signalling", ["collectives", "device"])
@pytest.mark.parametrize("world,n_entities", [(2, 700), (3, 333)])
def test_p2p_exchange_processes_sharing_one_gpu(tmp_path, oracle, world, n_entities, signalling):
    """signalling = "device": orbit_exchange_list — counts and completion are stores into the peers' control blocks
    awaited on the device; the step has no collective in it and is also replayed from a captured graph."""
    import torch.multiprocessing as mp

    seed = 35
    mp.spawn(_worker, args=(world, _free_port(), seed, n_entities, str(tmp_path), signalling), nprocs=world, join=True)
    scene = sc.make_scene(seed, n_entities, n_meshes=60)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, cam.planes)
    disp, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                    scene.entities, scene.max_dispatches() + 8)
    ref, _, _ = oracle.meshlet_cull(ci, disp, scene.meshlets, scene.lod0_meshlets + 8, scene.entities, scene.materials)
    n = int(ref[:4].view(np.uint32)[0])
    assert n > 100
    for r in range(world):
        for name in ("cmd", "exp"):  # 28-B exchange, and 8-B exchange + expansion
            got = np.load(tmp_path / f"{name}_{r}.npy")
            assert int(got[:4].view(np.uint32)[0]) == n, (name, r)
            assert np.array_equal(got[:4 + 28 * n], ref[:4 + 28 * n]), f"rank {r} {name}: != the single-GPU list"
