"""The exchange with no host in the step (orbit_p2p_* + orbit_exchange_list, dist.P2PExchange) with its REAL transport:
2, 3, 5 and 8 processes that share the test box's one GPU map each other's exchange buffers and control blocks through
HIP IPC and scatter their shard's list into every peer's buffer with the HIP kernels — counts and completion signalled
on the device.  torch.distributed (gloo: RCCL refuses several ranks on one device) only carries the set-up and the
test's own barriers; the all-gather transport's compaction is fed by the ranks' real segments."""
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


def _worker(rank, world, port, seed, n_entities, out_dir):
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
    ents, mats, mlt = up(scene.entities), up(scene.materials), up(scene.meshlets)
    edb, mi = up(scene.entity_draw_buffer()), up(scene.mesh_infos)
    # the sharded engine's product: ONE call for the shard — the 12-B record list (one entry per dispatch record, written
    # by the evaluation), in a buffer of a whole all-gather segment (the same capacity on every rank), and the shard's own
    # 28-B commands
    seg_cap = max(e_ - b_ for b_, e_ in shard_ranges(scene.entity_draw_count, world)) * 4 + 8
    rec = torch.full((L.VISIBLE_HEADER + 12 * seg_cap,), 0x33, dtype=torch.uint8, device=dev)

    def cull(stream=None):
        eng.cull_shard(ci, edb, mi, disp, ents, b, e - b, disp_cap, mlt, mats, rec, seg_cap, draw_commands_buffer=draw,
                       draw_capacity=cap, stream=stream)

    cull()
    ex_c = P2PExchange(eng, cap, L.DRAW_HEADER, 28, dev)          # any header / stride: the 28-B commands themselves
    ex_r = P2PExchange(eng, seg_cap * world, L.VISIBLE_HEADER, 12, dev)  # the record lists
    assert ex_c.fallback is None and ex_r.fallback is None and ex_r.signalling == "device"
    expanded = torch.zeros(L.DRAW_HEADER + 28 * cap, dtype=torch.uint8, device=dev)
    for step in range(12):  # buffers are reused step after step: twelve epochs on every control block
        out_c = ex_c.exchange(draw)
        out_r = ex_r.exchange(rec)
        eng.expand_visible_records(out_r, mlt, expanded, cap)
    torch.cuda.synchronize()
    eng.status()
    # north_star's transport fed by `world` REAL segments: every rank's whole list buffer {count | header | seg_cap
    # entries, slack included} gathered (over gloo here), compacted on the device by orbit_compact_segments — byte for
    # byte what the p2p exchange delivered
    segs = [torch.zeros(rec.numel(), dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(segs, rec.cpu())
    segments = torch.cat(segs).to(dev)
    compacted = torch.full((L.VISIBLE_HEADER + 12 * seg_cap * world,), 0x55, dtype=torch.uint8, device=dev)
    eng.compact_segments(segments, world, seg_cap, compacted, seg_cap * world, L.VISIBLE_HEADER, 12)
    torch.cuda.synchronize()
    n_all = int(out_r[:4].view(torch.int32).item())
    assert torch.equal(compacted[:L.VISIBLE_HEADER + 12 * n_all], out_r[:L.VISIBLE_HEADER + 12 * n_all]), \
        "compact_segments of the ranks' real segments != the p2p exchange's list"
    assert bool((compacted[L.VISIBLE_HEADER + 12 * n_all:] == 0x55).all()), "compact_segments wrote past the list"
    # the same step — cull of the shard, both exchanges, the expansion — captured into ONE graph and replayed: no
    # collective and no host call inside (the epochs live in the control blocks)
    expanded.zero_()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            cull(stream=s)
            ex_c.exchange(draw, stream=s)
            ex_r.exchange(rec, stream=s)
            eng.expand_visible_records(ex_r.out, mlt, expanded, cap, stream=s)
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
    ex_r.close()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_entities", [(2, 700), (3, 333), (5, 500), (8, 900), (8, 200)])
def test_p2p_exchange_processes_sharing_one_gpu(tmp_path, oracle, world, n_entities):
    """orbit_exchange_list — counts and completion are stores into the peers' control blocks awaited on the device; the
    step has no collective in it and is also replayed from a captured graph.  Worlds up to the 8 north_star names (and 5:
    not a power of two; 200 entity-draws over 8 ranks: an empty shard): eight writers per control block, seven
    destinations per scatter, twelve eager epochs + five replays on top of the constructor's probe."""
    import torch.multiprocessing as mp

    seed = 35
    mp.spawn(_worker, args=(world, _free_port(), seed, n_entities, str(tmp_path)), nprocs=world, join=True)
    scene = sc.make_scene(seed, n_entities, n_meshes=60)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, cam.planes)
    disp, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                    scene.entities, scene.max_dispatches() + 8)
    ref, _, _ = oracle.meshlet_cull(ci, disp, scene.meshlets, scene.lod0_meshlets + 8, scene.entities, scene.materials)
    n = int(ref[:4].view(np.uint32)[0])
    assert n > 100
    for r in range(world):
        for name in ("cmd", "exp"):  # the 28-B exchange, and the 12-B record exchange + expansion
            got = np.load(tmp_path / f"{name}_{r}.npy")
            assert int(got[:4].view(np.uint32)[0]) == n, (name, r)
            assert np.array_equal(got[:4 + 28 * n], ref[:4 + 28 * n]), f"rank {r} {name}: != the single-GPU list"
