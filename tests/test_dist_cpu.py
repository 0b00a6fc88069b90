"""world_size-2 gloo test of the rank-ordered visible-list gather (orbit_amd/dist.py): the
concatenation of per-shard oracle outputs over contiguous 32-aligned entity ranges must be
the single-process canonical sequence, and the gather must reproduce it on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import scenes as sc
from orbit_amd import layouts as L
from orbit_amd.dist import gather_visible, shard_ranges, warm_up


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, seed, n_entities, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle

    scene = sc.make_scene(seed, n_entities, n_meshes=40)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, cam.planes)
    b, e = shard_ranges(scene.entity_draw_count, world)[rank]
    cap = scene.max_dispatches() + 8
    disp, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), e - b, scene.mesh_infos, scene.entities, cap,
                                    draw_first=b)
    draw, _, _ = oracle.meshlet_cull(ci, disp, scene.meshlets, scene.lod0_meshlets + 8, scene.entities,
                                     scene.materials)
    local = torch.from_numpy(draw.copy())
    out = torch.zeros(L.DRAW_HEADER + 28 * (scene.lod0_meshlets + 8), dtype=torch.uint8)
    assert warm_up(torch.device("cpu"))
    counts = gather_visible(local, out)
    assert counts[rank] == int(draw[:4].view(np.uint32)[0])
    np.save(os.path.join(out_dir, f"gathered_{rank}.npy"), out.numpy())
    # the compact exchange format (8 B per survivor after an 8-B header) through the same collective pattern
    n, cmds = L.draw_buffer_commands(draw)
    vis = np.zeros(L.VISIBLE_HEADER + 8 * n, dtype=np.uint8)
    vis[:4].view(np.uint32)[0] = n
    v = vis[8:].view(L.VISIBLE_MESHLET)
    v["meshlet_index"], v["entity_index"] = cmds["meshlet_index"], cmds["cmd_first_instance"]
    out_v = torch.full((L.VISIBLE_HEADER + 8 * (scene.lod0_meshlets + 8),), 0x77, dtype=torch.uint8)
    counts_v = gather_visible(torch.from_numpy(vis), out_v, header=L.VISIBLE_HEADER, stride=8)
    assert counts_v == counts
    np.save(os.path.join(out_dir, f"gathered_vis_{rank}.npy"), out_v.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_entities", [257, 64])
def test_gather_visible_gloo_world2(tmp_path, oracle, n_entities):
    world, seed = 2, 31
    port = _free_port()
    mp.spawn(_worker, args=(world, port, seed, n_entities, str(tmp_path)), nprocs=world, join=True)
    scene = sc.make_scene(seed, n_entities, n_meshes=40)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, cam.planes)
    disp, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                    scene.entities, scene.max_dispatches() + 8)
    ref, _, _ = oracle.meshlet_cull(ci, disp, scene.meshlets, scene.lod0_meshlets + 8, scene.entities,
                                    scene.materials)
    n = int(ref[:4].view(np.uint32)[0])
    assert n > 0
    for r in range(world):
        got = np.load(tmp_path / f"gathered_{r}.npy")
        assert int(got[:4].view(np.uint32)[0]) == n
        assert np.array_equal(got[:4 + 28 * n], ref[:4 + 28 * n]), f"rank {r}: gathered list != single-process list"
        gv = np.load(tmp_path / f"gathered_vis_{r}.npy")
        _, cmds = L.draw_buffer_commands(ref)
        assert [int(v) for v in gv[:8].view(np.uint32)] == [n, 0]
        items = gv[8:8 + 8 * n].view(L.VISIBLE_MESHLET)
        assert np.array_equal(items["meshlet_index"], cmds["meshlet_index"])
        assert np.array_equal(items["entity_index"], cmds["cmd_first_instance"])
        assert np.all(gv[8 + 8 * n:] == 0x77)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_launches_its_own_ranks(world):
    """`python bench.py --gpus N` outside torch.distributed.run starts N fresh rank processes itself (before anything
    touches a GPU), forwards rank 0's ONE JSON line and fails if a rank fails.  ORBIT_BENCH_DRY_RANKS=1 stops the ranks
    after the gloo rendezvous and the exchange of their shard ranges (no device here)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["ORBIT_BENCH_DRY_RANKS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 2
    assert [tuple(s) for s in out["shards"]] == shard_ranges(195_313, world)
    # a world that does not match --gpus is an error, not something silently ignored
    env2 = dict(env, WORLD_SIZE=str(world + 1), RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world)], env=env2,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r2.returncode != 0 and b"WORLD_SIZE" in r2.stderr


# ----------------------------------------------------------------------------- exchange without a host round trip
class _ShmTransport:
    """CPU stand-in for the HIP transport of dist.P2PExchange: exchange buffers are files in /dev/shm mapped by every
    rank (the IPC mapping), the scatter is orbit_scatter_list restated with numpy (same offsets, same header rule)."""

    def __init__(self, tag):
        self.tag, self.paths = tag, []

    def alloc(self, nbytes):
        path = f"/dev/shm/orbit_test_{self.tag}_{os.getpid()}"
        m = np.lib.format.open_memmap(path, mode="w+", dtype=np.uint8, shape=(nbytes,))
        m[:] = 0x55
        self.paths.append(path)
        return torch.from_numpy(m), path

    def open(self, handle, nbytes):
        return torch.from_numpy(np.load(handle, mmap_mode="r+"))

    def scatter(self, local_list, counts, rank, world, peers, capacity, header, stride, stream):
        c = [int(v) for v in counts.tolist()]
        first, total = sum(c[:rank]), sum(c)
        n = min(c[rank], max(capacity - first, 0))
        for r in range(world):
            peers[r][header + stride * first:header + stride * (first + n)] = local_list[header:header + stride * n]
        peers[rank][:header] = 0
        peers[rank][:4].view(torch.int32)[0] = min(total, capacity)

    def close(self):
        for p in self.paths:
            if os.path.exists(p):
                os.remove(p)


def _visible_records(disp, cmds):
    """{records, survivors} + 12-B {entity_index, meshlet_offset, mask} of every dispatch record (mask 0: no survivor):
    what orbit_meshlet_cull_visible_records makes of the same cull (the commands come in (record, lane) order)."""
    _, recs = L.dispatch_buffer_records(disp)
    out, j = [], 0
    for r in recs:
        ent, off, cnt = int(r["entity_index"]), int(r["meshlet_offset"]), int(r["meshlet_count"])
        mask = 0
        while j < len(cmds) and int(cmds["cmd_first_instance"][j]) == ent and off <= int(cmds["meshlet_index"][j]) < off + cnt:
            mask |= 1 << (int(cmds["meshlet_index"][j]) - off)
            j += 1
        out.append((ent, off, mask))
    assert j == len(cmds)
    arr = np.array(out, dtype=L.VISIBLE_RECORD) if out else np.zeros(0, dtype=L.VISIBLE_RECORD)
    buf = np.zeros(L.VISIBLE_HEADER + 12 * max(len(arr), 1), dtype=np.uint8)
    buf[:8].view(np.uint32)[:] = (len(arr), len(cmds))
    buf[8:8 + 12 * len(arr)] = arr.view(np.uint8)
    return buf


def _compact_segments_numpy(segments, world, segment_capacity, out, out_capacity, header, stride, stream=None):
    """orbit_compact_segments restated with numpy (meshlet_lists.hip compact_segments_kernel): the segments' items in
    rank order behind {total, 0 ..}; a count beyond the segment's capacity is cut, nothing is written past out_capacity."""
    seg_bytes = header + stride * segment_capacity
    counts = [min(int(segments[seg_bytes * r:seg_bytes * r + 4].view(torch.int32)[0]), segment_capacity) for r in range(world)]
    first = 0
    for r in range(world):
        n = min(counts[r], max(out_capacity - first, 0))
        out[header + stride * first:header + stride * (first + n)] = \
            segments[seg_bytes * r + header:seg_bytes * r + header + stride * n]
        first += counts[r]
    out[:header] = 0
    out[:4].view(torch.int32)[0] = min(sum(counts), out_capacity)


def _p2p_worker(rank, world, port, seed, n_entities, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from orbit_amd.dist import AllGatherExchange, P2PExchange

    scene = sc.make_scene(seed, n_entities, n_meshes=40)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, cam.planes)
    b, e = shard_ranges(scene.entity_draw_count, world)[rank]
    cap = scene.max_dispatches() + 8
    disp, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), e - b, scene.mesh_infos, scene.entities, cap,
                                    draw_first=b)
    draw, _, _ = oracle.meshlet_cull(ci, disp, scene.meshlets, scene.lod0_meshlets + 8, scene.entities,
                                     scene.materials)
    n, cmds = L.draw_buffer_commands(draw)
    vis = np.zeros(L.VISIBLE_HEADER + 8 * max(n, 1), dtype=np.uint8)
    vis[:4].view(np.uint32)[0] = n
    v = vis[8:8 + 8 * n].view(L.VISIBLE_MESHLET)
    v["meshlet_index"], v["entity_index"] = cmds["meshlet_index"], cmds["cmd_first_instance"]
    rec = _visible_records(disp, cmds)  # the same list at record granularity (12 B per dispatch record)
    capacity = scene.lod0_meshlets + 8
    # the self-check a launcher runs before it relies on an exchange (bench.py): a known list per rank, compared
    pr = P2PExchange(None, 64, L.VISIBLE_HEADER, 12, torch.device("cpu"), transport=_ShmTransport(f"probe{rank}"))
    assert pr.probe()
    pr.close()
    for name, local, header, stride in (("cmd", draw, L.DRAW_HEADER, 28), ("vis", vis, L.VISIBLE_HEADER, 8),
                                        ("rec", rec, L.VISIBLE_HEADER, 12)):
        t = _ShmTransport(f"{name}{rank}")
        ex = P2PExchange(None, capacity, header, stride, torch.device("cpu"), transport=t)
        for step in range(2):  # the buffers are reused step after step
            out = ex.exchange(torch.from_numpy(local.copy()))
            dist.barrier()
        np.save(os.path.join(out_dir, f"p2p_{name}_{rank}.npy"), out.numpy().copy())
        dist.barrier()
        ex.close()
    # north_star's transport: the record list as ONE all-gather of fixed-capacity segments + the compaction — the same
    # capacity on every rank (the largest shard's records), this rank's list in a buffer of a whole segment
    seg_cap = max(e_ - b_ for b_, e_ in shard_ranges(scene.entity_draw_count, world)) * 3 + 8
    n_rec = int(rec[:4].view(np.uint32)[0])
    assert n_rec <= seg_cap
    local = np.full(L.VISIBLE_HEADER + 12 * seg_cap, 0x33, dtype=np.uint8)  # the slack behind the list travels too
    local[:8 + 12 * n_rec] = rec[:8 + 12 * n_rec]
    ag = AllGatherExchange(None, seg_cap, capacity, L.VISIBLE_HEADER, 12, torch.device("cpu"), compact=_compact_segments_numpy)
    assert ag.probe()
    ag.out[:] = 0x55
    for step in range(2):
        out = ag.exchange(torch.from_numpy(local.copy()))
        dist.barrier()
    np.save(os.path.join(out_dir, f"allgather_rec_{rank}.npy"), out.numpy().copy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_entities", [(2, 257), (3, 300), (3, 40)])
def test_p2p_exchange_protocol_gloo(tmp_path, oracle, world, n_entities):
    """dist.P2PExchange (counts all-gather -> device-side offsets -> every rank writes its list into every peer's
    buffer -> completion collective), both list formats, worlds of 2 and 3 (one shard may be empty): every rank ends
    with the single-process canonical list."""
    seed = 33
    port = _free_port()
    mp.spawn(_p2p_worker, args=(world, port, seed, n_entities, str(tmp_path)), nprocs=world, join=True)
    scene = sc.make_scene(seed, n_entities, n_meshes=40)
    cam = sc.default_camera()
    ci = sc.make_cull_info(cam.view, cam.planes)
    disp, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                    scene.entities, scene.max_dispatches() + 8)
    ref, _, _ = oracle.meshlet_cull(ci, disp, scene.meshlets, scene.lod0_meshlets + 8, scene.entities,
                                    scene.materials)
    n, cmds = L.draw_buffer_commands(ref)
    assert n > 0
    for r in range(world):
        got = np.load(tmp_path / f"p2p_cmd_{r}.npy")
        assert int(got[:4].view(np.uint32)[0]) == n
        assert np.array_equal(got[:4 + 28 * n], ref[:4 + 28 * n]), f"rank {r}: exchanged list != single-process list"
        assert np.all(got[4 + 28 * n:] == 0x55)  # nothing written past the list
        gv = np.load(tmp_path / f"p2p_vis_{r}.npy")
        assert [int(v) for v in gv[:8].view(np.uint32)] == [n, 0]
        items = gv[8:8 + 8 * n].view(L.VISIBLE_MESHLET)
        assert np.array_equal(items["meshlet_index"], cmds["meshlet_index"])
        assert np.array_equal(items["entity_index"], cmds["cmd_first_instance"])
        gr = np.load(tmp_path / f"p2p_rec_{r}.npy")  # the record-granular list: the single-process one, in rank order
        want = _visible_records(disp, cmds)
        nr = int(want[:4].view(np.uint32)[0])
        assert int(gr[:4].view(np.uint32)[0]) == nr and np.array_equal(gr[8:8 + 12 * nr], want[8:8 + 12 * nr])
        assert np.all(gr[8 + 12 * nr:] == 0x55)
        # ... and the same list through the all-gather of fixed-capacity segments: what the p2p exchange delivered, byte
        # for byte (header included), and nothing written behind it
        ga = np.load(tmp_path / f"allgather_rec_{r}.npy")
        assert np.array_equal(ga[:8 + 12 * nr], gr[:8 + 12 * nr]) and np.all(ga[8 + 12 * nr:] == 0x55)
