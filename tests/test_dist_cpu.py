"""Multi-process gloo tests of the sharded engine's exchange (orbit_amd/dist.py), worlds of 2, 3, 5 and 8: the
concatenation of per-shard oracle outputs over contiguous 32-aligned entity ranges must be the single-process canonical
sequence, and both transports — the device-signalled p2p exchange (its control-block protocol restated over shared
memory) and the all-gather of fixed-capacity segments — must reproduce it on every rank."""
import os
import socket
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import scenes as sc
from orbit_amd import layouts as L
from orbit_amd.dist import shard_ranges


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 3, 8])
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


# ----------------------------------------------------------------------------- the exchange, no host in the step
class _ShmTransport:
    """CPU stand-in for the HIP transport of dist.P2PExchange: exchange buffers and control blocks are files in /dev/shm
    mapped by every rank (the IPC mapping), and `exchange` restates orbit_exchange_list's two launches
    (orbit_amd/csrc/meshlet_lists.hip: exchange_scatter_kernel, exchange_wait_kernel) with numpy — the same control-block
    layout, the same epochs and count-slot parity, the same bounded waits — run by real processes polling real shared
    memory.  ExchangeCtrl: u64 count[2][16] @0 (items | epoch << 32), u64 done[16] @256, u32 epoch @384."""

    MAX_WORLD, TIMEOUT_S = 16, 60.0

    def __init__(self, tag):
        self.tag, self.paths, self.timeouts = tag, [], 0

    def alloc(self, nbytes):
        path = f"/dev/shm/orbit_test_{self.tag}_{os.getpid()}_{len(self.paths)}"
        m = np.lib.format.open_memmap(path, mode="w+", dtype=np.uint8, shape=(nbytes,))
        m[:] = 0  # orbit_p2p_alloc zero-fills
        self.paths.append(path)
        return torch.from_numpy(m), path

    def open(self, handle, nbytes):
        return torch.from_numpy(np.load(handle, mmap_mode="r+"))

    @staticmethod
    def _count(ctrl):
        return ctrl.numpy()[:256].view(np.uint64).reshape(2, 16)

    @staticmethod
    def _done(ctrl):
        return ctrl.numpy()[256:384].view(np.uint64)

    def _wait(self, cond):
        t0 = time.time()
        while not cond():
            if time.time() - t0 > self.TIMEOUT_S:
                self.timeouts += 1  # ORBIT_E_TIMEOUT
                return False
            time.sleep(0.0005)
        return True

    def exchange(self, local_list, rank, world, peers, ctrls, capacity, header, stride, stream):
        mine = ctrls[rank]
        epoch = mine.numpy()[384:388].view(np.uint32)
        e = int(epoch[0]) + 1
        n_mine = int(local_list[:4].view(torch.int32)[0])
        # scatter launch: this rank's count into every rank's block ...
        for r in range(world):
            self._count(ctrls[r])[e & 1, rank] = np.uint64((e << 32) | n_mine)
        # ... bounded wait for EVERY rank's count of this epoch (those of the lower ranks fix the offset)
        slot = self._count(mine)[e & 1]
        ok = self._wait(lambda: all(int(slot[r]) >> 32 == e for r in range(world)))
        first = sum(int(slot[r]) & 0xFFFFFFFF for r in range(rank))
        n = min(n_mine, max(capacity - first, 0)) if ok else 0  # a count that never came: nothing is copied anywhere
        for r in range(world):
            peers[r][header + stride * first:header + stride * (first + n)] = local_list[header:header + stride * n]
        for r in range(world):  # ... and done, at every destination
            self._done(ctrls[r])[rank] = np.uint64(e)
        # wait launch: every rank's count and done for this epoch, then the header and the epoch
        done = self._done(mine)
        ok = self._wait(lambda: all(int(slot[r]) >> 32 == e and int(done[r]) == e for r in range(world)))
        total = sum(int(slot[r]) & 0xFFFFFFFF for r in range(world)) if ok else 0
        peers[rank][:header] = 0
        peers[rank][:4].view(torch.int32)[0] = min(total, capacity)
        epoch[0] = e

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
    rec = _visible_records(disp, cmds)  # the shard's list at record granularity (12 B per dispatch record)
    capacity = scene.lod0_meshlets + 8
    # the probe the constructor runs (a known list per rank, compared on every rank) is part of every object below
    for name, local, header, stride in (("cmd", draw, L.DRAW_HEADER, 28), ("rec", rec, L.VISIBLE_HEADER, 12)):
        t = _ShmTransport(f"{name}{rank}")
        ex = P2PExchange(None, capacity, header, stride, torch.device("cpu"), transport=t)
        assert ex.fallback is None and ex.signalling == "device"
        ex.out[:] = 0x55
        dist.barrier()
        for step in range(4):  # the buffers are reused step after step: epochs 2 .. 5 behind the probe's, both count slots
            out = ex.exchange(torch.from_numpy(local.copy()))
        assert t.timeouts == 0
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


@pytest.mark.parametrize("world,n_entities", [(2, 257), (3, 300), (3, 40), (5, 333), (8, 200), (8, 1000)])
def test_p2p_exchange_protocol_gloo(tmp_path, oracle, world, n_entities):
    """dist.P2PExchange — orbit_exchange_list's protocol (count out to every control block; wait for all counts; copy
    to the rank-ordered position in every peer's buffer; done; wait for all counts and dones; header; epoch) restated
    over shared memory — with 28-B commands and 12-B records, and dist.AllGatherExchange, worlds of 2, 3, 5 and 8
    (shards may be empty: 200 entity-draws over 8 ranks, 40 over 3): every rank ends with the single-process list."""
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
        gr = np.load(tmp_path / f"p2p_rec_{r}.npy")  # the record-granular list: the single-process one, in rank order
        want = _visible_records(disp, cmds)
        nr = int(want[:4].view(np.uint32)[0])
        assert int(gr[:4].view(np.uint32)[0]) == nr and np.array_equal(gr[8:8 + 12 * nr], want[8:8 + 12 * nr])
        assert np.all(gr[8 + 12 * nr:] == 0x55)
        # ... and the same list through the all-gather of fixed-capacity segments: what the p2p exchange delivered, byte
        # for byte (header included), and nothing written behind it
        ga = np.load(tmp_path / f"allgather_rec_{r}.npy")
        assert np.array_equal(ga[:8 + 12 * nr], gr[:8 + 12 * nr]) and np.all(ga[8 + 12 * nr:] == 0x55)
