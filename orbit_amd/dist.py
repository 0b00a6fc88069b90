"""Multi-GPU sharding of the cull path: one process per GPU,
contiguous 32-aligned entity-draw ranges per rank, and a rank-ordered gather of
the compacted visible list over RCCL (torch.distributed backend "nccl" on ROCm;
"gloo" in the CPU tests).

The reference is single-GPU (no collective anywhere); the exchange step exists
only because the sharded engine must hand one canonical MeshletDrawCommandBuffer
to the renderer.  Because every rank's survivor list is already in canonical
(record, lane) order and the shards are contiguous in entity order, concatenating
the per-rank lists in rank order IS the single-GPU sequence.

Pattern: all_gather of the 4-byte counts, then a grouped send/recv in which each
rank pushes its variable-sized list straight into `draws + prefix[rank]` of every
peer — on a fully connected xGMI node that drives all 7 links of a GPU at once,
which a ring all-gather (per-link bound) would not.
"""
import torch
import torch.distributed as dist

from . import layouts as L


def shard_ranges(entity_draw_count, world):
    """[begin, end) per rank, cut at multiples of 32 so entity-bitset words are rank-private.
    Mirrors orbit_shard_range() of the C ABI (pure integer math, no device needed)."""
    words = (entity_draw_count + 31) // 32
    out = []
    for r in range(world):
        b = min(words * r // world * 32, entity_draw_count)
        e = min(words * (r + 1) // world * 32, entity_draw_count)
        out.append((b, e))
    return out


def warm_up(device, group=None):
    """Creates the collective and the point-to-point channels of gather_visible (RCCL sets them up lazily on
    first use, which costs far more than a frame): one count all-gather and a 4-byte exchange with every peer."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = torch.full((1,), rank, dtype=torch.int64, device=device)
    counts = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    send = torch.full((4,), rank, dtype=torch.uint8, device=device)
    recv = torch.zeros(world, 4, dtype=torch.uint8, device=device)
    ops = []
    for peer in range(world):
        if peer != rank:
            ops.append(dist.P2POp(dist.isend, send, peer, group))
            ops.append(dist.P2POp(dist.irecv, recv[peer], peer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    return [int(c) for c in counts.cpu().tolist()] == list(range(world))


def gather_visible(local_draw_buffer, out_draw_buffer, group=None, header=L.DRAW_HEADER, stride=28):
    """Rank-ordered all-gather of MeshletDrawCommandBuffers (header 4, stride 28: the defaults) or of compact
    visible lists (header=L.VISIBLE_HEADER, stride=8: orbit_meshlet_cull_visible / orbit_expand_visible).

    local_draw_buffer : uint8 tensor, this rank's buffer (u32 count @0, 28-B commands @4)
    out_draw_buffer   : uint8 tensor large enough for the concatenation; receives
                        count @0 and all ranks' commands @4 in rank order.
    Returns the per-rank counts (python list).  One host sync (the counts) is
    inherent: RCCL needs message sizes on the host.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = local_draw_buffer[:4].view(torch.int32).to(torch.int64)
    counts_t = torch.empty(world, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(counts_t, mine, group=group)
    counts = [int(c) for c in counts_t.cpu().tolist()]
    total = sum(counts)
    need = header + stride * total
    if out_draw_buffer.numel() < need:
        raise ValueError(f"gather_visible: output holds {out_draw_buffer.numel()} B, needs {need} B")
    prefix = [0]
    for c in counts:
        prefix.append(prefix[-1] + c)
    out_draw_buffer[:4].view(torch.int32)[0] = total
    if header > 4:
        out_draw_buffer[4:header].zero_()
    my_bytes = local_draw_buffer[header:header + stride * counts[rank]]
    ops = []
    for peer in range(world):
        if peer == rank:
            continue
        if counts[rank] > 0:
            ops.append(dist.P2POp(dist.isend, my_bytes, peer, group))
        if counts[peer] > 0:
            dst = out_draw_buffer[header + stride * prefix[peer]:header + stride * prefix[peer + 1]]
            ops.append(dist.P2POp(dist.irecv, dst, peer, group))
    out_draw_buffer[header + stride * prefix[rank]:header + stride * prefix[rank + 1]].copy_(my_bytes)
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return counts


def device_view(ptr, nbytes, device):
    """uint8 torch view of `nbytes` of device memory at raw pointer `ptr` (an exchange buffer of the library)."""
    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 3}
    return torch.as_tensor(h, device=device)


class P2PExchange:
    """Rank-ordered all-gather of the ranks' lists WITHOUT a host round trip (include/orbit_abi.h,
    orbit_scatter_list): the counts are all-gathered into a device tensor (fixed size, nothing is read back), every
    rank copies its list straight into every peer's output buffer at out + header + stride * prefix[rank] — the
    offsets are computed on the device, the peers' buffers are mapped over xGMI (IPC) — and a second fixed-size
    collective is the completion barrier.  RCCL carries 2 x 4 B per rank and step; the lists travel as plain stores
    on all links at once.

    `signalling` = "device" (default): not even those two collectives — counts and completion are stores into the
    peers' control blocks, awaited on the device (orbit_exchange_list): a step is kernels only, nothing in it is issued
    through torch.distributed, and it can be captured into a graph.  "collectives": the form above.

    The device-signalled form polls words that a PEER DEVICE stores while the polling kernel runs.  Whether that works
    between the devices of a node is a property of the node, so the constructor proves it before anyone relies on it:
    with `probe_on_init` (the default) a known exchange is run and compared on every rank (`probe`), and if it fails on
    any rank ALL ranks move to the collectives form together — `signalling` says which form the object ended up with and
    `fallback` why.  A library user who constructs the object directly therefore never gets an unproven path.

    `transport` (tests): an object with the same `alloc / open / scatter` surface; the default drives the HIP library.
    """

    CTRL_BYTES = 1024  # ORBIT_P2P_CTRL_BYTES

    def __init__(self, engine, item_capacity, header, stride, device, group=None, transport=None, signalling="device",
                 probe_on_init=True):
        self.fallback = None
        self.signalling = signalling if (transport is None or hasattr(transport, "exchange")) else "collectives"
        self.group, self.header, self.stride, self.capacity = group, header, stride, int(item_capacity)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = device
        self.t = transport if transport is not None else _HipTransport(engine, device)
        nbytes = header + stride * self.capacity
        # every step of the set-up is agreed on by all ranks (a rank that fails alone would leave the others waiting
        # in the next collective): failure anywhere raises everywhere
        err = None
        try:
            self.out, handle = self.t.alloc(nbytes)  # uint8 tensor over this rank's output buffer
        except Exception as ex:  # noqa: BLE001 - reported below, on every rank
            err, handle = ex, None
        self._agree(err, "allocating / exporting the exchange buffer")
        handles = [None] * self.world
        dist.all_gather_object(handles, handle, group=group)
        try:
            self.peers = [self.out if r == self.rank else self.t.open(handles[r], nbytes) for r in range(self.world)]
        except Exception as ex:  # noqa: BLE001
            err = ex
        self._agree(err, "mapping a peer's exchange buffer")
        self.ctrls = None
        if self.signalling == "device":  # every rank's control block, mapped like the output buffers
            err = None
            try:
                self.ctrl, chandle = self.t.alloc(self.CTRL_BYTES)
            except Exception as ex:  # noqa: BLE001
                err, chandle = ex, None
            self._agree(err, "allocating / exporting the control block")
            chandles = [None] * self.world
            dist.all_gather_object(chandles, chandle, group=group)
            try:
                self.ctrls = [self.ctrl if r == self.rank else self.t.open(chandles[r], self.CTRL_BYTES)
                              for r in range(self.world)]
            except Exception as ex:  # noqa: BLE001
                err = ex
            self._agree(err, "mapping a peer's control block")
        self.counts = torch.zeros(self.world, dtype=torch.int32, device=device)
        self.token = torch.zeros(1, dtype=torch.int32, device=device)
        dist.barrier(group=group)  # every peer has mapped every buffer before the first scatter
        if self.signalling == "device" and probe_on_init:
            status = engine.status if engine is not None else None
            if not self.probe(check_status=status):  # agreed on by all ranks inside probe()
                self.signalling = "collectives"
                self.fallback = "device signalling -> collectives: the probe exchange failed on at least one rank"
                if status is not None:
                    try:
                        status()  # clear what the failed probe latched (a poll that ran out)
                    except Exception:  # noqa: BLE001
                        pass
                dist.barrier(group=group)

    def _agree(self, err, what):
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=self.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:
            raise RuntimeError(f"P2PExchange: {what} failed on at least one rank" + (f" (here: {err})" if err else ""))

    def exchange(self, local_list, stream=None):
        """Enqueues one exchange of `local_list` ({count | header | items}); afterwards (in stream order) `self.out`
        holds {total | header | all ranks' items in rank order}.  No host synchronisation."""
        if self.signalling == "device":  # two launches, no collective: counts and completion travel like the lists
            self.t.exchange(local_list, self.rank, self.world, self.peers, self.ctrls, self.capacity, self.header,
                            self.stride, stream)
            return self.out
        mine = local_list[:4].view(torch.int32)
        dist.all_gather_into_tensor(self.counts, mine, group=self.group)       # 4 B per rank, stays on the device
        self.t.scatter(local_list, self.counts, self.rank, self.world, self.peers, self.capacity, self.header,
                       self.stride, stream)
        dist.all_reduce(self.token, group=self.group)                          # completion: peers' scatters are done
        return self.out

    def probe(self, check_status=None):
        """One exchange of a known list per rank (rank r sends r + 1 items whose bytes are r + 1), read back and
        compared: True on every rank only if every rank received all ranks' items in rank order.  A launcher calls it
        before it relies on this exchange (bench.py: device signalling has to work between the devices of THIS node;
        if it does not, the step falls back to the collectives form and says so).  `check_status`: called after the
        synchronisation, raises if the device latched an error (a bounded poll that ran out)."""
        n = self.rank + 1
        if self.capacity < self.world * (self.world + 1) // 2:
            return True  # nothing this small can be sent through it
        local = torch.zeros(self.header + self.stride * self.world, dtype=torch.uint8, device=self.device)
        local[:4].view(torch.int32)[0] = n
        local[self.header:self.header + self.stride * n] = n
        ok = 1
        try:
            self.exchange(local, stream=None)
            if torch.device(self.device).type == "cuda":
                torch.cuda.synchronize(self.device)
            if check_status is not None:
                check_status()
            total = self.world * (self.world + 1) // 2
            got = self.out[:self.header + self.stride * total].clone()
            want = torch.cat([torch.full((self.stride * (r + 1),), r + 1, dtype=torch.uint8, device=self.device)
                              for r in range(self.world)])
            if int(got[:4].view(torch.int32)[0]) != total or not torch.equal(got[self.header:], want):
                ok = 0
        except Exception:  # noqa: BLE001 - the verdict is agreed on below
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return int(flag.item()) == 1

    def close(self):
        self.t.close()


class AllGatherExchange:
    """north_star's transport: the rank-ordered all-gather of the ranks' lists as ONE RCCL all-gather — of a FIXED size,
    so that nothing about it needs the host: every rank contributes its whole list buffer as a segment
    {count | header | segment_capacity items} (torch.distributed.all_gather_into_tensor = ncclAllGather on ROCm), and one
    launch (orbit_compact_segments) reads the counts on the device and compacts the segments into the contiguous list
    {total | header | items} that P2PExchange delivers, byte for byte.  No count read-back, no stream synchronisation,
    capturable.  What the fixed size costs is the segments' slack on the links; the record list (one entry per dispatch
    record of the shard) has next to none.  Same surface as P2PExchange (`exchange`, `out`, `probe`, `close`); it is
    what bench.py's `--transport rccl` runs and what the p2p transport falls back to.

    `compact` (tests): a callable with orbit_compact_segments' arguments; the default drives the HIP library."""

    signalling = "rccl-allgather"

    def __init__(self, engine, segment_capacity, item_capacity, header, stride, device, group=None, compact=None):
        self.group, self.header, self.stride = group, header, stride
        self.segment_capacity, self.capacity = int(segment_capacity), int(item_capacity)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = device
        self.seg_bytes = header + stride * self.segment_capacity
        self.segments = torch.zeros(self.world * self.seg_bytes, dtype=torch.uint8, device=device)
        self.out = torch.zeros(header + stride * self.capacity, dtype=torch.uint8, device=device)
        self.compact = compact if compact is not None else (
            lambda seg, world, cap, out, out_cap, hdr, st, stream: engine.compact_segments(seg, world, cap, out, out_cap,
                                                                                           hdr, st, stream=stream))

    def exchange(self, local_list, stream=None):
        """Enqueues one exchange of `local_list` ({count | header | items}, a buffer of at least one segment's bytes);
        afterwards (in stream order) `self.out` holds {total | header | all ranks' items in rank order}."""
        if local_list.numel() < self.seg_bytes:
            raise ValueError(f"AllGatherExchange: the list buffer holds {local_list.numel()} B, a segment is {self.seg_bytes} B")
        # issued in the caller's stream context: ProcessGroupNCCL orders the collective behind the current stream and the
        # current stream behind the collective, all on the device
        dist.all_gather_into_tensor(self.segments, local_list[:self.seg_bytes], group=self.group)
        self.compact(self.segments, self.world, self.segment_capacity, self.out, self.capacity, self.header, self.stride,
                     stream)
        return self.out

    def probe(self, check_status=None):
        """As P2PExchange.probe: rank r sends r + 1 known items; True on every rank only if every rank received all
        ranks' items in rank order."""
        n = self.rank + 1
        if self.segment_capacity < self.world or self.capacity < self.world * (self.world + 1) // 2:
            return True
        local = torch.zeros(self.seg_bytes, dtype=torch.uint8, device=self.device)
        local[:4].view(torch.int32)[0] = n
        local[self.header:self.header + self.stride * n] = n
        ok = 1
        try:
            self.exchange(local, stream=None)
            if torch.device(self.device).type == "cuda":
                torch.cuda.synchronize(self.device)
            if check_status is not None:
                check_status()
            total = self.world * (self.world + 1) // 2
            got = self.out[:self.header + self.stride * total].clone()
            want = torch.cat([torch.full((self.stride * (r + 1),), r + 1, dtype=torch.uint8, device=self.device)
                              for r in range(self.world)])
            if int(got[:4].view(torch.int32)[0]) != total or not torch.equal(got[self.header:], want):
                ok = 0
        except Exception:  # noqa: BLE001 - the verdict is agreed on below
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return int(flag.item()) == 1

    def close(self):
        pass


class _HipTransport:
    def __init__(self, engine, device):
        self.eng, self.device, self.own, self.opened = engine, device, [], []
        self._calls = {}

    def alloc(self, nbytes):
        ptr, handle = self.eng.p2p_alloc(nbytes)
        self.own.append(ptr)
        return device_view(ptr, nbytes, self.device), handle

    def open(self, handle, nbytes):
        # a peer's buffer stays a raw address: it lives on ANOTHER device, and a torch tensor built over it with this
        # rank's device would be a local copy (torch looks the pointer's device up and converts), not a mapping
        ptr = self.eng.p2p_open(handle)
        self.opened.append(ptr)
        return ptr

    def scatter(self, local_list, counts, rank, world, peers, capacity, header, stride, stream):
        self.eng.scatter_list(local_list, counts, rank, world, peers, capacity, header, stride, stream=stream)

    def exchange(self, local_list, rank, world, peers, ctrls, capacity, header, stride, stream):
        # the argument blocks of a (list buffer, stream) pair are built once: a step loop calls this every step with the
        # same buffers, and filling two pointer arrays per call is ~15 us of host time in a 50-us step
        key = (local_list.data_ptr() if hasattr(local_list, "data_ptr") else int(local_list), id(stream))
        call = self._calls.get(key)
        if call is None:
            call = self._calls[key] = self.eng.prepare_exchange_list(local_list, rank, world, peers, ctrls, capacity, header,
                                                                    stride, stream=stream)
        call()

    def close(self):
        for ptr in self.opened:
            self.eng.p2p_close(ptr)
        for ptr in self.own:
            self.eng.p2p_free(ptr)
        self.own, self.opened = [], []
