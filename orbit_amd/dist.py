"""Multi-GPU sharding of the cull path: one process per GPU, contiguous 32-aligned entity-draw ranges
per rank, and the rank-ordered all-gather of the ranks' visible lists — 12-B records, one per dispatch record, written
by the shard's evaluation launch (orbit_cull_shard).

The reference is single-GPU (no collective anywhere); the exchange step exists only because the sharded engine must hand
one canonical visible list to the renderer.  Because every rank's list is already in canonical (record) order and the
shards are contiguous in entity order, concatenating the per-rank lists in rank order IS the single-GPU sequence.

Two transports, neither of which reads a count on the host:
  P2PExchange        every rank stores its list straight into every peer's IPC-mapped buffer at a device-computed offset;
                     counts and completion are stores into the peers' control blocks, awaited on the device
                     (orbit_exchange_list) — on a fully connected xGMI node that drives all 7 links of a GPU at once;
  AllGatherExchange  north_star's transport: ONE RCCL all-gather of fixed-capacity list segments + a device-side
                     compaction (orbit_compact_segments); what the p2p transport falls back to.
"""
import torch
import torch.distributed as dist


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


def device_view(ptr, nbytes, device):
    """uint8 torch view of `nbytes` of device memory at raw pointer `ptr` (an exchange buffer of the library)."""
    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 3}
    return torch.as_tensor(h, device=device)


class P2PExchange:
    """Rank-ordered all-gather of the ranks' lists with NO host in the step (include/orbit_abi_ext.h,
    orbit_exchange_list): every rank copies its list straight into every peer's output buffer at out + header + stride *
    prefix[rank] — the peers' buffers are mapped over xGMI (HIP IPC), the offsets come from counts that travel the same
    way, as stores into the peers' control blocks; completion likewise.  One call = two launches on `stream`; nothing in it
    is issued through torch.distributed, and it can be captured into a graph.

    The exchange polls words that a PEER DEVICE stores while the polling kernel runs.  Whether that works between the
    devices of a node is a property of the node, so the constructor proves it before anyone relies on it: with
    `probe_on_init` (the default) a known exchange is run and compared on every rank (`probe`).  If it fails on any rank,
    `fallback` says so on ALL ranks and `exchange` refuses: the caller moves to AllGatherExchange (bench.py does, all
    ranks together).  A library user who constructs the object directly therefore never gets an unproven path.

    `transport` (tests): an object with the same `alloc / open / exchange / close` surface; the default drives the HIP
    library."""

    CTRL_BYTES = 1024  # ORBIT_P2P_CTRL_BYTES
    signalling = "device"

    def __init__(self, engine, item_capacity, header, stride, device, group=None, transport=None, probe_on_init=True):
        self.fallback = None
        self.group, self.header, self.stride, self.capacity = group, header, stride, int(item_capacity)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = device
        self.t = transport if transport is not None else _HipTransport(engine, device)
        nbytes = header + stride * self.capacity
        # every step of the set-up is agreed on by all ranks (a rank that fails alone would leave the others waiting
        # in the next collective): failure anywhere raises everywhere
        self.out, self.peers = self._shared(nbytes, "the exchange buffer")
        self.ctrl, self.ctrls = self._shared(self.CTRL_BYTES, "the control block")  # mapped like the output buffers
        dist.barrier(group=group)  # every peer has mapped every buffer before the first scatter
        if probe_on_init:
            status = engine.status if engine is not None else None
            if not self.probe(check_status=status):  # agreed on by all ranks inside probe()
                self.fallback = "the device-signalled exchange failed its probe on at least one rank"
                if status is not None:
                    try:
                        status()  # clear what the failed probe latched (a poll that ran out)
                    except Exception:  # noqa: BLE001
                        pass
                dist.barrier(group=group)

    def _shared(self, nbytes, what):
        err = None
        try:
            own, handle = self.t.alloc(nbytes)  # uint8 tensor over this rank's buffer
        except Exception as ex:  # noqa: BLE001 - reported below, on every rank
            err, own, handle = ex, None, None
        self._agree(err, f"allocating / exporting {what}")
        handles = [None] * self.world
        dist.all_gather_object(handles, handle, group=self.group)
        mapped = None
        try:
            mapped = [own if r == self.rank else self.t.open(handles[r], nbytes) for r in range(self.world)]
        except Exception as ex:  # noqa: BLE001
            err = ex
        self._agree(err, f"mapping a peer's copy of {what}")
        return own, mapped

    def _agree(self, err, what):
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=self.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 0:
            raise RuntimeError(f"P2PExchange: {what} failed on at least one rank" + (f" (here: {err})" if err else ""))

    def exchange(self, local_list, stream=None):
        """Enqueues one exchange of `local_list` ({count | header | items}); afterwards (in stream order) `self.out`
        holds {total | header | all ranks' items in rank order}.  No host synchronisation, no collective."""
        if self.fallback is not None:
            raise RuntimeError(f"P2PExchange: {self.fallback}; use AllGatherExchange")
        self.t.exchange(local_list, self.rank, self.world, self.peers, self.ctrls, self.capacity, self.header, self.stride,
                        stream)
        return self.out

    def probe(self, check_status=None):
        """One exchange of a known list per rank (rank r sends r + 1 items whose bytes are r + 1), read back and
        compared: True on every rank only if every rank received all ranks' items in rank order.  `check_status`: called
        after the synchronisation, raises if the device latched an error (a bounded poll that ran out)."""
        return _probe(self, self.capacity, self.header + self.stride * self.world, check_status)

    def close(self):
        self.t.close()


def _probe(x, out_capacity, local_bytes, check_status):
    """Shared by both exchanges: rank r sends r + 1 known items; the verdict is agreed on by all ranks."""
    n = x.rank + 1
    total = x.world * (x.world + 1) // 2
    if out_capacity < total:
        return True  # nothing this small can be sent through it
    local = torch.zeros(local_bytes, dtype=torch.uint8, device=x.device)
    local[:4].view(torch.int32)[0] = n
    local[x.header:x.header + x.stride * n] = n
    ok = 1
    try:
        x.exchange(local, stream=None)
        if torch.device(x.device).type == "cuda":
            torch.cuda.synchronize(x.device)
        if check_status is not None:
            check_status()
        got = x.out[:x.header + x.stride * total].clone()
        want = torch.cat([torch.full((x.stride * (r + 1),), r + 1, dtype=torch.uint8, device=x.device)
                          for r in range(x.world)])
        if int(got[:4].view(torch.int32)[0]) != total or not torch.equal(got[x.header:], want):
            ok = 0
    except Exception:  # noqa: BLE001 - the verdict is agreed on below
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=x.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=x.group)
    return int(flag.item()) == 1


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
        # ProcessGroupNCCL orders the collective behind torch's CURRENT stream and the current stream behind the
        # collective, all on the device — so the collective is issued with `stream` current: the compaction launch below
        # goes to `stream`, and a caller that passes a side stream without entering it (as P2PExchange allows) would
        # otherwise get the compaction racing the all-gather
        mine = local_list[:self.seg_bytes]
        cuda = torch.device(self.device).type == "cuda"
        if cuda and dist.get_backend(self.group) == "gloo":
            # tests only (several ranks sharing ONE device, where RCCL refuses to run): gloo moves host memory — the
            # segments are staged through it, which waits for `stream`.  Never the product's path: bench.py's default
            # backend is nccl, where the branch below runs.
            (stream or torch.cuda.current_stream()).synchronize()
            host = torch.empty(self.world * self.seg_bytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, mine.cpu(), group=self.group)
            with torch.cuda.stream(stream or torch.cuda.current_stream()):
                self.segments.copy_(host.to(self.device, non_blocking=False))
        elif stream is not None and cuda:
            with torch.cuda.stream(stream):
                dist.all_gather_into_tensor(self.segments, mine, group=self.group)
        else:
            dist.all_gather_into_tensor(self.segments, mine, group=self.group)
        self.compact(self.segments, self.world, self.segment_capacity, self.out, self.capacity, self.header, self.stride,
                     stream)
        return self.out

    def probe(self, check_status=None):
        """As P2PExchange.probe: rank r sends r + 1 known items; True on every rank only if every rank received all
        ranks' items in rank order."""
        if self.segment_capacity < self.world:
            return True
        return _probe(self, self.capacity, self.seg_bytes, check_status)

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
