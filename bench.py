#!/usr/bin/env python3
"""bench.py — meshlets culled / s on BASELINE.json config 5 (50 M synthetic meshlets).

One "step" = one pass of the hot path over the scene: entity_cull + meshlet_cull
(frustum + normal cone, occlusion_pass 0) on this rank's entity shard and, for
N > 1, the rank-ordered all-gather of the visible list at record granularity (12 B per dispatch record: entity,
meshlet offset, should-draw ballot — written by the evaluation launch itself; orbit_amd/dist.py) —
what BASELINE.json's north_star names.  By default (--expand own) a rank's cull also ends in the 28-B
MeshletDrawCommandBuffer of its own shard, the indirect-draw buffer it draws from; --expand none leaves that
out.  At N > 1 the transport not chosen is timed in a short leg behind the timed region (`transport_alt`).
Inputs are generated on the device and stay resident in HBM; the timed region is
K steps between barrier + synchronize on both sides, max over ranks.

Prints ONE JSON line on rank 0 (contract: see the task statement / DESIGN.md §Measurement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The host driver of this pool only supports dmabuf IPC: without this, HIP-IPC mappings of the peers' list buffers
# (the p2p transport) and RCCL's own device-memory sharing fail with "hipIpcGetMemHandle: invalid argument".  Set
# before torch (and with it the HIP runtime) is loaded, in EVERY rank — also when the ranks are started by
# torch.distributed.run and not by spawn_ranks() below.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


RUNTIME_WARM_FRAMES = 0  # rounds 4-5: 1024 frames enqueued before anything was timed (see the settle phase); no longer


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--entities", type=int, default=195_313, help="config 5: 195313 x 256 = 50 000 128 meshlets")
    ap.add_argument("--meshlets-per-entity", type=int, default=256)
    ap.add_argument("--survive", type=float, default=0.10, help="target survivor fraction (SURVEY.md §8d prices 10 %%)")
    ap.add_argument("--cpu-sample-entities", type=int, default=195_313,
                    help="entities of the scene the CPU oracle is timed on (rank 0, N=1 only); 0 disables")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="minimum CPU time spent in the oracle baseline")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL gather (cull only)")
    ap.add_argument("--transport", choices=("p2p", "rccl"), default="p2p",
                    help="N>1: how the record lists travel — p2p: direct stores into the peers' IPC-mapped buffers at "
                         "device-computed offsets, counts and completion signalled on the device (orbit_exchange_list; "
                         "default; falls back to rccl when IPC mapping or the device-signalled probe fails); rccl: ONE "
                         "RCCL all-gather of fixed-capacity list segments + a device-side compaction "
                         "(orbit_compact_segments: north_star's transport).  Neither reads a count on the host.  The "
                         "transport NOT chosen is timed in a short leg behind the timed region (transport_alt)")
    ap.add_argument("--no-transport-alt", action="store_true", help="N>1: skip the other transport's leg")
    ap.add_argument("--expand", choices=("own", "none"), default="own",
                    help="N>1: own (default) — every rank's cull ends in BOTH the record list it sends and the 28-B "
                         "MeshletDrawCommandBuffer of its own shard, the indirect-draw buffer it issues its own draws "
                         "from (SURVEY.md §8e; one evaluation, orbit_cull_shard); none — the step is cull + all-gather "
                         "only: the visible list is the product")
    ap.add_argument("--verify-single-rank", action="store_true",
                    help="N>1: behind the timed region rank 0 culls the WHOLE scene alone (one orbit_cull_shard over all "
                         "entity-draws) and compares its record list byte for byte with the gathered one")
    ap.add_argument("--graph", action="store_true",
                    help="N>1 (transport p2p with device signalling, --expand own / none, one frame slot): replay every "
                         "timed step as a captured HIP graph — cull on the slot's stream, the exchange of the previous "
                         "step's list beside it on the exchange stream, fork and join inside the capture — instead of "
                         "enqueueing its launches from Python.  The host then spends 0.03 ms per step instead of 0.045 and "
                         "wall / K stays within 4 %% of the median step interval (eager: 12 %%), but consecutive graph "
                         "launches leave a gap of their own on this runtime: a 1/8 shard's step 0.084 ms against 0.079 "
                         "eager (profiles/r05_notes.md) — so it is an option, not the default")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frame slots (OrbitCtx + stream + buffers each) the steps go round-robin over; 1 = strictly serial "
                         "frames.  0 (default) = 1 at N = 1 (step time = frame latency, a launch's HIP-event duration is "
                         "the kernel's own) and 2 at N > 1, the renderer's FRAME_COUNT (context.rs:1275-1316): a shard's "
                         "cull is a chain of dependent round trips — its entity stage alone is 15-19 us of a 60-us launch "
                         "at a 1/8 shard — which only another frame's work can fill: 0.076 -> 0.049 ms per step there")
    ap.add_argument("--time-every", type=int, default=3,
                    help="the dominant kernel of every n-th launch per frame slot is timed with HIP events")
    ap.add_argument("--meshlet-source", choices=("stream", "buffer"), default="stream",
                    help="what the streaming evaluation reads: the derived meshlet streams (22 B per meshlet — 20.25 B "
                         "with their alpha classes — built once from the meshlet buffer before the timed region; "
                         "include/orbit_abi.h) or the 32-B Meshlet buffer itself")
    ap.add_argument("--scene-shape", choices=("scattered", "coherent"), default="scattered",
                    help="how the ~10 %% survivors lie: scattered uniformly over every entity's meshlets (default, the "
                         "worst case for everything that touches survivors), or coherent — every entity a ball whose "
                         "camera-facing cap survives: runs of neighbouring meshlets, records without any "
                         "(orbit_amd/synth.py)")
    ap.add_argument("--no-mesh-table", action="store_true",
                    help="stream source without orbit_meshlet_stream_update_meshes: the entity stage reads every mesh's "
                         "128-B MeshInfo line instead of its 32-B side entry")
    ap.add_argument("--no-alpha-classes", action="store_true",
                    help="stream source without orbit_meshlet_stream_set_materials: the evaluation reads the 2-B "
                         "material index of every meshlet (22 B per meshlet)")
    ap.add_argument("--settle-with", choices=("stream", "frames"), default="frames",
                    help="what keeps the device busy during the settle: frames of the engine (default: the power management "
                         "follows the workload — see the settle phase) or streaming reads of the meshlet buffer (no frame "
                         "of the engine runs, warmup_effective stays 8 + W; the region then runs ~3 %% slower)")
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="device time spent on back-to-back frames BEFORE the warmup steps, so that the device is in "
                         "its sustained power state when they start: from idle, the firmware's power controller "
                         "overshoots for ~15 ms (frames 5-45 run up to 25 %% slower than both the first frames and "
                         "the steady state, tools/sustained.sh) — exactly where 5 warmup + 20 timed steps would sit. "
                         "Not steps: reported as config.settle_ms; 0 disables")
    ap.add_argument("--runtime-warm-frames", type=int, default=RUNTIME_WARM_FRAMES,
                    help="frames a process has enqueued before anything is timed (the settle phase runs at least until "
                         "then; see RUNTIME_WARM_FRAMES)")
    ap.add_argument("--planes", type=int, default=5, help="cull planes used (5 = forward.rs:268; fewer only for triage)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: this process touches no GPU (importing torch does
    not initialise HIP) and starts one fresh child per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set — the
    environment `python -m torch.distributed.run --nproc-per-node N` would have given them.  Rank 0's stdout (the ONE
    JSON line) is forwarded; the exit code is non-zero if any rank failed.  Nothing is re-executed in a process that
    has initialised the GPU."""
    import socket
    import subprocess

    # A free port, found by binding to port 0.  The socket is bound, never listening, and closed once the children
    # exist: between that and rank 0's TCPStore binding the port (after it has imported torch) the port is NOT
    # reserved — another process could take it, and the rendezvous would then fail with EADDRINUSE within the job's
    # deadline below (a loud failure, not a hang).  SO_REUSEADDR lets the TCPStore bind right behind this socket.
    sk = socket.socket()
    sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    sk.close()
    # Rank 0's stdout is drained by a thread while ALL children are polled: the first rank that dies takes the others
    # with it (they would otherwise sit in a rendezvous or a collective until the process group's timeout), and the
    # whole job has a deadline.
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("ORBIT_BENCH_TIMEOUT_S", "1800"))
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.time() > deadline:
            failed = bad or [("deadline", -1)]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.time() + 10
            while any(p.poll() is None for p in procs) and time.time() < t_kill:
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=5)
    sys.stdout.buffer.write(out0[0] if out0 else b"")
    sys.stdout.flush()
    bad = failed or [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit(f"bench.py: rank(s) failed (rank, exit code): {bad}; exit codes of all ranks: {rcs}")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL writes its version banner to
    # stdout, C-buffered, i.e. after anything Python printed): keep the real stdout aside for the JSON line and
    # point fd 1 at stderr for everything else, in every rank.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch N ranks for --gpus N")
    # Dry-run knobs for boxes with fewer GPUs than ranks (never set by the driver): all ranks on device 0 and a
    # gloo process group exercise the N>1 control flow (sharding, barriers, reductions, JSON); gloo cannot move
    # device buffers point-to-point, so the visible-list gather is skipped there.  ORBIT_BENCH_FORCE_DIST=1 runs
    # the RCCL group and the gather with a world of one.  ORBIT_BENCH_DRY_RANKS=1 (CPU test of the launcher): the
    # ranks rendezvous over gloo, exchange their shard ranges and stop before any device work.
    backend = os.environ.get("ORBIT_BENCH_BACKEND", "nccl")
    if os.environ.get("ORBIT_BENCH_DRY_RANKS") == "1":
        return dry_ranks(args, rank, world, json_fd)
    device_index = 0 if os.environ.get("ORBIT_BENCH_ONE_DEVICE") == "1" else local_rank
    use_dist = world > 1 or os.environ.get("ORBIT_BENCH_FORCE_DIST") == "1"
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": dev} if backend == "nccl" else {}))
        else:
            dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    from orbit_amd import dist as odist
    from orbit_amd import layouts as L
    from orbit_amd import camera, synth
    from orbit_amd.engine import Engine

    spec = synth.C5Spec(entities=args.entities, meshlets_per_entity=args.meshlets_per_entity)
    E, M = spec.entities, spec.meshlets_per_entity
    b, e = odist.shard_ranges(E, world)[rank]
    n_local_entities = e - b
    n_local_meshlets = n_local_entities * M
    rec_per_ent = spec.records_per_entity

    # N > 1: the exchange step — the rank-ordered all-gather of the ranks' visible lists at record granularity: 12 B
    # {entity_index, meshlet_offset, should-draw ballot} per dispatch record, written by the evaluation launch itself
    # (orbit_cull_shard).  Transport "p2p" (default): every rank stores its list straight into every peer's buffer
    # (IPC-mapped, xGMI) at device-computed offsets, counts and completion signalled on the device (orbit_exchange_list,
    # orbit_amd/dist.py P2PExchange).  Transport "rccl": ONE ncclAllGather of fixed-capacity segments + a device-side
    # compaction (AllGatherExchange).  No count is read on the host by either; a rank holds only its meshlet shard.
    do_gather = use_dist and not args.no_gather
    records_exchange = do_gather
    own_expand = records_exchange and args.expand == "own"  # this rank's commands from this rank's evaluation
    t0 = time.time()
    draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
    mb, me = b, e
    meshlets = synth.gen_meshlets(spec, mb, me, dev, half, survive_target=args.survive, shape=args.scene_shape)
    materials = synth.gen_materials(spec, dev)
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    # buffer passed under global meshlet indices: base pointer = pointer - first_meshlet * 32
    meshlets_base = meshlets.data_ptr() - mb * M * 32

    # camera outside the +-512 m box so that every entity passes the frustum: all E*M meshlets
    # reach meshlet_cull (the HBM-bound case the metric is about)
    # the renderer's default camera moved to (0, 0, 1300), through the product's host mirror (orbit_amd/camera.py)
    ci = camera.frame_cull_info((0.0, 0.0, 1300.0), n_planes=args.planes)

    disp_cap = n_local_entities * rec_per_ent + 8
    draw_cap = n_local_meshlets // 2 + 1024
    # Frame slots.  The renderer keeps FRAME_COUNT = 2 frames in flight with per-frame transient buffers
    # (context.rs:1275-1316); here a frame slot owns an OrbitCtx (scan scratch), a stream, a dispatch buffer and its
    # list buffers, and the K steps go round-robin over the slots.  With two slots frame k+1's entity cull and the head
    # of its streaming evaluation run under frame k's scan / emit and the tail of its evaluation
    # (tools/bench_pipeline.py, in-process alternation: 0.419 -> 0.402 ms per frame; ordering the evaluations with
    # events instead: 0.434, worse than serial) — but then the evaluations of consecutive frames share the device for
    # part of their run and the HIP-event duration of a launch is no longer that kernel's own time.  The default is
    # ONE slot: strictly serial frames, step time = frame latency, clean per-kernel durations.
    # N > 1: the exchange of step i runs on its own stream while step i+1 culls into the slot's other list buffer.
    F = args.frames_in_flight if args.frames_in_flight > 0 else (2 if use_dist else 1)
    hdr, item = (L.VISIBLE_HEADER, 12) if records_exchange else (L.DRAW_HEADER, 28)
    # the record list is also a fixed-capacity segment of an RCCL all-gather: the same capacity on every rank
    seg_cap = max(e_ - b_ for b_, e_ in odist.shard_ranges(E, world)) * rec_per_ent + 8
    list_cap = seg_cap if records_exchange else draw_cap  # entries a rank's list can hold (records / commands)

    class Slot:
        def __init__(self):
            self.eng = Engine(device_index, max_entities=n_local_entities + 256, max_dispatches=disp_cap,
                              max_draws=draw_cap, cull_path=int(os.environ.get("ORBIT_BENCH_CULL_PATH", "0")))  # triage
            self.stream = torch.cuda.Stream(device=dev)
            self.disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
            self.lists = [torch.zeros(hdr + item * list_cap, dtype=torch.uint8, device=dev)
                          for _ in range(2 if do_gather else 1)]
            # --expand own: the shard's indirect-draw buffer, rebuilt every step from the shard's record list
            self.own_cmds = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev) if own_expand else None
            self.cull_done = [torch.cuda.Event() for _ in self.lists]
            self.xdone = [torch.cuda.Event() for _ in self.lists]
            self.uses = 0

    slots = [Slot() for _ in range(F)]
    # Derived meshlet streams: static geometry re-laid out once (as the renderer uploads a mesh once), not per step.
    mstream, mstream_build_ms = None, None
    if args.meshlet_source == "stream":
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mstream = slots[0].eng.meshlet_stream(meshlets_base, mb * M, (me - mb) * M)
        if not args.no_alpha_classes:  # as the renderer would after add_material: the evaluation reads no material index
            mstream.set_materials(materials, spec.materials)
        if not args.no_mesh_table:  # as the renderer would after add_mesh: the entity stage reads 32 B per mesh, not 128
            mstream.update_meshes(mesh, 0, E)
        torch.cuda.synchronize()
        mstream_build_ms = (time.perf_counter() - t0) * 1e3
        for s in slots:
            s.eng.bind_meshlet_stream(mstream)
    # The exchange stage: two exchange buffers (alternating) on a stream of its own, so that the cull of step i+1 (slot
    # stream) and the wire of step i (exchange stream) overlap in steady state: step time = max(cull, wire), not their sum.
    xstream = None
    exchanges, recv_bufs = [], []
    transport = transport_fallback = None
    x_cap = list_cap * world  # entries of the gathered list
    n_x = 2

    def make_exchanges(name):
        """The exchange objects of transport `name`, agreed on by all ranks: (list, None) or (None, why not)."""
        ok, made, why = 1, [], None
        try:
            if name == "p2p":
                made = [odist.P2PExchange(slots[0].eng, x_cap, hdr, item, dev) for _ in range(n_x)]
                fell = [x.fallback for x in made if x.fallback]
                if fell:  # the device-signalled exchange failed its probe between the devices of THIS node (all ranks alike)
                    ok, why = 0, fell[0]
            else:
                made = [odist.AllGatherExchange(slots[0].eng, seg_cap, x_cap, hdr, item, dev) for _ in range(n_x)]
                if not all([x.probe(check_status=slots[0].eng.status) for x in made]):
                    ok, why = 0, "the all-gather exchange failed its probe"
        except Exception as ex:  # noqa: BLE001 - e.g. IPC not available between these processes: agreed on below
            print(f"bench.py: transport {name} unavailable on rank {rank} ({ex})", file=sys.stderr)
            ok, why = 0, f"{name}: {ex}"
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            return made, None
        for x in made:
            try:
                x.close()
            except Exception:  # noqa: BLE001
                pass
        return None, why or f"{name} failed on another rank"

    if do_gather:
        xstream = torch.cuda.Stream(device=dev)
        transport = args.transport
        exchanges, why = make_exchanges(transport)
        if exchanges is None and transport == "p2p":  # every rank lands on north_star's transport together
            transport_fallback = f"p2p -> rccl: {why}"
            print(f"bench.py: {transport_fallback}", file=sys.stderr)
            transport = "rccl"
            exchanges, why = make_exchanges(transport)
        if exchanges is None:
            raise SystemExit(f"bench.py: no transport for the visible list: {why}")
        recv_bufs = [x.out for x in exchanges]
    pending = []  # (slot, list index) culled but not yet exchanged (at most one)
    x_count = [0]

    def gather(s, k):
        # the exchange of one list on the exchange stream, behind the cull that wrote the list — not behind the cull
        # enqueued after it.  The device-signalled p2p exchange is enqueued by the library on the stream it is given (no
        # stream context: torch's costs ~10 us of host time per step); the all-gather is a torch.distributed call and goes
        # to the stream AllGatherExchange makes current
        j = x_count[0] % len(recv_bufs)
        x_count[0] += 1
        xstream.wait_event(s.cull_done[k])
        exchanges[j].exchange(s.lists[k], stream=xstream)
        s.xdone[k].record(xstream)  # the list buffer may be overwritten
        return j

    step_no = [0]

    shard_calls = {}  # (slot, list index) -> the prepared orbit_cull_shard call (argument blocks built once)

    def cull(s, k):
        if records_exchange:
            # a rank's whole cull as ONE call (orbit_cull_shard) and, up to 65 536 entity-draws, ONE launch: entity test,
            # record append, meshlet test and the record list the rank sends — plus, with --expand own, the emit launch for
            # the 28-B commands of its own shard (its indirect-draw buffer, "each rank issues its own draws": SURVEY §8e)
            call = shard_calls.get((id(s), k))
            if call is None:
                call = shard_calls[(id(s), k)] = s.eng.prepare_cull_shard(
                    ci, draws, mesh, s.disp, ent, b, n_local_entities, disp_cap, meshlets_base, materials, s.lists[k], list_cap,
                    draw_commands_buffer=s.own_cmds if own_expand else None, draw_capacity=draw_cap if own_expand else 0,
                    material_count=spec.materials, stream=s.stream)
            call()
            return
        s.eng.entity_cull(ci, draws, mesh, s.disp, ent, n_local_entities, disp_cap, draw_first=b, stream=s.stream)
        s.eng.meshlet_cull(ci, s.disp, meshlets_base, s.lists[k], ent, materials, disp_cap, draw_cap,
                           material_count=spec.materials, stream=s.stream)

    step_marks = []  # N > 1: one timing event per step of the timed region, behind that step's exchange
    mark_ring = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 2)] if do_gather else []  # created (and
    # recorded once, below) before the region: creating an event per step inside it is host work in the step
    for ev in mark_ring:
        ev.record()

    def step(i=None):
        s = slots[step_no[0] % F]
        step_no[0] += 1
        k = s.uses % len(s.lists)
        s.uses += 1
        if do_gather:
            s.stream.wait_event(s.xdone[k])  # the exchange that read this list buffer two uses ago
        cull(s, k)
        if do_gather:
            s.cull_done[k].record(s.stream)
            if pending:
                gather(*pending.pop())  # previous step's list travels while this step's cull runs
                if i is not None:
                    ev = mark_ring[len(step_marks)]
                    ev.record(xstream)
                    step_marks.append(ev)
            pending.append((s, k))

    def drain():
        while pending:
            gather(*pending.pop())
        for s in slots:
            torch.cuda.current_stream().wait_stream(s.stream)
        if xstream is not None:
            torch.cuda.current_stream().wait_stream(xstream)

    # ---- before the timed region (config.warmup_effective counts every frame of it):
    #   1. eight cold frames: code objects, first touches of every buffer;
    #   2. the power-state settle (--settle-ms, default 60): from idle, the device's power management needs ~50 ms OF THIS
    #      WORKLOAD to reach the clocks it then holds — W + K = 25 steps straight from idle run at 0.30 ms per step, after
    #      ~200 back-to-back frames at 0.259-0.260 (profiles/r06_notes.md §settle).  Keeping the device busy with something
    #      else does not do it: 60, 300 or 1 000 ms of streaming reads of the same meshlet buffer (--settle-with stream)
    #      leave the region at 0.268-0.272 — the governor follows the workload, not the clock on the wall.  So the settle
    #      is frames of the engine, as many as fill --settle-ms: ~230 at config 5.
    #      Rounds 4-5 ran at least 1 024 frames here (--runtime-warm-frames) to keep the HIP runtime's one long host-side
    #      stall of a process out of the region.  That stall comes when the enqueuing thread is ~1 200 commands AHEAD of the
    #      device (tools/stall_find.py, profiles/r06_notes.md §stall: after frame 393 with three launches per frame, 336 with
    #      a fresh timing event per frame as well, 200 with two; never when the host is slowed down) — a state a region of
    #      K <= 300 steps entered through a synchronisation does not reach, and neither do the ~230 settle frames;
    #   3. the measurement hook's event pairs, created before the region (orbit_ctx_profile_reserve) instead of by
    #      ceil(K / 3) + 1 extra frames;
    #   4. the W warm-up steps the command line asks for.
    #   The SURVEY §8d protocol run and the from-idle W + K steps — reported, never the metric — now run BEHIND the region.
    frames_before_timed = 0
    settle_frames = 0
    for _ in range(8):
        step()
    drain()
    torch.cuda.synchronize()
    frames_before_timed += 8
    if args.settle_ms > 0 and args.settle_with == "stream":
        words = meshlets.view(torch.float32)
        t_end = time.perf_counter() + args.settle_ms * 1e-3
        while time.perf_counter() < t_end:
            for _ in range(8):
                words.sum()
            torch.cuda.synchronize()
    elif args.settle_ms > 0:
        # (the frame count must be the same on every rank — a step of the sharded engine may contain collectives — so it is
        # derived from the slowest rank's time for eight frames, not from each rank's own clock)
        t_settle = time.perf_counter()
        for _ in range(8):
            step()
        drain()
        torch.cuda.synchronize()
        per_frame = torch.tensor([(time.perf_counter() - t_settle) / 8], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(per_frame, op=dist.ReduceOp.MAX)
        more = int(np.ceil(args.settle_ms * 1e-3 / max(float(per_frame.item()), 1e-6))) - 8
        more = min(max(max(more, args.runtime_warm_frames - 16), 0), 8192)
        for _ in range(more):
            step()
        drain()
        torch.cuda.synchronize()
        settle_frames = 8 + more
        frames_before_timed += settle_frames
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    frames_before_timed += args.warmup
    # HIP events around the dominant kernel (meshlet_eval), recorded by the library on the launch stream.  An event
    # pair costs ~7 us of stream time (it keeps the next launch from being prefetched: tools/t_host.py), so every
    # `--time-every`-th launch per slot is timed, not all of them, and no other event is recorded inside the region.
    # The library creates a launch's event pair on first use (hipEventCreate + the event's first record: not free): every
    # pair the timed region will record is created before it (orbit_ctx_profile_reserve), on the slot's stream.
    prewarm = -(-args.steps // max(args.time_every, 1)) + 1
    # N > 1: the step as a captured graph (--graph).  Two graphs, because consecutive steps alternate between the
    # slot's two list buffers and the two exchange buffers: graph k = { cull into list k  ||  exchange of list 1 - k, the
    # previous step's } — fork and join inside the capture, so that a replay is the steady-state step: K replays are K
    # culls and K exchanges, each exchange beside the cull of the step after it.
    use_graph = bool(do_gather and args.graph and F == 1 and transport == "p2p")
    graphs = []
    if use_graph:
        s0g = slots[0]
        torch.cuda.synchronize()
        for k in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s0g.stream):
                with torch.cuda.graph(g, stream=s0g.stream):
                    xstream.wait_stream(s0g.stream)
                    cull(s0g, k)
                    with torch.cuda.stream(xstream):
                        exchanges[1 - k].exchange(s0g.lists[1 - k], stream=xstream)
                    s0g.stream.wait_stream(xstream)
            graphs.append(g)
        if use_dist:
            dist.barrier()  # every rank has captured (capture enqueues nothing)
        for i in range(4):  # the replays' own warm-up
            with torch.cuda.stream(s0g.stream):
                graphs[i % 2].replay()
        torch.cuda.synchronize()
        frames_before_timed += 4
    else:
        for s in slots:
            s.eng.profile_reserve(prewarm + 1, stream=s.stream)
            s.eng.profile(True, every=args.time_every)
        torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if use_graph:
        with torch.cuda.stream(s0g.stream):
            for i in range(args.steps):
                graphs[i % 2].replay()
                mark_ring[i].record(s0g.stream)
                step_marks.append(mark_ring[i])
        t_enqueued = time.perf_counter()
    else:
        for i in range(args.steps):
            step(i)
        t_enqueued = time.perf_counter()
        drain()  # every one of the K lists has been produced (and, N > 1, exchanged) inside the timed region
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # N > 1: the intervals between the completions of consecutive steps' exchanges inside the timed region (HIP events on
    # the exchange stream) — the steady-state step time without whatever a single slow step adds to wall / K
    step_intervals = None
    if len(step_marks) >= 3:
        iv = [step_marks[i].elapsed_time(step_marks[i + 1]) for i in range(len(step_marks) - 1)]
        step_intervals = {"median_ms": float(np.median(iv)), "min_ms": float(min(iv)), "max_ms": float(max(iv)),
                          "intervals": len(iv),
                          "what": "time between the completions of consecutive steps' exchanges inside the timed region"}
    if mstream is not None and not args.no_alpha_classes:
        assert all(s.eng.meshlet_class_culls() >= args.steps // F for s in slots), "the alpha classes were not used"
    host_enqueue_ms_per_step = (t_enqueued - t0) / args.steps * 1e3  # what the host spent handing the K steps over
    if use_graph:  # the dominant kernel's launches, timed by the library in eager steps behind the region (an event pair
        # recorded inside a captured graph cannot be read)
        for s in slots:
            s.eng.status()
            s.eng.profile(True, every=1)
        for _ in range(3):
            step()
        drain()
        torch.cuda.synchronize()
        for s in slots:
            s.eng.profile_read()
            s.eng.profile(True, every=1)
        for _ in range(max(8, args.steps // 2)):
            step()
        drain()
        torch.cuda.synchronize()
    eval_sum, eval_launches = 0.0, 0
    for s in slots:
        s.eng.status()
        ms, n = s.eng.profile_read()
        eval_sum += ms * n
        eval_launches += n
        s.eng.profile(False)
    assert eval_launches >= 1, "no launch of the dominant kernel was timed"
    eval_ms = eval_sum / eval_launches

    # For the record, behind the timed region (never the metric): the same W + K steps straight from an idle device, i.e.
    # inside the power controller's transient — what the settle phase is there to keep out of the number.
    time.sleep(0.25)
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    tc = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize()
    cold = torch.tensor([(time.perf_counter() - tc) / args.steps * 1e3], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(cold, op=dist.ReduceOp.MAX)
    cold_ms_per_step = float(cold.item())
    # SURVEY.md §8d, literally: the 32-B Meshlet buffer (no derived stream), 5 warm-up steps from an idle device, then
    # >= 20 steps each bracketed by its own HIP event pair on the launch stream; the median.  Never the headline (that is
    # the settled stream path above), always in the line — N = 1 only (a step of N > 1 contains an exchange).
    protocol_8d = None
    if world == 1 and not use_dist:
        s0 = slots[0]
        if mstream is not None:
            s0.eng.bind_meshlet_stream(None)
        time.sleep(0.25)
        for _ in range(5):
            cull(s0, 0)
        n8 = max(20, args.steps)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n8)]
        for a, z in evs:
            a.record(s0.stream)
            cull(s0, 0)
            z.record(s0.stream)
        torch.cuda.synchronize()
        ms8 = sorted(a.elapsed_time(z) for a, z in evs)
        protocol_8d = {"what": "SURVEY.md §8d as written: Meshlet buffer (32 B per meshlet, no derived stream, no "
                               "update contract), 5 warm-up steps from idle (a 0.25-s pause behind the timed region), no "
                               "settle phase, then one HIP event pair per step; median",
                       "steps": n8, "warmup": 5, "ms_per_step_median": ms8[len(ms8) // 2], "ms_per_step_min": ms8[0],
                       "ms_per_step_max": ms8[-1], "meshlets_per_sec": n_local_meshlets / (ms8[len(ms8) // 2] * 1e-3)}
        if mstream is not None:
            s0.eng.bind_meshlet_stream(mstream)

    # one frame alone on an idle device (nothing else in flight): the latency a single cull costs, and its parts
    s0 = slots[0]
    serial, op_ms, x_ms = [], [], []
    last_recv = 0
    for _ in range(5):
        a, m, z = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record(s0.stream)
        if records_exchange:  # one call, one launch: there is no "meshlet stage alone"
            m = a
            cull(s0, 0)
        else:
            s0.eng.entity_cull(ci, draws, mesh, s0.disp, ent, n_local_entities, disp_cap, draw_first=b, stream=s0.stream)
            m.record(s0.stream)
            s0.eng.meshlet_cull(ci, s0.disp, meshlets_base, s0.lists[0], ent, materials, disp_cap, draw_cap,
                                material_count=spec.materials, stream=s0.stream)
        z.record(s0.stream)
        torch.cuda.synchronize()
        serial.append(a.elapsed_time(z))
        op_ms.append(m.elapsed_time(z))
        if do_gather:  # the exchange of that list alone
            dist.barrier()
            xa, xm = (torch.cuda.Event(enable_timing=True) for _ in range(2))
            s0.cull_done[0].record(s0.stream)
            xa.record(xstream)
            last_recv = gather(s0, 0)
            xm.record(xstream)
            torch.cuda.synchronize()
            x_ms.append(xa.elapsed_time(xm))
    # the same frames from the 32-B Meshlet buffer (stream unbound), for the record, outside any metric
    buffer_path_latency_ms = buffer_path_ms_per_step = None
    if mstream is not None and not do_gather:
        s0.eng.bind_meshlet_stream(None)
        # K of them back to back (the timed region's loop, stream unbound; K more in front as its warm-up) ...
        for _ in range(args.steps):
            cull(s0, 0)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for _ in range(args.steps):
            cull(s0, 0)
        torch.cuda.synchronize()
        buffer_path_ms_per_step = (time.perf_counter() - tb) / args.steps * 1e3
        # ... and five alone
        lat = []
        for _ in range(6):
            a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s0.stream)
            cull(s0, 0)
            z.record(s0.stream)
            torch.cuda.synchronize()
            lat.append(a.elapsed_time(z))
        buffer_path_latency_ms = float(np.median(lat[1:]))
        s0.eng.bind_meshlet_stream(mstream)
        cull(s0, 0)  # slot 0's buffers hold the stream path's result again: that is what gets checked below
        torch.cuda.synchronize()
    frame_latency_ms = float(np.median(serial))
    kernel_ms = float(np.median(op_ms))  # whole meshlet_cull op (eval + scan + emit) of a frame alone on the device
    exchange_ms = float(np.median(x_ms)) if x_ms else None

    disp, draw = s0.disp, s0.lists[0]  # the latency frames above ran on slot 0: its buffers hold the last result
    n_rec = int(disp[:4].view(torch.int32).item())
    n_list = int(draw[:4].view(torch.int32).item())  # entries of this rank's list: records, items or commands
    n_draw = int(draw[4:8].view(torch.int32).item()) if records_exchange else n_list  # its survivors
    transport_alt = single_rank_check = None
    if do_gather:  # the gathered list holds this rank's entries at its rank-order position
        counts = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([n_list, n_draw], dtype=torch.int64, device=dev))
        first = int(sum(int(c[0].item()) for c in counts[:rank]))
        wire = recv_bufs[last_recv]  # what travelled: 12-B records, rank order
        assert torch.equal(wire[hdr + item * first:hdr + item * (first + n_list)], draw[hdr:hdr + item * n_list]), \
            "gathered list != local list"
        total = int(wire[:4].view(torch.int32).item())
        assert total == int(sum(int(c[0].item()) for c in counts))

        def list_checksum(buf):
            words = buf[hdr:hdr + item * total].view(torch.int32).to(torch.int64)
            return torch.stack([words.sum(), (words * (torch.arange(words.numel(), device=dev) % 8191 + 1)).sum()])

        # the whole gathered list is the same on every rank: a checksum of it must agree
        chk = list_checksum(wire)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), "ranks hold different gathered lists"
        gathered_main = wire[:hdr + item * total].clone()

        # The OTHER transport, behind the timed region (as protocol_8d at N = 1: in the line, never the metric): the same
        # steps with the exchange objects swapped — so that ONE scaling run compares north_star's RCCL all-gather with the
        # direct xGMI stores on the same node, scene and step.  Its gathered list must be the main transport's, byte for byte.
        if not args.no_transport_alt:
            alt_name = "rccl" if transport == "p2p" else "p2p"
            alt, why = make_exchanges(alt_name)
            if alt is None:
                transport_alt = {"name": alt_name, "ms_per_step": None, "unavailable": why}
            else:
                main_ex, main_recv = exchanges[:], recv_bufs[:]
                exchanges[:], recv_bufs[:] = alt, [x.out for x in alt]
                for _ in range(max(args.warmup, 2) + 2 * F):
                    step()
                drain()
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                ta = time.perf_counter()
                for _ in range(args.steps):
                    step()
                drain()
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                alt_t = torch.tensor([time.perf_counter() - ta], dtype=torch.float64, device=dev)
                dist.all_reduce(alt_t, op=dist.ReduceOp.MAX)
                slots[0].eng.status()
                j = (x_count[0] - 1) % len(recv_bufs)
                same = torch.tensor([1 if torch.equal(recv_bufs[j][:hdr + item * total], gathered_main) else 0],
                                    dtype=torch.int32, device=dev)
                dist.all_reduce(same, op=dist.ReduceOp.MIN)
                if int(same.item()) != 1:
                    raise SystemExit(f"bench.py: transport {alt_name} delivered a different list than {transport}")
                transport_alt = {"name": alt_name, "ms_per_step": float(alt_t.item()) / args.steps * 1e3, "steps": args.steps,
                                 "signalling": alt[0].signalling, "list_identical_to_main_transport": True,
                                 "what": "the same steps with the other transport's exchange objects, behind the timed "
                                         "region; wall / K between barrier + synchronize, max over ranks"}
                exchanges[:], recv_bufs[:] = main_ex, main_recv
                for x in alt:
                    x.close()
        # --verify-single-rank: the gathered list against ONE rank culling the whole scene (rank 0, the Meshlet buffer —
        # no derived stream: another evaluation path — through one orbit_cull_shard over all entity-draws)
        if args.verify_single_rank:
            ok = 1
            if rank == 0:
                full = synth.gen_meshlets(spec, 0, E, dev, half, survive_target=args.survive, shape=args.scene_shape)
                cap_f = E * rec_per_ent + 8
                eng_f = Engine(device_index, max_entities=E + 256, max_dispatches=cap_f, max_draws=draw_cap)
                disp_f = torch.zeros(L.DISPATCH_HEADER + 16 * cap_f, dtype=torch.uint8, device=dev)
                rec_f = torch.zeros(hdr + item * cap_f, dtype=torch.uint8, device=dev)
                eng_f.cull_shard(ci, draws, mesh, disp_f, ent, 0, E, cap_f, full, materials, rec_f, cap_f,
                                 material_count=spec.materials)
                torch.cuda.synchronize()
                eng_f.status()
                n_f = int(rec_f[:4].view(torch.int32).item())
                if n_f != total or not torch.equal(rec_f[hdr:hdr + item * n_f], gathered_main[hdr:]):
                    ok = 0
                eng_f.close()
                del full, disp_f, rec_f
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                raise SystemExit("bench.py: the gathered list differs from a single rank's list of the whole scene")
            single_rank_check = {"gathered_list_equals_single_rank_list": True, "records": total}
    if own_expand:  # the shard's indirect-draw buffer holds exactly the survivors the shard's record list names, in order
        cmds = s0.own_cmds
        assert int(cmds[:4].view(torch.int32).item()) == n_draw, "own command buffer: wrong count"
        cw = cmds[L.DRAW_HEADER:L.DRAW_HEADER + 28 * n_draw].view(torch.int32).view(-1, 7)
        recs = draw[hdr:hdr + 12 * n_list].view(torch.int32).view(-1, 3)
        bits = (recs[:, 2:3] >> torch.arange(32, device=dev, dtype=torch.int32)) & 1
        r_idx, b_idx = torch.nonzero(bits, as_tuple=True)
        assert torch.equal(cw[:, 6], recs[r_idx, 1] + b_idx.to(torch.int32)) and torch.equal(cw[:, 4], recs[r_idx, 0]), \
            "own command buffer != the shard's visible list"
    for s in slots:  # every slot and list buffer computed the same frame
        for l in s.lists:
            assert torch.equal(l[:hdr + item * n_list], draw[:hdr + item * n_list]), "frame slots disagree"

    n_in = int(disp[L.DISPATCH_HEADER:L.DISPATCH_HEADER + 16 * n_rec].view(torch.int32).view(-1, 4)[:, 2].sum().item())
    # cull-only rate of this rank (for the N>1 breakdown)
    if use_dist:
        agg = torch.tensor([n_in, n_draw], dtype=torch.int64, device=dev)
        dist.all_reduce(agg)
        tot_in, tot_draw = int(agg[0].item()), int(agg[1].item())
        km = torch.tensor([kernel_ms, frame_latency_ms, exchange_ms or 0.0, 0.0, float(n_in), float(n_draw)],
                          dtype=torch.float64, device=dev)
        per = [torch.zeros_like(km) for _ in range(world)]
        dist.all_gather(per, km)
        per = torch.stack(per).cpu().numpy()  # [rank, (meshlet_cull op, frame, exchange, -, meshlets in, survivors)]
        kernel_ms_max, kernel_ms_max_frame = float(per[:, 0].max()), float(per[:, 1].max())
        exchange_ms_max = float(per[:, 2].max()) if exchange_ms is not None else None
        per_rank_rows = {"cull_ms": [float(v) for v in per[:, 1]],
                         "exchange_ms": [float(v) for v in per[:, 2]] if exchange_ms is not None else None,
                         "meshlets_in": [int(v) for v in per[:, 4]], "survivors": [int(v) for v in per[:, 5]]}
        world_reported = dist.get_world_size()
    else:
        tot_in, tot_draw, kernel_ms_max, kernel_ms_max_frame = n_in, n_draw, kernel_ms, frame_latency_ms
        exchange_ms_max = None
        per_rank_rows = {"cull_ms": [frame_latency_ms], "exchange_ms": None,
                         "meshlets_in": [n_in], "survivors": [n_draw]}
        world_reported = 1

    ms_per_step = elapsed / args.steps * 1e3
    value = tot_in / (elapsed / args.steps)

    # Roofline of the dominant kernel (meshlet_eval, the streaming evaluation), this rank.
    # B_m = 32 N_m + 16 N_d + 64 N_e + 28 N_draw + 4 for the whole op (occlusion_pass 0);
    # the eval launch carries the read side (32 N_m + 16 N_d + 64 N_e = 32.75 B per meshlet here), the emit
    # launch the 28 N_draw + 4 written bytes (DESIGN.md "Algorithmic bytes").
    # From the derived meshlet streams the evaluation reads 22 B per meshlet (sphere 16, cone 4, material index 2)
    # instead of the 32-B Meshlet, and the emit launch streams 2 B of counts per meshlet of the records that have
    # survivors (the command chain) instead of a payload.
    # (20.25 with the streams' alpha classes: two class bits instead of the 2-B material index)
    per_meshlet = (20.25 if not args.no_alpha_classes else 22) if mstream is not None else 32
    alg_bytes = per_meshlet * n_in + 16 * n_rec + 64 * n_local_entities
    achieved = alg_bytes / (eval_ms * 1e-3) / 1e9
    # the chain emit: records again, 2-B counts of every meshlet (records without survivors are skipped: a few %), 64-B
    # ballots per tile, one link bit per meshlet
    emit_read = (16 * n_rec + 2 * n_in + n_rec * 4 + n_in // 8) if (mstream is not None and not records_exchange) else 0
    op_bytes = alg_bytes + emit_read + item * n_list + 4
    # the PMC passes were taken on the whole scene at N=1
    traffic, traffic_stamp = load_traffic(E, M, args.meshlet_source) if world == 1 else (None, None)
    lib_sha = lib_sha256()

    # Box reference (not part of any metric): what a plain streaming read of the same bytes reaches on THIS box right
    # now.  Boxes of the pool differ by up to 30 % (profiles/r01_notes.md, "Measurement hygiene"); the ratio below is
    # comparable across them, the absolute numbers are not.
    stream_ref = None
    if rank == 0 and world == 1:
        words = meshlets.view(torch.float32)  # the value is irrelevant (NaN bit patterns included), only the read
        se = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
        for a, z in se:
            a.record()
            words.sum()
            z.record()
        torch.cuda.synchronize()
        sms = float(np.median([a.elapsed_time(z) for a, z in se[2:]]))
        stream_ref = {"GBps": words.numel() * 4 / (sms * 1e-3) / 1e9,
                      "what": f"torch.sum over the {words.numel() * 4 / 1e9:.2f} GB meshlet buffer viewed as f32, median of 5",
                      "meshlet_eval_vs_stream": achieved / (words.numel() * 4 / (sms * 1e-3) / 1e9)}

    cpu_baseline = None
    if rank == 0 and world == 1 and args.cpu_sample_entities > 0:
        cpu_baseline = run_cpu_baseline(args, spec, ci, draws, mesh, ent, meshlets, materials, disp, draw, L)

    # "frustum + cone (pass 0) as headline, two-pass as secondary" — the same scene through the whole
    # two-pass HiZ frame, after everything that looks at the headline frame's buffers (it reuses slot 0's), N = 1 only.
    two_pass = run_two_pass(args, spec, slots[0], draws, mesh, ent, meshlets_base, materials, disp_cap, draw_cap, dev, L) \
        if (world == 1 and not use_dist) else None

    if rank == 0:
        out = {
            "metric": "meshlets_culled_per_sec",
            "value": value,
            "unit": "meshlets/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"config5: {E} entities x {M} = {E * M} synthetic meshlets, entity_cull + meshlet_cull, "
                            f"frustum + normal cone (occlusion_pass 0), all entities in view, "
                            f"{100.0 * tot_draw / max(tot_in, 1):.1f}% survivors ({args.scene_shape})"
                            + ((", + rank-ordered all-gather of the visible list (12 B per dispatch record: entity, "
                                "meshlet offset, should-draw ballot)")
                               + (" by direct xGMI stores at device-computed offsets" if transport == "p2p" else
                                  " by ONE RCCL all-gather of fixed-capacity list segments + a device-side compaction")
                               if do_gather else ""),
                "entities": E, "meshlets": E * M, "meshlets_into_meshlet_cull": tot_in, "survivors": tot_draw,
                "scene_shape": args.scene_shape,
                "sharding": f"entity ranges over {world} rank(s)", "gather": bool(do_gather),
                "frames_in_flight": F,
                "settle_ms": args.settle_ms, "settle_frames": settle_frames,
                "warmup_effective": frames_before_timed,  # every frame this process ran before the timed region
                "settle_with": args.settle_with,
                "ms_per_step_without_settle": cold_ms_per_step,  # W + K steps from idle, behind the timed region
                "world_size": world_reported, "backend": (backend if use_dist else None),
                "expand": (args.expand if do_gather else None),
                "step_product": (None if not do_gather else
                                 "the all-gathered visible list on every GPU + every rank's own shard as a 28-B "
                                 "MeshletDrawCommandBuffer (its indirect draws)" if own_expand else
                                 "the all-gathered visible list only (no command buffer is built inside the step)"),
                "signalling": (exchanges[0].signalling if exchanges else None),
                "step_intervals": step_intervals,
                "steps_as_graph": (use_graph if do_gather else None),
                "host_enqueue_ms_per_step": host_enqueue_ms_per_step,
                "list_bytes_per_rank_max": (int(max(c[0].item() for c in counts)) * item if do_gather else None),
                "exchange": ("records" if do_gather else None), "transport": transport,
                "transport_requested": (args.transport if do_gather else None), "transport_fallback": transport_fallback,
                "meshlet_source": (f"derived meshlet streams ({per_meshlet:g} B per meshlet evaluated"
                                   f"{' (alpha classes instead of material indices)' if not args.no_alpha_classes else ''}, "
                                   f"commands derived from the 2-B count chain"
                                   f"{'; the entity stage reads a 32-B side entry per mesh instead of its 128-B MeshInfo' if not args.no_mesh_table else ''}; built once from the meshlet buffer "
                                   f"before the timed region in {mstream_build_ms:.2f} ms, +36.65 B of HBM per meshlet)"
                                   if mstream is not None
                                   else "the 32-B Meshlet buffer"),
            },
            "roofline": {
                "bound": "hbm", "kernel": "meshlet_eval_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": (f"profiles/{traffic_file(args.meshlet_source)} (rocprofv3 --pmc passes of an earlier run of this workload, "
                                   "tools/traffic.sh; not measured in this run)" if traffic is not None else None),
                # is that file about THIS library?  (sha256 of liborbit_cull.so stamped into it by tools/traffic.py)
                "traffic_current": (traffic_stamp == lib_sha) if traffic is not None else None,
                "library_sha256": lib_sha,
                "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_meshlet": per_meshlet,
                "avg_launch_ms": eval_ms, "launches_timed": eval_launches,
                "box_stream_reference": stream_ref,
                "op": {"name": "meshlet_cull (eval + scan + emit)", "algorithmic_bytes": op_bytes,
                       "avg_ms": kernel_ms, "achieved": op_bytes / (kernel_ms * 1e-3) / 1e9},
            },
            "cpu_baseline": cpu_baseline,
            "protocol_8d": protocol_8d,
            "two_pass": two_pass,
            "transport_alt": transport_alt,
            "single_rank_check": single_rank_check,
            "frame_latency_ms": frame_latency_ms,
            "frame_latency_ms_from_meshlet_buffer": buffer_path_latency_ms,
            "ms_per_step_from_meshlet_buffer": buffer_path_ms_per_step,
            "per_rank": dict(per_rank_rows, cull_ms_max=kernel_ms_max_frame, exchange_ms_max=exchange_ms_max,
                             what="one frame alone on each rank's device, per rank: the cull of its shard (entity + "
                                  "meshlet stage), the exchange of its list.  In the timed region they overlap (cull of "
                                  "step i+1, wire of step i): step time = the slower stage, not the sum"),
            "cull_only_meshlets_per_sec": tot_in / (kernel_ms_max * 1e-3) if kernel_ms_max > 0 else None,
            "scene_gen_s": gen_s,
        }
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    for x in exchanges:
        x.close()
    for s in slots:
        s.eng.close()
    if use_dist:
        dist.destroy_process_group()


def run_two_pass(args, spec, slot, draws, mesh, ent, meshlets_base, materials, disp_cap, draw_cap, dev, L):
    """The bench scene through the full two-pass frame of the renderer (forward.rs:286-298,391-403): pass-1 cull (last
    frame's bits) -> depth_reduce (1920x1080) -> pass-2 cull, steady state (frames >= 3), one HIP event pair per frame,
    the median.  Depth: a wall 950 m in front of the camera over 60 % of the screen (tools/bench_c5_two_pass.py, which
    also checks the frame against the oracle; tests/test_gpu_full_size.py does so in the GPU suite)."""
    from orbit_amd import camera
    from orbit_amd.engine import depth_pyramid_desc

    eng = slot.eng
    E, M = spec.entities, spec.meshlets_per_entity
    W, H = 1920, 1080
    from orbit_amd import passes

    z_near = 0.01  # the renderer's camera (src/app.rs:744-752), as camera.frame_cull_info builds it
    depth = torch.zeros(H, W, dtype=torch.float32, device=dev)
    depth[:, : int(0.6 * W)] = z_near / 950.0
    d = depth_pyramid_desc(W, H)
    pyr = torch.zeros(d.total_texels, dtype=torch.float32, device=dev)
    ps = (d.width, d.height)
    draw1 = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev)
    draw2 = torch.zeros_like(draw1)
    evis = torch.zeros((E + 31) // 32, dtype=torch.int32, device=dev)
    mvis = torch.zeros(E * spec.vis_words_per_entity, dtype=torch.int32, device=dev)
    st = slot.stream
    # OcclusionCullInfo::VisibilityRead / VisibilityWrite through the host mirror (forward.rs:266-298, 371-403)
    ci1 = camera.frame_cull_info((0.0, 0.0, 1300.0), n_planes=args.planes, occlusion_kind=passes.VISIBILITY_READ,
                                 visibility_buffer=evis, meshlet_visibility_buffer=mvis)
    ci2 = camera.frame_cull_info((0.0, 0.0, 1300.0), n_planes=args.planes, occlusion_kind=passes.VISIBILITY_WRITE,
                                 visibility_buffer=evis, meshlet_visibility_buffer=mvis, depth_pyramid=pyr,
                                 depth_pyramid_size=ps)

    def frame():
        eng.entity_cull(ci1, draws, mesh, slot.disp, ent, E, disp_cap, visibility_buffer=evis, stream=st)
        eng.meshlet_cull(ci1, slot.disp, meshlets_base, draw1, ent, materials, disp_cap, draw_cap,
                         meshlet_visibility_buffer=mvis, material_count=spec.materials, stream=st)
        eng.depth_reduce(depth, W, H, pyr, stream=st)
        eng.entity_cull(ci2, draws, mesh, slot.disp, ent, E, disp_cap, visibility_buffer=evis, depth_pyramid=pyr,
                        depth_pyramid_size=ps, stream=st)
        eng.meshlet_cull(ci2, slot.disp, meshlets_base, draw2, ent, materials, disp_cap, draw_cap,
                         meshlet_visibility_buffer=mvis, depth_pyramid=pyr, depth_pyramid_size=ps,
                         material_count=spec.materials, stream=st)

    ts = []
    for f in range(11):
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        frame()
        z.record(st)
        torch.cuda.synchronize()
        if f >= 3:
            ts.append(a.elapsed_time(z))
    eng.status()
    ms = float(np.median(ts))
    return {"what": "pass-1 cull -> depth_reduce 1920x1080 -> pass-2 cull of the same scene (entity_cull + meshlet_cull "
                    "each), steady state, median of 8 frames each inside one HIP event pair; wall at 950 m over 60 % of the "
                    "screen",
            "frame_ms": ms, "frame_ms_min": float(min(ts)), "frame_ms_max": float(max(ts)),
            "meshlets_per_sec": E * M / (ms * 1e-3),
            "pass1_draws": int(draw1[:4].view(torch.int32).item()), "pass2_draws": int(draw2[:4].view(torch.int32).item()),
            "visible_meshlets": int(torch.count_nonzero((mvis.view(torch.uint8).unsqueeze(1) >> torch.arange(
                8, device=dev, dtype=torch.uint8)) & 1).item())}


def dry_ranks(args, rank, world, json_fd):
    """Launcher test (CPU): rendezvous, shard ranges, one reduction, rank 0's JSON line."""
    from orbit_amd import dist as odist

    if world > 1:
        dist.init_process_group("gloo")
    b, e = odist.shard_ranges(args.entities, world)[rank]
    mine = torch.tensor([b, e], dtype=torch.int64)
    allr = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    if world > 1:
        dist.all_gather(allr, mine)
        dist.barrier()
    else:
        allr = [mine]
    if rank == 0:
        out = {"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "shards": [[int(t[0]), int(t[1])] for t in allr]}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


def traffic_file(meshlet_source):
    return "traffic.json" if meshlet_source == "buffer" else "traffic_stream.json"


def lib_sha256():
    import hashlib

    try:
        with open(os.path.join(ROOT, "orbit_amd", "liborbit_cull.so"), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()
    except OSError:
        return None


def load_traffic(entities, meshlets_per_entity, meshlet_source):
    """(HBM bytes per launch of the dominant kernel, sha256 of the library they were measured on) from the committed
    rocprofv3 PMC passes (profiles/traffic.json — Meshlet buffer — or profiles/traffic_stream.json — derived streams —
    written by tools/traffic.py); (None, None) when absent or for another workload."""
    path = os.path.join(ROOT, "profiles", traffic_file(meshlet_source))
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("entities") == entities and t.get("meshlets_per_entity") == meshlets_per_entity:
            return t.get("hbm_bytes_per_launch"), t.get("library_sha256")
    except (OSError, ValueError):
        pass
    return None, None


def run_cpu_baseline(args, spec, ci, draws, mesh, ent, meshlets, materials, disp, draw, L):
    """Times the CPU oracle on all host cores (oracle/orbit_oracle_mt.c: the scalar C port on contiguous shards,
    one OpenMP thread each; built here with -O3 -march=native -ffp-contract=off) on the first
    `cpu_sample_entities` of the same scene (default: all of them), and REQUIRES the GPU's output for those entities
    to be bit-identical (canonical order: the sample's records / commands are a prefix of the full-scene output).
    A single-thread pass over 1/16 of the sample is reported beside."""
    from oracle import oracle

    oracle.build()
    M = spec.meshlets_per_entity
    es = min(args.cpu_sample_entities, spec.entities)
    if es <= 0:
        return None
    threads = oracle.max_threads()
    h_draws, h_mesh, h_ent = draws.cpu().numpy(), mesh.cpu().numpy(), ent.cpu().numpy()
    h_meshlets = meshlets[:es * M * 32].cpu().numpy()
    h_mat = materials.cpu().numpy()
    cap_d, cap_c = es * spec.records_per_entity + 8, es * M // 2 + 1024
    cpu_s, reps, out = 0.0, 0, None
    while (cpu_s < args.cpu_seconds / 4 or reps < 3) and reps < 200:
        t0 = time.perf_counter()
        odisp, odraw, _, _, _ = oracle.cull_frame_mt(ci, h_draws, es, h_mesh, h_ent, h_meshlets, h_mat, cap_d, cap_c,
                                                     threads, native=True, out=out)
        cpu_s += time.perf_counter() - t0
        out = (odisp, odraw)
        reps += 1
    n_rec = int(odisp[:4].view(np.uint32)[0])
    n_in = int(odisp[12:12 + 16 * n_rec].view(np.uint32).reshape(-1, 4)[:, 2].sum())
    n_draw = int(odraw[:4].view(np.uint32)[0])
    # canonical order => the GPU's full-scene output starts with exactly these records / commands
    g_disp = disp[12:12 + 16 * n_rec].cpu().numpy()
    g_draw = draw[4:4 + 28 * n_draw].cpu().numpy()
    same = bool(np.array_equal(g_disp, odisp[12:12 + 16 * n_rec]) and np.array_equal(g_draw, odraw[4:4 + 28 * n_draw]))
    if es == spec.entities:  # whole scene: the counts must agree as well
        same = same and n_rec == int(disp[:4].view(torch.int32).item()) and n_draw == int(draw[:4].view(torch.int32).item())
    if not same:
        raise SystemExit("bench.py: GPU output differs from the CPU oracle on the bench scene — no metric is reported")
    # one thread, 1/16 of the sample
    e1 = max(256, es // 16 // 256 * 256)
    t0 = time.perf_counter()
    d1, _, _, _, _ = oracle.cull_frame_mt(ci, h_draws, e1, h_mesh, h_ent, h_meshlets, h_mat,
                                          e1 * spec.records_per_entity + 8, e1 * M // 2 + 1024, 1, native=True)
    t1 = time.perf_counter() - t0
    n1 = int(d1[12:12 + 16 * int(d1[:4].view(np.uint32)[0])].view(np.uint32).reshape(-1, 4)[:, 2].sum())
    return {
        "value": n_in * reps / cpu_s, "unit": "meshlets/s", "cores": threads, "kind": "port",
        "sample": f"{'all' if es == spec.entities else 'first'} {es} of {spec.entities} entities ({n_in} meshlets into "
                  f"meshlet_cull) of the same scene, {reps} frames of oracle/orbit_oracle.c entity_cull + meshlet_cull "
                  f"on {threads} OpenMP threads (contiguous shards, oracle/orbit_oracle_mt.c) in {cpu_s:.2f} s wall on a "
                  f"{os.cpu_count()}-core host; gcc {oracle.NATIVE_CFLAGS}",
        "flags": oracle.NATIVE_CFLAGS,
        "single_thread_value": n1 / t1,
        "gpu_output_bit_identical": same,
    }


if __name__ == "__main__":
    main()
