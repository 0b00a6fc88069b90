"""bench.py's output contract on a small scene (a child process, as the driver runs it): ONE JSON line on stdout with
the metric, the roofline object of the dominant kernel and the CPU baseline — from both meshlet sources."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("source", ["stream", "buffer"])
def test_bench_line_has_the_contract_fields(source):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                        "--entities", "6000", "--cpu-sample-entities", "6000", "--cpu-seconds", "0.5",
                        "--settle-ms", "5", "--meshlet-source", source],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # libraries' banners go to stderr
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "meshlets/s" and d["value"] > 0 and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["meshlets_into_meshlet_cull"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    per = 20.25 if source == "stream" else 32  # the stream source with its alpha classes (bench.py default)
    assert rf["algorithmic_bytes_per_meshlet"] == per and rf["launches_timed"] >= 1
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert rf["traffic"] is None  # the committed PMC passes are of the full-size scene only
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "meshlets/s" and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["gpu_output_bit_identical"] is True and "sample" in cb
    assert ("derived meshlet streams" in d["config"]["meshlet_source"]) == (source == "stream")
    p8 = d["protocol_8d"]  # SURVEY.md §8d literally: Meshlet buffer, 5 warm-ups, >= 20 event-timed steps, median
    assert p8["warmup"] == 5 and p8["steps"] >= 20 and p8["ms_per_step_min"] <= p8["ms_per_step_median"] <= p8["ms_per_step_max"]
    # every frame the process ran before the timed region: eight cold ones, the settle's (5 ms here), the W asked for
    assert d["config"]["warmup_effective"] == 8 + d["config"]["settle_frames"] + d["warmup"] and d["config"]["settle_frames"] >= 8
    assert d["config"]["settle_with"] == "frames" and d["config"]["ms_per_step_without_settle"] > 0
    assert len(d["roofline"]["library_sha256"]) == 64 and d["roofline"]["traffic_current"] is None
    tp = d["two_pass"]  # config 5's two-pass frame as the secondary figure
    assert tp["frame_ms_min"] <= tp["frame_ms"] <= tp["frame_ms_max"] and tp["meshlets_per_sec"] > 0
    assert tp["pass1_draws"] + tp["pass2_draws"] > 0 and tp["visible_meshlets"] > 0


def _run_ranks(extra, env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--entities", "6000",
                        "--settle-ms", "5"] + extra, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ORBIT_BENCH_TIMEOUT_S="600", **env))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("expand", ["own", "none"])
def test_bench_two_ranks_on_one_device_whole_device_path(expand):
    """`bench.py --gpus 2` as the driver starts it (bench.py spawns its ranks), both ranks on the box's one GPU: the
    real exchange — HIP IPC mappings of the peers' buffers and control blocks, the scatter kernel, counts and completion
    signalled on the device (the process group over gloo: RCCL cannot put two ranks on one device) — then every rank's
    segment check, the check of its own command buffer and the cross-rank checksum of the gathered list inside bench.py
    (it exits non-zero if they fail); behind the timed region the OTHER transport's leg (the all-gather of fixed-capacity
    segments + compaction), whose list must be the main transport's byte for byte."""
    d = _run_ranks(["--gpus", "2", "--expand", expand], dict(ORBIT_BENCH_ONE_DEVICE="1", ORBIT_BENCH_BACKEND="gloo"))
    c = d["config"]
    assert d["n_gpus"] == 2 and c["world_size"] == 2 and c["gather"] is True and c["transport"] == "p2p"
    assert c["exchange"] == "records" and c["backend"] == "gloo" and c["signalling"] == "device"
    assert c["transport_fallback"] is None and c["step_intervals"]["intervals"] >= 2
    assert c["expand"] == expand and ("own shard" in c["step_product"]) == (expand == "own")
    pr = d["per_rank"]
    assert len(pr["cull_ms"]) == 2 and len(pr["exchange_ms"]) == 2 and all(v > 0 for v in pr["cull_ms"] + pr["exchange_ms"])
    assert sum(pr["meshlets_in"]) == c["meshlets_into_meshlet_cull"] == 6000 * 256
    assert sum(pr["survivors"]) == c["survivors"] > 0
    assert abs(d["value"] - c["meshlets_into_meshlet_cull"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    ta = d["transport_alt"]
    assert ta["name"] == "rccl" and ta["signalling"] == "rccl-allgather" and ta["ms_per_step"] > 0
    assert ta["list_identical_to_main_transport"] is True and ta["steps"] == 4


@pytest.mark.parametrize("world,transport", [(8, "p2p"), (5, "p2p"), (8, "rccl")])
def test_bench_at_the_world_size_north_star_names_on_one_device(world, transport):
    """`ORBIT_BENCH_ONE_DEVICE=1 bench.py --gpus 8` (and 5: not a power of two): the whole N > 1 bench path — spawn, shard
    ranges, eight contexts' culls and eight device-signalled exchanges sharing the box's one GPU, two frames in flight —
    at the world size the driver's scaling run ends with.  bench.py checks every rank's segment of the gathered list, its
    own command buffer and the cross-rank checksum; --verify-single-rank additionally holds the gathered list against ONE
    rank culling the whole scene (from the Meshlet buffer: another evaluation path), and the other transport's leg must
    deliver the same bytes."""
    # (with the all-gather as the MAIN transport every step of this one-device configuration stages its fixed-capacity
    # segments through gloo and the host — seconds per step at eight ranks, and no number of this run is a measurement:
    # a third of the entities and no settle phase there, so that the test stays minutes away from the bench's own deadline
    # on a slow box)
    entities = 24000 if transport == "p2p" else 8000
    d = _run_ranks(["--gpus", str(world), "--entities", str(entities), "--transport", transport, "--verify-single-rank",
                    "--cpu-sample-entities", "0"] + ([] if transport == "p2p" else ["--settle-ms", "0"]),
                   dict(ORBIT_BENCH_ONE_DEVICE="1", ORBIT_BENCH_BACKEND="gloo"))
    c = d["config"]
    assert d["n_gpus"] == world and c["world_size"] == world and c["transport"] == transport and c["transport_fallback"] is None
    assert c["signalling"] == ("device" if transport == "p2p" else "rccl-allgather")
    pr = d["per_rank"]
    assert len(pr["cull_ms"]) == world and sum(pr["meshlets_in"]) == entities * 256 and sum(pr["survivors"]) == c["survivors"] > 0
    assert d["single_rank_check"]["gathered_list_equals_single_rank_list"] is True
    assert d["single_rank_check"]["records"] == entities * 8
    ta = d["transport_alt"]
    assert ta["name"] == ("rccl" if transport == "p2p" else "p2p") and ta["list_identical_to_main_transport"] is True


def test_bench_two_ranks_on_one_device_at_full_size():
    """The same with the whole 50 M-meshlet scene: two processes whose device-filling launches share the GPU.  A launch
    that hands work over inside itself must not depend on a workgroup that has not been dispatched — with two such
    launches resident each can hold the slots the other's first workgroups need (round 4: the chain emit's scan duty
    did, both ranks ran into the poll bound, ORBIT_E_TIMEOUT, until a waiting workgroup learned to do a missing chunk
    itself: meshlet_emit.hip emit_scan_wait).  bench.py checks every rank's outputs and the device status."""
    d = _run_ranks(["--gpus", "2", "--entities", "195313", "--steps", "6", "--cpu-sample-entities", "0"],
                   dict(ORBIT_BENCH_ONE_DEVICE="1", ORBIT_BENCH_BACKEND="gloo"))
    c = d["config"]
    assert d["n_gpus"] == 2 and c["transport"] == "p2p" and c["signalling"] == "device" and c["transport_fallback"] is None
    assert sum(d["per_rank"]["meshlets_in"]) == 195313 * 256 and sum(d["per_rank"]["survivors"]) == c["survivors"] > 0


def test_bench_rccl_transport_with_a_world_of_one():
    """The RCCL transport through a real communicator — of one rank, all this box can hold: the record list as ONE
    all-gather of fixed-capacity segments + the device-side compaction (north_star's transport, nothing waits for the
    host)."""
    d = _run_ranks(["--gpus", "1", "--transport", "rccl", "--cpu-sample-entities", "0"], dict(ORBIT_BENCH_FORCE_DIST="1"))
    c = d["config"]
    assert c["gather"] is True and c["transport"] == "rccl" and c["world_size"] == 1 and c["backend"] == "nccl"
    assert c["signalling"] == "rccl-allgather" and "all-gather of fixed-capacity" in c["workload"]
    assert d["per_rank"]["exchange_ms"][0] > 0
    ta = d["transport_alt"]  # ... and the p2p exchange as the other transport's leg, with a world of one
    assert ta["name"] == "p2p" and ta["signalling"] == "device" and ta["list_identical_to_main_transport"] is True


@pytest.mark.parametrize("extra", [["--graph", "--frames-in-flight", "1"], ["--frames-in-flight", "1"],
                                   ["--frames-in-flight", "2", "--expand", "none"]])
def test_bench_two_ranks_step_variants(extra):
    """The N > 1 step's other shapes, two ranks on the box's one GPU: every timed step replayed as a captured graph (cull
    beside the previous list's exchange, fork and join inside the capture), strictly serial frames, two frames in flight
    without the command buffer.  bench.py checks every rank's lists."""
    d = _run_ranks(["--gpus", "2"] + extra, dict(ORBIT_BENCH_ONE_DEVICE="1", ORBIT_BENCH_BACKEND="gloo"))
    c = d["config"]
    assert d["n_gpus"] == 2 and c["transport"] == "p2p" and c["signalling"] == "device" and c["transport_fallback"] is None
    assert c["steps_as_graph"] == ("--graph" in extra)
    assert c["frames_in_flight"] == (1 if "1" in extra else 2)
    assert c["host_enqueue_ms_per_step"] > 0 and c["step_intervals"]["intervals"] >= 2
    assert sum(d["per_rank"]["meshlets_in"]) == 6000 * 256 and sum(d["per_rank"]["survivors"]) == c["survivors"] > 0


def test_bench_under_torch_distributed_run_as_the_driver_starts_it():
    """The driver's own launch line for N > 1 — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...` — with both ranks on the box's one GPU: bench.py
    must take RANK / LOCAL_RANK / WORLD_SIZE from the launcher (not start ranks of its own), and rank 0 alone prints
    the ONE JSON line."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "4", "--warmup", "2", "--entities", "6000", "--settle-ms", "5"],
                       capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ORBIT_BENCH_ONE_DEVICE="1", ORBIT_BENCH_BACKEND="gloo", ORBIT_BENCH_TIMEOUT_S="600"))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world_size"] == 2 and d["steps"] == 4 and d["warmup"] == 2
    assert sum(d["per_rank"]["meshlets_in"]) == 6000 * 256 and d["value"] > 0
