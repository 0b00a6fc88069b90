"""What a fork and a join cost the stream they sit in (config 4's cluster chain on the caller's stream, config 3's 1080p
pyramid as the forked work): the chain alone; with an event recorded in front of it that nobody waits for; with a side
stream waiting for that event and running the pyramid, not joined; joined; a join on an event that completed long ago.
HIP-event medians of 30 after 5 warm-ups, eager and as replayed graphs.  GPU box, repo root: python tools/fork_join_cost.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import config_scenes as cs
from oracle import oracle
from orbit_amd.engine import Engine, depth_pyramid_desc


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()


cam = cs.camera()
W, H = cs.SCREEN
depth = cs.config3_depth(cam)
push, info, lights = cs.config4_inputs(oracle, cam)
cx, cy, cz = cs.CLUSTERS
total = cx * cy * cz
cap, lcap = cx * cy * max(4, cz), total * 32
eng = Engine(0, max_lights=12_000, max_clusters=total)
gdepth, glights = dev(depth).view(torch.float32), dev(lights)
gm = torch.zeros(cx * cy, dtype=torch.int32, device="cuda")
gb = torch.zeros((total, 2), dtype=torch.int32, device="cuda")
gu = torch.zeros(16 + 4 * cap, dtype=torch.uint8, device="cuda")
gl = torch.zeros(4 + 4 * lcap, dtype=torch.uint8, device="cuda")
gimg = torch.zeros((total, 2), dtype=torch.int32, device="cuda")
d = depth_pyramid_desc(W, H)
pyr = torch.zeros(d.total_texels, dtype=torch.float32, device="cuda")
filler = torch.zeros(1 << 20, dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
ev0, ev1, ev_old = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()


def chain():
    eng.compute_clusters(push, info, gdepth, glights, gm, gb, gu, cap, gl, lcap, gimg)


def pre():  # something in the stream in front (hides the first launch's latency, as a frame's earlier work does)
    filler.add_(1.0)
    filler.add_(1.0)


def v_alone():
    pre(); chain()


def v_record():
    pre(); ev0.record(); chain()


def v_fork_only():
    pre(); ev0.record()
    with torch.cuda.stream(side):
        side.wait_event(ev0)
        eng.depth_reduce(gdepth, W, H, pyr)
    chain()


def v_fork_join():
    pre(); ev0.record()
    with torch.cuda.stream(side):
        side.wait_event(ev0)
        eng.depth_reduce(gdepth, W, H, pyr)
        ev1.record()
    chain()
    torch.cuda.current_stream().wait_event(ev1)


def v_join_old():
    pre(); chain()
    torch.cuda.current_stream().wait_event(ev_old)


def v_serial():
    pre(); eng.depth_reduce(gdepth, W, H, pyr); chain()


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return round(float(np.median(ts)), 1)


with torch.cuda.stream(side):
    ev_old.record()
torch.cuda.synchronize()
variants = dict(alone=v_alone, event_recorded_in_front=v_record, forked_not_joined=v_fork_only, forked_and_joined=v_fork_join,
                join_on_an_old_event=v_join_old, serial=v_serial)
out = {"eager_us": {}, "graph_us": {}, "pre_us": timeit(pre)}
for name, fn in variants.items():
    out["eager_us"][name] = timeit(fn)
for name, fn in variants.items():
    if name in ("join_on_an_old_event", "forked_not_joined"):  # (an unjoined fork cannot be captured either)
        continue  # an event recorded outside the capture cannot be waited for inside it
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    out["graph_us"][name] = timeit(g.replay)
    del g
    torch.cuda.synchronize()
print(json.dumps(out))
