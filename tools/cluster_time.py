"""Config 4's cluster chain alone (tests/config_scenes.py: 1920x1080 depth, 10 000 point lights, 240x135x32 clusters), timed
per entry point and as orbit_compute_clusters (eager and as a graph), for the product library and any variants, and the
outputs of every library compared with the first one's byte for byte (tests/test_configs_gpu.py holds them to the oracle).
HIP-event medians of 40 after 5 warm-ups, the libraries interleaved in rounds so that drift hits all of them alike.
usage (GPU box): python tools/cluster_time.py [tools/variants/x.so ...] [--also-1080p-pyramid]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

import config_scenes as cs
from oracle import oracle
from orbit_amd import _lib
from orbit_amd.engine import Engine, depth_pyramid_desc


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()


def once(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


def main():
    libs = [("product", None)] + [(os.path.basename(p), p) for p in sys.argv[1:] if not p.startswith("--")]
    cam = cs.camera()
    W, H = cs.SCREEN
    depth = cs.config3_depth(cam)
    gdepth = dev(depth).view(torch.float32)
    push, info, lights = cs.config4_inputs(oracle, cam)
    cx, cy, cz = cs.CLUSTERS
    total = cx * cy * cz
    cap, lcap = cx * cy * max(4, cz), total * 32
    glights = dev(lights)
    runs = {}
    for name, path in libs:
        eng = Engine(0, _library=None if path is None else _lib.load_variant(path), max_lights=12_000, max_clusters=total)
        b = dict(gm=torch.zeros(cx * cy, dtype=torch.int32, device="cuda"),
                 gb=torch.zeros((total, 2), dtype=torch.int32, device="cuda"),
                 gu=torch.zeros(16 + 4 * cap, dtype=torch.uint8, device="cuda"),
                 gl=torch.zeros(4 + 4 * lcap, dtype=torch.uint8, device="cuda"),
                 gimg=torch.zeros((total, 2), dtype=torch.int32, device="cuda"))
        d = depth_pyramid_desc(W, H)
        pyr = torch.zeros(d.total_texels, dtype=torch.float32, device="cuda")
        fns = {
            "mark": lambda e=eng, b=b: e.cluster_mark(push, gdepth, b["gm"], b["gb"]),
            "compact": lambda e=eng, b=b: e.cluster_compact((cx, cy, cz), b["gm"], b["gu"], cap),
            "assign": lambda e=eng, b=b: e.cluster_assign(info, b["gu"], b["gb"], glights, b["gl"], lcap, b["gimg"]),
            "compute_clusters": lambda e=eng, b=b: e.compute_clusters(push, info, gdepth, glights, b["gm"], b["gb"], b["gu"],
                                                                      cap, b["gl"], lcap, b["gimg"]),
            "depth_reduce_1080p": lambda e=eng, p=pyr: e.depth_reduce(gdepth, W, H, p),
        }
        for f in fns.values():
            for _ in range(5):
                f()
        torch.cuda.synchronize()
        fns["compute_clusters"]()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fns["compute_clusters"]()
        fns["compute_clusters_graph"] = g.replay
        runs[name] = dict(eng=eng, bufs=b, fns=fns, ts={k: [] for k in fns}, graph=g)
    for _ in range(40):
        for name, r in runs.items():
            for k, f in r["fns"].items():
                r["ts"][k].append(once(f))
    out = {}
    first = None
    for name, r in runs.items():
        torch.cuda.synchronize()
        r["eng"].status()
        b = r["bufs"]
        n_active = int(b["gu"][12:16].view(torch.int32).item())
        n_idx = int(b["gl"][:4].view(torch.int32).item())
        act = b["gu"][16:16 + 4 * n_active].view(torch.int32).long()
        sig = (b["gm"].cpu(), b["gb"].cpu(), b["gu"][:16 + 4 * n_active].cpu(), b["gl"][:4 + 4 * n_idx].cpu(), b["gimg"][act].cpu())
        same = True if first is None else all(torch.equal(x, y) for x, y in zip(sig, first))
        first = sig if first is None else first
        out[name] = {"us": {k: round(float(np.median(v)) * 1e3, 2) for k, v in r["ts"].items()},
                     "min_us": {k: round(float(min(v)) * 1e3, 2) for k, v in r["ts"].items()},
                     "active_clusters": n_active, "light_indices": n_idx, "same_bytes_as_first": bool(same)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
