"""The shard launch's phases from three stamped runs (triage build; tools/wave_tail_shard.py with EXTRA_FLAGS 2048 / 4096 / 0):
per workgroup, relative to the launch's first wave — entry, its entity chunks done (workgroups that took chunks), tiles
begin (all chunks in, first rows requested), end.  usage (GPU box): E=24414 python tools/shard_phases.py"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.environ.get("E", "24414")
st = {}
for name, flags in (("enter", 2048), ("chunks", 4096), ("tiles", 0)):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wave_tail_shard.py")], check=True, stdout=subprocess.DEVNULL,
                   env=dict(os.environ, E=E, STREAM="1", EXTRA_FLAGS=str(flags)))
    st[name] = np.load(os.path.join(ROOT, "gpurun_out", "wave_stamps_shard.npy")).astype(np.int64)
q = lambda a: " ".join(f"{np.percentile(a, p):6.1f}" for p in (0, 10, 50, 90, 100))
n_chunks = (int(E) + 255) // 256
blk = np.arange(4096) // 4
part = blk < n_chunks
for name in ("enter", "chunks", "tiles"):
    s = st[name]
    t0 = st[name][:, 1].max()  # every run's own clock: the launch's end is the common reference
    b, e = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0
    print(f"{name:7s} stamp, us before the launch's last wave ends (p0 p10 p50 p90 p100): all {q(b)} | workgroups with chunks {q(b[part])}")
    if name == "tiles":
        print(f"        end of waves                                                     : all {q(e)}")
