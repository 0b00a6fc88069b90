"""depth_reduce at 1920x1080 hop by hop: wall-clock stamps (100 MHz) of the workgroup that finishes last among the tile
workgroups — the one that also reduces the tail — from a -DORBIT_TRIAGE_STAMPS build of depth_reduce.hip
(bash tools/mkvariant_any.sh dstamps depth_reduce -DORBIT_TRIAGE_STAMPS), beside the product library's launch timed by
events in the same process.  usage (GPU box): python tools/depth_stamps.py [out.json]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import config_scenes as cs
from orbit_amd import _lib
from orbit_amd.engine import Engine, depth_pyramid_desc

path = os.path.join(ROOT, "tools", "variants", "dstamps.so")
if not os.path.exists(path):
    sys.exit(f"{path} is missing: bash tools/mkvariant_any.sh dstamps depth_reduce -DORBIT_TRIAGE_STAMPS")
lib = _lib.load_variant(path)
lib.orbit_debug_read_depth_stamps.restype = C.c_int32
lib.orbit_debug_read_depth_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
W, H = cs.SCREEN
depth = torch.from_numpy(cs.config3_depth(cs.camera()).reshape(-1).copy()).cuda()
d = depth_pyramid_desc(W, H)
rows = []
for name, library in (("stamps", lib), ("product", None)):
    eng = Engine(0, _library=library)
    pyr = torch.zeros(d.total_texels, dtype=torch.float32, device="cuda")
    for _ in range(5):
        eng.depth_reduce(depth, W, H, pyr)
    torch.cuda.synchronize()
    ev = []
    for _ in range(40):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.depth_reduce(depth, W, H, pyr); b.record(); torch.cuda.synchronize()
        ev.append(a.elapsed_time(b) * 1e3)
        if library is not None:
            st = (C.c_ulonglong * 7)()
            assert lib.orbit_debug_read_depth_stamps(eng._ctx, st) == 0
            rows.append([v / 100.0 for v in st])  # us
    if library is None:
        product_us = float(np.median(ev))
    else:
        stamps_build_us = float(np.median(ev))
    eng.close()
r = np.array(rows)
hop = lambda a, b: round(float(np.median(r[:, b] - r[:, a])), 2)
out = {"workload": f"depth_reduce {W}x{H} -> {d.width}x{d.height}, {d.mip_levels} levels", "launch_by_events_us": {"product": round(product_us, 2), "stamps_build": round(stamps_build_us, 2)},
       "last_workgroup_hops_us": {
           "first workgroup's start -> the last one's start (dispatch of the grid)": hop(0, 1),
           "start -> levels 0..5 (level 0: four footprints of four loads per thread; 1..3 by lane exchanges, 4 and 5 by one wave behind one barrier)": hop(1, 2),
           "(a tile that is not whole inside its levels: levels 1..5 through LDS)": hop(2, 3),
           "its stores drained (level 5 written through)": hop(3, 4),
           "two arrival tickets (row, then rows)": hop(4, 5),
           "tail: level 5 loaded (sc1), levels 6..10 by one wave (registers, lane exchanges), stored, drained": hop(5, 6),
           "first workgroup's start -> tail done": hop(0, 6)},
       "note": "what the events see beyond the last line is the launch itself (tools/ubench/hop.hip: an empty launch between two events takes ~6 us)"}
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
