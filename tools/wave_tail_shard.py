"""perf triage: the shard launch (orbit_cull_shard: entity stage inside the evaluation launch) — when do its waves start their tiles and finish?  (ORBIT_SP_DEBUG=8: phase-stamped build,
per-wave {begin, end} wall_clock64 stamps read with the non-public orbit_debug_read_wave_stamps; 100 MHz ticks)"""
import ctypes as C, os, sys
os.environ["ORBIT_SP_DEBUG"] = str(16 | int(os.environ.get("EXTRA_FLAGS", "0")))  # 16: per-wave wall-clock stamps  # 8: stamped build, 16: wall-clock stamps only (no per-phase waits)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _triage; _triage.use_triage_library()  # -DORBIT_TRIAGE build (bash tools/mktriage.sh)
import torch, numpy as np
from orbit_amd import camera, synth, _lib
from orbit_amd.engine import Engine
spec = synth.C5Spec(entities=int(os.environ.get("E", 195_313)))  # E=24414: a rank's shard at N = 8
dev = torch.device("cuda", 0)
draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
ml = synth.gen_meshlets(spec, 0, spec.entities, dev, half)
mat = synth.gen_materials(spec, dev)
ci = camera.frame_cull_info((0.0, 0.0, 1300.0))
E = spec.entities; dc = E * 8 + 8; cc = E * 128
eng = Engine(0, max_entities=E + 256, max_dispatches=dc, max_draws=cc)
disp = torch.zeros(12 + 16 * dc, dtype=torch.uint8, device=dev); draw = torch.zeros(4 + 28 * cc, dtype=torch.uint8, device=dev)
if os.environ.get("STREAM", "0") == "1":  # from the derived streams with alpha classes, as bench.py culls
    ms = eng.meshlet_stream(ml, 0, E * spec.meshlets_per_entity); ms.set_materials(mat, spec.materials); eng.bind_meshlet_stream(ms)
lib = _lib.load()
rec = torch.zeros(8 + 12 * dc, dtype=torch.uint8, device=dev)
own = os.environ.get("OWN", "0") == "1"
out = np.zeros(2 * 8192, dtype=np.uint64)
for rep in range(3):
    for _ in range(3):
        eng.cull_shard(ci, draws, mesh, disp, ent, 0, E, dc, ml, mat, rec, dc, draw_commands_buffer=draw if own else None,
                       draw_capacity=cc if own else 0, material_count=spec.materials)
    torch.cuda.synchronize()
    lib.orbit_debug_read_wave_stamps(eng._ctx, out.ctypes.data_as(C.c_void_p))
    st = out.reshape(-1, 2)[:4096].astype(np.int64)
    b, e = (st[:, 0] - st[:, 0].min()) / 100.0, (st[:, 1] - st[:, 0].min()) / 100.0  # us
    q = lambda a: " ".join(f"{np.percentile(a, p):7.1f}" for p in (0, 1, 10, 50, 90, 99, 100))
    print(f"begin us (p0 p1 p10 p50 p90 p99 p100): {q(b)}")
    print(f"end   us (p0 p1 p10 p50 p90 p99 p100): {q(e)}")
    print(f"life  us                             : {q(e - b)}")
    # how many waves are still running as the kernel drains
    for back in (40, 30, 20, 10, 5, 2):
        print(f"   waves alive {back:3d} us before the last one ends: {(e > e.max() - back).sum():5d}", end="")
    print()
    # per XCD (block index round-robins over the 8 XCDs)
    xcd = (np.arange(4096) // 4) % 8
    print("   median end per XCD:", " ".join(f"{np.median(e[xcd == x]):7.1f}" for x in range(8)))
    blk = np.arange(4096) // 4
    print("   median end by block-index octile:", " ".join(f"{np.median(e[(blk >= k * 128) & (blk < (k + 1) * 128)]):7.1f}" for k in range(8)))
    for x in range(8):
        ex = e[xcd == x]
        print(f"   XCD {x}: end p0 {ex.min():7.1f} p10 {np.percentile(ex, 10):7.1f} p50 {np.median(ex):7.1f} p90 {np.percentile(ex, 90):7.1f} p100 {ex.max():7.1f}")
    # waves of one block (same CU): spread inside blocks
    eb = e.reshape(-1, 4)
    print(f"   spread inside a block (max - min of its 4 waves): median {np.median(eb.max(1) - eb.min(1)):6.1f}  p90 {np.percentile(eb.max(1) - eb.min(1), 90):6.1f}")
    if rep == 2:
        np.save(os.path.join(ROOT, "gpurun_out", "wave_stamps_shard.npy"), st)
