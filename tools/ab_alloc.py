"""Where the caller's 1.6 GB meshlet buffer comes from, against the strict drop-in's evaluation
(meshlet_eval_kernel<0,0,false,false,false>): the same bytes in memory from hipMalloc (through torch's allocator), from a
fresh hipMalloc, and from the virtual-memory API (hipMemCreate + hipMemMap at the recommended granularity) — the two
allocators a Vulkan-interop host can choose between for the imported buffer.  Also: N engines alive at once on ONE buffer
(each with its own scratch arena), interleaved — is an arena's speed a stable property of the arena?  GPU box, repo root:
`for i in 1 2 3; do python tools/ab_alloc.py; done`."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from orbit_amd import camera, layouts as L, synth
from orbit_amd.engine import Engine

REPS = int(os.environ.get("REPS", 9))
dev = torch.device("cuda", 0)
spec = synth.C5Spec()
E, M = spec.entities, spec.meshlets_per_entity
draws, mesh, ent, half = synth.gen_entity_tables(spec, dev)
meshlets = synth.gen_meshlets(spec, 0, E, dev, half)
materials = synth.gen_materials(spec, dev)
ci = camera.frame_cull_info((0.0, 0.0, 1300.0))
disp_cap, draw_cap = E * spec.records_per_entity + 8, E * M // 2 + 1024
disp = torch.zeros(L.DISPATCH_HEADER + 16 * disp_cap, dtype=torch.uint8, device=dev)
draw = torch.zeros(L.DRAW_HEADER + 28 * draw_cap, dtype=torch.uint8, device=dev)
nbytes = meshlets.numel()
hip = C.CDLL("libamdhip64.so")


def timed(eng, buf):
    out = []
    for _ in range(REPS):
        eng.profile(True)
        eng.meshlet_cull(ci, disp, buf, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
        torch.cuda.synchronize()
        out.append(eng.profile_read()[0])
    return round(float(np.median(out[2:]) * 1e3), 1)


def raw_malloc(n):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), C.c_size_t(n)) == 0
    return p.value


class Prop(C.Structure):  # hipMemAllocationProp
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("loc_type", C.c_int), ("loc_id", C.c_int),
                ("win32", C.c_void_p), ("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]


class Access(C.Structure):  # hipMemAccessDesc
    _fields_ = [("loc_type", C.c_int), ("loc_id", C.c_int), ("flags", C.c_int)]


def vmm_alloc(n):
    prop = Prop(type=1, requestedHandleType=0, loc_type=1, loc_id=0)  # pinned, device 0
    gran = C.c_size_t()
    rc = hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), 1)  # recommended
    assert rc == 0, rc
    size = (n + gran.value - 1) // gran.value * gran.value
    h = C.c_void_p()
    assert hip.hipMemCreate(C.byref(h), C.c_size_t(size), C.byref(prop), C.c_ulonglong(0)) == 0
    va = C.c_void_p()
    assert hip.hipMemAddressReserve(C.byref(va), C.c_size_t(size), C.c_size_t(gran.value), None, C.c_ulonglong(0)) == 0
    assert hip.hipMemMap(va, C.c_size_t(size), C.c_size_t(0), h, C.c_ulonglong(0)) == 0
    acc = Access(loc_type=1, loc_id=0, flags=3)
    assert hip.hipMemSetAccess(va, C.c_size_t(size), C.byref(acc), C.c_size_t(1)) == 0
    return va.value, gran.value


def copy_to(ptr):
    assert hip.hipMemcpy(C.c_void_p(ptr), C.c_void_p(meshlets.data_ptr()), C.c_size_t(nbytes), 3) == 0  # device to device
    torch.cuda.synchronize()


N = int(os.environ.get("ENGINES", 5))
engs = [Engine(0, max_entities=E + 256, max_dispatches=disp_cap, max_draws=draw_cap) for _ in range(N)]
engs[0].entity_cull(ci, draws, mesh, disp, ent, E, disp_cap)
torch.cuda.synchronize()
for _ in range(400):  # settle
    engs[0].meshlet_cull(ci, disp, meshlets, draw, ent, materials, disp_cap, draw_cap, material_count=spec.materials)
torch.cuda.synchronize()
rows = {"engines_on_torch_buffer": [[timed(e, meshlets) for e in engs] for _ in range(3)]}
bufs = {"torch": meshlets.data_ptr()}
bufs["hipMalloc"] = raw_malloc(nbytes)
copy_to(bufs["hipMalloc"])
try:
    bufs["vmm"], gran = vmm_alloc(nbytes)
    rows["vmm_granularity"] = gran
    copy_to(bufs["vmm"])
except AssertionError as ex:
    rows["vmm_error"] = repr(ex)
rows["buffers"] = {k: {"addr_mod_1GiB": hex(p % (1 << 30)), "eval_us": [[timed(e, p) for e in engs[:3]] for _ in range(2)]}
                   for k, p in bufs.items()}
print(json.dumps(rows))
