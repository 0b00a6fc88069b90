"""Device-side synthetic scene generator for BASELINE.json config 5 (50 M meshlets).

The reference ships no large scene (its commented-out stress generator clones one
prefab 200 000 times, src/app.rs:655-701) and the 50 M-meshlet scene of SURVEY.md
§8d is synthetic by definition.  The generator is counter based (splitmix64 of
(seed, stream, index)), written with torch integer/float32 ops so that it runs on
the GPU that will cull the data: every rank can materialise exactly its shard,
and the bytes it produces are the *input* of both the HIP engine and (downloaded)
the CPU oracle — no second generator has to agree with it.

Layouts: orbit_amd/layouts.py (= shaders/include/types.glsl of the reference).
"""
import math

import torch

_MASK = (1 << 64) - 1


def _s64(c):
    c &= _MASK
    return c - (1 << 64) if c >= (1 << 63) else c


_G = _s64(0x9E3779B97F4A7C15)
_C1 = _s64(0xBF58476D1CE4E5B9)
_C2 = _s64(0x94D049BB133111EB)
_K = _s64(0xD1B54A32D192ED03)


def _lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def _mix(z):
    z = z + _G
    z = (z ^ _lsr(z, 30)) * _C1
    z = (z ^ _lsr(z, 27)) * _C2
    return z ^ _lsr(z, 31)


def _mix_scalar(x):
    x = (x + 0x9E3779B97F4A7C15) & _MASK
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _MASK
    return x ^ (x >> 31)


def rnd_u64(seed, stream, idx):
    base = _s64(_mix_scalar((seed * 0x2545F4914F6CDD1D + stream) & _MASK))
    return _mix(idx * _K + base)


def rnd_f32(seed, stream, idx):
    """Uniform float32 in [0, 1): top 24 bits times 2^-24 (exact)."""
    return _lsr(rnd_u64(seed, stream, idx), 40).to(torch.float32) * (2.0 ** -24)


def rnd_int(seed, stream, idx, lo, hi):
    return _lsr(rnd_u64(seed, stream, idx), 1) % (hi - lo + 1) + lo


class C5Spec:
    """E entities, each its own mesh of M meshlets, on a
    jittered lattice filling +-extent; every entity is inside the view frustum so
    that meshlet_cull streams all E*M meshlets; cone cutoffs are tuned so that
    about 10 % of the meshlets survive (the survivor ratio SURVEY.md §8d prices)."""

    def __init__(self, entities=195_313, meshlets_per_entity=256, extent=512.0, seed=69, materials=25):
        self.entities = entities
        self.meshlets_per_entity = meshlets_per_entity
        self.extent = extent
        self.seed = seed
        self.materials = materials
        self.side = int(math.ceil(entities ** (1.0 / 3.0)))
        self.cell = 2.0 * extent / self.side

    @property
    def total_meshlets(self):
        return self.entities * self.meshlets_per_entity

    @property
    def records_per_entity(self):
        return (self.meshlets_per_entity + 31) // 32

    @property
    def vis_words_per_entity(self):
        return (self.meshlets_per_entity + 31) // 32  # scene.rs:427


def gen_entity_tables(spec: C5Spec, device):
    """Full (unsharded) per-entity tables: EntityDrawBuffer bytes, MeshInfo[], EntityData[].
    Small (52 MB at 195 k entities), replicated on every rank."""
    E, M, S = spec.entities, spec.meshlets_per_entity, spec.seed
    e = torch.arange(E, dtype=torch.int64, device=device)
    # entity draws: identity mapping, visibility words prefix-summed (scene.rs:422-431)
    draws = torch.zeros(1 + 3 * E, dtype=torch.int32, device=device)
    draws[0] = E
    d = draws[1:].view(E, 3)
    d[:, 0] = e.to(torch.int32)
    d[:, 1] = e.to(torch.int32)
    d[:, 2] = (e * spec.vis_words_per_entity).to(torch.int32)
    # mesh infos (128 B = 32 words)
    half = 0.40 * spec.cell
    mesh = torch.zeros(E, 32, dtype=torch.int32, device=device)
    mf = mesh.view(torch.float32)
    mf[:, 3] = half * 1.7320508 + 0.6
    mf[:, 4:7] = -half
    mf[:, 8:11] = half
    mesh[:, 14] = 1  # lod_count
    mesh[:, 16] = (e * M).to(torch.int32)  # mesh_lods[0].meshlet_offset
    mesh[:, 17] = M
    # entities: T(pos) * R_y * S, column-major
    side = spec.side
    ijk = torch.stack([e % side, (e // side) % side, e // (side * side)], dim=1).to(torch.float32)
    jit = torch.stack([rnd_f32(S, 40 + a, e) for a in range(3)], dim=1)
    pos = -spec.extent + (ijk + jit) * spec.cell
    rot_tab = torch.tensor([(1.0, 0.0), (0.0, 1.0), (-1.0, 0.0), (0.6, 0.8), (0.8, -0.6), (5 / 13, 12 / 13),
                            (-15 / 17, 8 / 17), (0.28, 0.96)], dtype=torch.float32, device=device)
    rot = rot_tab[rnd_int(S, 43, e, 0, 7)]
    scl = torch.ones(E, dtype=torch.float32, device=device)  # unit scale keeps the cone statistics clean
    ent = torch.zeros(E, 32, dtype=torch.float32, device=device)
    c, s = rot[:, 0] * scl, rot[:, 1] * scl
    ent[:, 0], ent[:, 2] = c, -s          # column 0 = (c, 0, -s, 0)
    ent[:, 5] = scl                       # column 1
    ent[:, 8], ent[:, 10] = s, c          # column 2 = (s, 0, c, 0)
    ent[:, 12:15] = pos                   # column 3
    ent[:, 15] = 1.0
    ent[:, 16], ent[:, 21], ent[:, 26], ent[:, 31] = 1.0, 1.0, 1.0, 1.0  # normal matrix = identity
    return draws.view(torch.uint8), mesh.view(torch.uint8), ent.view(torch.uint8), half


def gen_meshlets(spec: C5Spec, first_entity, last_entity, device, half, survive_target=0.10, chunk=2_000_000,
                 layout="uploaded", shape="scattered"):
    """Meshlet[] (32 B each) of entities [first_entity, last_entity), generated in chunks.

    layout "uploaded": vertex_offset / data_offset as compute_meshlets + GpuAssets::add_mesh produce them
    (src/assets/mesh.rs:309-316, assets/mod.rs:413-416) — every entity's mesh is an allocation of its own
    (80 data words per meshlet reserved), a meshlet's data follows its predecessor's (vertex_count +
    ceil(3 * triangle_count / 4) words), vertex_offset is per submesh, two submeshes per mesh (the second one
    starts at meshlet 25/64 * M, inside a dispatch record).  "scattered": offsets that follow no rule.

    shape "scattered" (the headline scene): centres, cone axes and cutoffs independent per meshlet, so the ~10 % that
    survive are scattered uniformly over every entity's meshlets — the worst case for everything that touches
    survivors.  "coherent": every entity is a ball whose meshlets follow a spiral from pole to pole (sqrt(pi M) / 2 turns) —
    centre on the surface, cone axis = the outward normal, one cutoff for all — so that, as with a real mesh and a
    real camera, the survivors are the cap that faces the camera: runs of neighbouring meshlets, whole records
    with none."""
    M, S = spec.meshlets_per_entity, spec.seed
    n = (last_entity - first_entity) * M
    chunk = max(M, chunk // M * M)  # whole entities per chunk
    out = torch.empty(n, 8, dtype=torch.int32, device=device)
    # cone cutoff range chosen so that P(cone cull) ~= 1 - survive_target / 0.95 (alpha filter keeps 95 %)
    keep = min(1.0, survive_target / 0.95)
    hi_cut = int(round((-1.0 + 2.0 * keep * 2.0 - 0.0) * 127.0))  # cutoff ~ U[-1, -1 + 4*keep]
    hi_cut = max(-126, min(127, hi_cut))
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        mid = torch.arange(first_entity * M + c0, first_entity * M + c1, dtype=torch.int64, device=device)
        o = out[c0:c1]
        of = o.view(torch.float32)
        if shape == "coherent":
            li = (mid % M).to(torch.float32)
            zc = 1.0 - 2.0 * (li + 0.5) / M
            rc = torch.sqrt(torch.clamp(1.0 - zc * zc, min=0.0))
            phi = li * (2.0 * math.pi * round(math.sqrt(M * math.pi) / 2.0) / M)  # neighbours in index are neighbours on the ball
            v = torch.stack([rc * torch.cos(phi), zc, rc * torch.sin(phi)], dim=1)
            of[:, 0:3] = v * (0.8 * half)
            of[:, 3] = 0.05 + rnd_f32(S, 13, mid) * 0.1
            q = (torch.clamp(v, -1.0, 1.0) * 127.0).to(torch.int32) & 0xFF
            # survivors: dot(view direction, normal) < cutoff, i.e. a cap of (1 + cutoff) / 2 of the ball
            cut = torch.full_like(mid, max(-126, min(127, int(round((2.0 * keep - 1.0) * 127.0))))).to(torch.int32) & 0xFF
        else:
            for a in range(3):
                of[:, a] = (rnd_f32(S, 10 + a, mid) * 2.0 - 1.0) * half
            of[:, 3] = 0.1 + rnd_f32(S, 13, mid) * 0.5
            v = torch.stack([rnd_f32(S, 14 + a, mid) * 2.0 - 1.0 for a in range(3)], dim=1)
            nrm = torch.sqrt((v * v).sum(dim=1))
            nrm = torch.where(nrm == 0, torch.ones_like(nrm), nrm)
            q = (torch.clamp(v / nrm[:, None], -1.0, 1.0) * 127.0).to(torch.int32) & 0xFF  # trunc, math.rs:201-203
            cut = rnd_int(S, 17, mid, -127, hi_cut).to(torch.int32) & 0xFF
        o[:, 4] = q[:, 0] | (q[:, 1] << 8) | (q[:, 2] << 16) | (cut << 24)
        mat = rnd_int(S, 20, mid, 0, spec.materials - 1)
        vc = rnd_int(S, 21, mid, 3, 64)
        tc = rnd_int(S, 22, mid, 1, 64)
        if layout == "scattered":
            o[:, 5] = (rnd_u64(S, 19, mid) & 0x00FFFFFF).to(torch.int32)
            o[:, 6] = ((mid * 88) & 0x7FFFFFFF).to(torch.int32)
        else:
            ent, local = mid // M, mid % M
            size = (vc + (tc * 3 + 3) // 4).view(-1, M)
            data = ent * (M * 80) + (torch.cumsum(size, dim=1) - size).reshape(-1)
            vtx = ent * 8192 + torch.where(local >= (M * 25) // 64, 4096, 0)
            for col, v in ((5, vtx), (6, data)):
                v = v & 0xFFFFFFFF
                o[:, col] = torch.where(v >= (1 << 31), v - (1 << 32), v).to(torch.int32)
        packed = mat | (vc << 16) | (tc << 24)
        o[:, 7] = torch.where(packed >= (1 << 31), packed - (1 << 32), packed).to(torch.int32)
    return out.view(torch.uint8).reshape(-1)


def gen_materials(spec: C5Spec, device):
    i = torch.arange(spec.materials, dtype=torch.int64, device=device)
    u = rnd_f32(spec.seed, 30, i)
    mat = torch.zeros(spec.materials, 20, dtype=torch.int32, device=device)
    mat.view(torch.float32)[:, 0:4] = 1.0
    mat[:, 16] = torch.where(u < 0.8, 0, torch.where(u < 0.95, 1, 2)).to(torch.int32)
    return mat.view(torch.uint8).reshape(-1)
