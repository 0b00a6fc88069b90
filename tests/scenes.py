"""Deterministic synthetic scenes for the parity tests.

Test-side input producers, independent of the product's host code: a counter
based splitmix64 generator fills the reference's buffer layouts
(``orbit_amd.layouts``) with numpy.  The real assets named by BASELINE.json
(Sponza, Bistro) are not in the reference repo or the container, so configs 2-4
are statistical stand-ins with the same shapes; config 5 is synthetic by
definition.
"""
from dataclasses import dataclass, field

import numpy as np

from orbit_amd import layouts as L

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def rnd_u64(seed, stream, idx):
    """Stateless random u64 for (seed, stream, idx)."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = splitmix64(np.uint64(seed) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(stream))
        return splitmix64(base + idx * np.uint64(0xD1B54A32D192ED03))


def rnd_f32(seed, stream, idx):
    """Uniform float32 in [0,1): top 24 bits * 2^-24 (exact)."""
    return (rnd_u64(seed, stream, idx) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)


def rnd_range(seed, stream, idx, lo, hi):
    return (np.float32(lo) + rnd_f32(seed, stream, idx) * np.float32(hi - lo)).astype(np.float32)


def rnd_int(seed, stream, idx, lo, hi):
    """Uniform integer in [lo, hi]."""
    span = np.uint64(hi - lo + 1)
    return (rnd_u64(seed, stream, idx) % span).astype(np.int64) + lo


# ----------------------------------------------------------------------------- matrices (glam restated)
def mat4_identity():
    return np.eye(4, dtype=np.float32)


def mat4_cols(m):
    """Column-major flat float32[16] of a 4x4 given as m[row, col]."""
    return np.ascontiguousarray(np.asarray(m, dtype=np.float32).T).reshape(16)


def perspective_infinite_reverse_rh(fov_y, aspect, z_near):
    """glam Mat4::perspective_infinite_reverse_rh (src/camera.rs:87)."""
    f = np.float32(1.0) / np.tan(np.float32(0.5) * np.float32(fov_y), dtype=np.float32)
    m = np.zeros((4, 4), dtype=np.float32)
    m[0, 0] = f / np.float32(aspect)
    m[1, 1] = f
    m[3, 2] = -1.0
    m[2, 3] = z_near
    return m


def orthographic_rh(left, right, bottom, top, near, far):
    """glam Mat4::orthographic_rh (src/camera.rs:91-96 passes far, near swapped)."""
    rw = np.float32(1.0) / np.float32(right - left)
    rh = np.float32(1.0) / np.float32(top - bottom)
    r = np.float32(1.0) / np.float32(near - far)
    m = np.zeros((4, 4), dtype=np.float32)
    m[0, 0] = rw + rw
    m[1, 1] = rh + rh
    m[2, 2] = r
    m[0, 3] = -(np.float32(left) + np.float32(right)) * rw
    m[1, 3] = -(np.float32(top) + np.float32(bottom)) * rh
    m[2, 3] = r * np.float32(near)
    m[3, 3] = 1.0
    return m


def translation(x, y, z):
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (x, y, z)
    return m


def rotation_y(c, s):
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    return m


def frustum_planes(proj, count=5):
    """math::frustum_planes_from_matrix + normalize_plane (src/math.rs:71-89), numpy float32."""
    p = np.asarray(proj, dtype=np.float32)
    rows = [p[3] + p[0], p[3] - p[0], p[3] + p[1], p[3] - p[1], p[3] + p[2], p[3] - p[2]]
    out = []
    for r in rows[:count]:
        r = r.astype(np.float32)
        n = np.sqrt(np.float32((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]), dtype=np.float32)
        out.append((r / n).astype(np.float32))
    return np.array(out, dtype=np.float32).reshape(-1, 4)


# rotations about Y with exactly reproducible (c, s): Pythagorean triples
_ROT = np.array([(1.0, 0.0), (0.0, 1.0), (-1.0, 0.0), (0.6, 0.8), (0.8, -0.6), (5 / 13, 12 / 13), (-15 / 17, 8 / 17),
                 (0.28, 0.96)], dtype=np.float32)
_SCALE = np.array([0.5, 1.0, 1.0, 1.5, 2.0, 0.75], dtype=np.float32)


@dataclass
class Scene:
    entity_draws: np.ndarray
    entities: np.ndarray
    mesh_infos: np.ndarray
    meshlets: np.ndarray
    materials: np.ndarray
    vis_words: int  # meshlet-visibility words allocated (scene.rs:422-431 prefix sum)
    lod0_meshlets: int  # sum over entity draws of LOD-0 meshlet counts
    meta: dict = field(default_factory=dict)

    @property
    def entity_draw_count(self):
        return len(self.entity_draws)

    def entity_draw_buffer(self, count=None):
        return L.entity_draw_buffer(self.entity_draws, count)

    def max_dispatches(self, lod=0):
        m = self.mesh_infos["mesh_lods"][self.entity_draws["mesh_index"], lod, 1].astype(np.int64)
        return int(((m + 31) // 32).sum())


def make_scene(seed, n_entities, n_meshes=None, meshlets_per_mesh=(4, 40), extent=(30.0, 12.0, 18.0), lods=1,
               n_materials=25, layout="uniform", shuffle=True, mesh_radius=(1.0, 4.0), unit_scale=False,
               meshlet_layout="uploaded"):
    """A scene of `n_entities` entity-draws over `n_meshes` meshes (default: one mesh per entity).

    Meshlet spheres are clustered around their mesh centre; cone axes are random
    unit vectors quantised like math::pack_f32_to_snorm_u8 (src/math.rs:201-203);
    30 % of the cones are degenerate (cutoff 127 = never culled).
    """
    n_meshes = n_entities if n_meshes is None else n_meshes
    S = seed
    mesh_id = np.arange(n_meshes)
    lo, hi = meshlets_per_mesh
    base_counts = rnd_int(S, 1, mesh_id, lo, hi)
    lod_counts = np.zeros((n_meshes, L.MAX_MESH_LODS), dtype=np.int64)
    for k in range(lods):
        lod_counts[:, k] = np.maximum(1, base_counts >> k) if lo > 0 else base_counts >> k
    flat = lod_counts[:, :lods].reshape(-1)
    offs = np.concatenate([[0], np.cumsum(flat)])[:-1].reshape(n_meshes, lods)
    n_meshlets = int(flat.sum())

    mesh_infos = np.zeros(n_meshes, dtype=L.MESH_INFO)
    rad = rnd_range(S, 2, mesh_id, *mesh_radius)
    mesh_infos["bounding_sphere"][:, :3] = np.stack(
        [rnd_range(S, 3 + a, mesh_id, -0.25, 0.25) for a in range(3)], axis=1)
    mesh_infos["bounding_sphere"][:, 3] = rad * np.float32(1.7320508) + np.float32(0.6)
    mesh_infos["aabb_min"][:, :3] = -rad[:, None]
    mesh_infos["aabb_max"][:, :3] = rad[:, None]
    mesh_infos["lod_count"] = lods
    mesh_infos["mesh_lods"][:, :lods, 0] = offs
    mesh_infos["mesh_lods"][:, :lods, 1] = lod_counts[:, :lods]

    # meshlets: owner mesh via repeat over (mesh, lod)
    owner = np.repeat(np.repeat(mesh_id, lods), flat)
    mid = np.arange(n_meshlets)
    meshlets = np.zeros(n_meshlets, dtype=L.MESHLET)
    r_owner = rad[owner]
    for a in range(3):
        meshlets["bounding_sphere"][:, a] = (rnd_f32(S, 10 + a, mid) * np.float32(2.0) - np.float32(1.0)) * r_owner
    meshlets["bounding_sphere"][:, 3] = rnd_range(S, 13, mid, 0.1, 0.6)
    v = np.stack([rnd_f32(S, 14 + a, mid) * np.float32(2.0) - np.float32(1.0) for a in range(3)], axis=1)
    n = np.sqrt((v * v).sum(axis=1, dtype=np.float32), dtype=np.float32)
    n = np.where(n == 0, np.float32(1.0), n)
    v = (v / n[:, None]).astype(np.float32)
    meshlets["cone_axis"] = (np.clip(v, -1.0, 1.0) * np.float32(127.0)).astype(np.int8)  # trunc toward zero
    cut = rnd_int(S, 17, mid, -127, 127)
    degenerate = rnd_f32(S, 18, mid) < np.float32(0.3)
    meshlets["cone_cutoff"] = np.where(degenerate, 127, cut).astype(np.int8)
    meshlets["material_index"] = rnd_int(S, 20, mid, 0, n_materials - 1).astype(np.uint16)
    meshlets["vertex_count"] = rnd_int(S, 21, mid, 3, 64).astype(np.uint8)
    meshlets["triangle_count"] = rnd_int(S, 22, mid, 1, 64).astype(np.uint8)
    if meshlet_layout == "scattered":  # offsets that follow no rule at all
        meshlets["vertex_offset"] = (rnd_u64(S, 19, mid) & np.uint64(0x00FFFFFF)).astype(np.uint32)
        meshlets["data_offset"] = (mid * 88).astype(np.uint32)
    else:
        # "uploaded": as compute_meshlets / add_mesh lay a mesh out (src/assets/mesh.rs:309-316, assets/mod.rs:413-416):
        # a meshlet's data follows its predecessor's, vertex_offset is per submesh, every (mesh, LOD) range is an
        # allocation of its own — with submesh boundaries (5 %) and jumps in the data (3 %) inside the ranges, and
        # 15 % of the meshes left scattered, so that every way the derived command chain can break occurs.
        grp = np.repeat(np.arange(n_meshes * lods), flat)
        first_of_grp = np.ones(n_meshlets, dtype=bool)
        first_of_grp[1:] = grp[1:] != grp[:-1]
        wild = (rnd_f32(S, 23, mesh_id) < np.float32(0.15))[owner]
        new_sub = first_of_grp | wild | (rnd_f32(S, 24, mid) < np.float32(0.05))
        jump = first_of_grp | wild | (rnd_f32(S, 25, mid) < np.float32(0.03))
        size = meshlets["vertex_count"].astype(np.int64) + (meshlets["triangle_count"].astype(np.int64) * 3 + 3) // 4
        gap = np.where(jump, rnd_int(S, 26, mid, 1, 1000), 0)
        data = np.cumsum(gap) + np.concatenate([[0], np.cumsum(size)[:-1]])
        meshlets["data_offset"] = (data & 0xFFFFFFFF).astype(np.uint32)
        sub = np.cumsum(new_sub)
        meshlets["vertex_offset"] = (rnd_u64(S, 19, sub) & np.uint64(0x00FFFFFF)).astype(np.uint32)

    materials = np.zeros(n_materials, dtype=L.MATERIAL)
    u = rnd_f32(S, 30, np.arange(n_materials))
    materials["alpha_mode"] = np.where(u < 0.8, 0, np.where(u < 0.95, 1, 2)).astype(np.uint32)
    materials["base_color"] = 1.0
    materials["alpha_cutoff"] = 0.5

    # entities
    eid = np.arange(n_entities)
    if layout == "lattice":
        side = int(np.ceil(n_entities ** (1.0 / 3.0)))
        cell = np.array([2.0 * extent[0] / side, 2.0 * extent[1] / side, 2.0 * extent[2] / side], dtype=np.float32)
        ijk = np.stack([eid % side, (eid // side) % side, eid // (side * side)], axis=1).astype(np.float32)
        jitter = np.stack([rnd_f32(S, 40 + a, eid) for a in range(3)], axis=1)
        pos = (-np.array(extent, dtype=np.float32) + (ijk + jitter) * cell).astype(np.float32)
    else:
        pos = np.stack([rnd_range(S, 40 + a, eid, -extent[a], extent[a]) for a in range(3)], axis=1)
    rot = _ROT[rnd_int(S, 43, eid, 0, len(_ROT) - 1)]
    scl = np.ones(n_entities, dtype=np.float32) if unit_scale else _SCALE[rnd_int(S, 44, eid, 0, len(_SCALE) - 1)]
    entities = np.zeros(n_entities, dtype=L.ENTITY_DATA)
    mm = np.zeros((n_entities, 4, 4), dtype=np.float32)  # [row, col]
    c, s = rot[:, 0], rot[:, 1]
    mm[:, 0, 0], mm[:, 0, 2], mm[:, 2, 0], mm[:, 2, 2] = c * scl, s * scl, -s * scl, c * scl
    mm[:, 1, 1] = scl
    mm[:, :3, 3] = pos
    mm[:, 3, 3] = 1.0
    entities["model_matrix"] = np.transpose(mm, (0, 2, 1)).reshape(n_entities, 16)
    entities["normal_matrix"] = np.eye(4, dtype=np.float32).reshape(16)

    # entity draws: draw g -> entity perm[g] (indirection exercised when shuffle)
    perm = eid.copy()
    if shuffle and n_entities > 1:
        key = rnd_u64(S, 50, eid)
        perm = np.argsort(key, kind="stable")
    mesh_of_draw = (perm % n_meshes).astype(np.uint32) if n_meshes != n_entities else perm.astype(np.uint32)
    words = (lod_counts[mesh_of_draw, 0] + 31) // 32  # scene.rs:427
    vis_off = np.concatenate([[0], np.cumsum(words)])[:-1]
    draws = np.zeros(n_entities, dtype=L.ENTITY_DRAW)
    draws["entity_index"] = perm
    draws["mesh_index"] = mesh_of_draw
    draws["visibility_offset"] = vis_off
    return Scene(draws, entities, mesh_infos, meshlets, materials, int(words.sum()),
                 int(lod_counts[mesh_of_draw, 0].sum()), dict(seed=seed, extent=extent))


# ----------------------------------------------------------------------------- CullInfo (tests' own builder)
def make_cull_info(view, planes, occlusion_pass=0, projection_type=0, alpha_mode_flag=L.ALPHA_OPAQUE | L.ALPHA_MASKED,
                   noskip_alphamode=0, meshlet_visibility=True, p00=0.0, p11=0.0, z_near=0.0, z_far=0.0,
                   lod_base=16.0, lod_step=2.0, lod_range=(0, 8), lod_target=(0.0, 0.0, 0.0)):
    """The 400-B GpuCullInfo as CullInfo::to_gpu lays it out (draw_gen.rs:121-203)."""
    ci = np.zeros((), dtype=L.GPU_CULL_INFO)
    ci["view_matrix"] = mat4_cols(view)
    planes = np.asarray(planes, dtype=np.float32).reshape(-1, 4)
    ci["cull_planes"][:len(planes)] = planes[:L.MAX_CULL_PLANES]
    ci["cull_plane_count"] = len(planes)
    ci["alpha_mode_flag"] = alpha_mode_flag
    ci["noskip_alphamode"] = noskip_alphamode
    ci["occlusion_pass"] = occlusion_pass
    ci["visibility_buffer"] = 1 if occlusion_pass else L.NONE
    ci["meshlet_visibility_buffer"] = 2 if (occlusion_pass and meshlet_visibility) else L.NONE
    ci["depth_pyramid"] = 3 if occlusion_pass == 2 else L.NONE
    ci["secondary_depth_pyramid"] = 0
    ci["projection_type"] = projection_type
    if occlusion_pass == 2:  # filled only for VisibilityWrite, draw_gen.rs:170-199
        ci["p00_or_width_recipx2"], ci["p11_or_height_recipx2"] = p00, p11
        ci["z_near"], ci["z_far"] = z_near, z_far
    ci["lod_base"], ci["lod_step"] = lod_base, lod_step
    ci["min_mesh_lod"], ci["max_mesh_lod"] = lod_range[0], lod_range[1] - 1
    ci["lod_target_pos_view_space"] = lod_target
    return ci


@dataclass
class Camera:
    view: np.ndarray  # [row, col]
    proj: np.ndarray
    fov: float
    aspect: float
    z_near: float

    @property
    def planes(self):
        return frustum_planes(self.proj, 5)  # forward.rs:264,268

    @property
    def p00(self):
        return float(self.proj[0, 0])

    @property
    def p11(self):
        return float(self.proj[1, 1])


def default_camera(position=(0.0, 2.0, 0.0), rot=(1.0, 0.0), fov_deg=90.0, aspect=16.0 / 9.0, z_near=0.01):
    """Camera of src/app.rs:744-752; view = inverse(T(position) * R_y)."""
    c, s = np.float32(rot[0]), np.float32(rot[1])
    rinv = rotation_y(c, -s)
    view = (rinv @ translation(-position[0], -position[1], -position[2])).astype(np.float32)
    fov = np.float32(np.deg2rad(np.float32(fov_deg)))
    return Camera(view, perspective_infinite_reverse_rh(fov, aspect, z_near), float(fov), aspect, z_near)


# ----------------------------------------------------------------------------- synthetic depth buffer
def make_depth(seed, width, height, cam: Camera, n_occluders=64, ground=True, background_fraction=0.3):
    """Reverse-Z depth (near / z_view, background 0.0): screen-space rectangles at
    seeded view depths over an optional ground plane (config 3 stand-in)."""
    d = np.zeros((height, width), dtype=np.float32)
    ys, xs = np.mgrid[0:height, 0:width]
    ndc_y = (np.float32(1.0) - (ys.astype(np.float32) + np.float32(0.5)) / np.float32(height)) * 2 - 1
    if ground:
        # plane y = -cam_height in view space: ray dir (x/p00, y/p11, -1); hit where ndc_y < 0
        h = np.float32(2.0)
        dir_y = ndc_y / np.float32(cam.p11)
        z = np.where(dir_y < -1e-6, h / np.maximum(-dir_y, np.float32(1e-6)), np.float32(np.inf))
        d = np.maximum(d, (np.float32(cam.z_near) / z).astype(np.float32))
    k = np.arange(n_occluders)
    cx = rnd_range(seed, 60, k, 0.0, 1.0)
    cy = rnd_range(seed, 61, k, 0.0, 1.0)
    hw = rnd_range(seed, 62, k, 0.02, 0.12)
    hh = rnd_range(seed, 63, k, 0.03, 0.2)
    zz = rnd_range(seed, 64, k, 3.0, 120.0)
    for i in range(n_occluders):
        x0, x1 = int(max(0, (cx[i] - hw[i]) * width)), int(min(width, (cx[i] + hw[i]) * width))
        y0, y1 = int(max(0, (cy[i] - hh[i]) * height)), int(min(height, (cy[i] + hh[i]) * height))
        if x1 > x0 and y1 > y0:
            d[y0:y1, x0:x1] = np.maximum(d[y0:y1, x0:x1], np.float32(cam.z_near) / zz[i])
    # punch seeded background holes so sky (depth 0) exists for the cluster path
    if background_fraction > 0 and not ground:
        hole = rnd_f32(seed, 65, (ys * width + xs).reshape(-1)).reshape(height, width) < background_fraction
        d = np.where(hole, np.float32(0.0), d)
    return np.ascontiguousarray(d, dtype=np.float32)


def make_lights(seed, n, extent=(75.0, 20.0, 75.0), cutoff=0.25, point_fraction=1.0):
    """LightData[] with outer_radius = sqrt(intensity / cutoff) (scene.rs:273-275)."""
    i = np.arange(n)
    lights = np.zeros(n, dtype=L.LIGHT)
    is_point = rnd_f32(seed, 70, i) < np.float32(point_fraction)
    lights["light_type"] = np.where(is_point, L.LIGHT_TYPE_POINT, L.LIGHT_TYPE_DIRECTIONAL)
    for a in range(3):
        lights["position"][:, a] = rnd_range(seed, 71 + a, i, -extent[a], extent[a])
        lights["color"][:, a] = rnd_f32(seed, 74 + a, i)
    lights["intensity"] = rnd_range(seed, 77, i, 1.0, 6.0)  # src/app.rs:685
    lights["inner_radius"] = 0.1
    lights["outer_radius"] = np.sqrt(lights["intensity"] / np.float32(cutoff), dtype=np.float32)
    lights["direction"][:, 1] = -1.0
    return lights


# ----------------------------------------------------------------------------- procedural triangle meshes
# (closed surfaces for the asset-side tests and tools: tests/test_assets_cpu.py, tools/hiz_geometry_check.py)
def uv_sphere(n_lat=24, n_lon=48, radius=1.0):
    lat = np.linspace(0, np.pi, n_lat + 1)
    lon = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    pos = np.array([[np.sin(a) * np.cos(b), np.cos(a), np.sin(a) * np.sin(b)] for a in lat for b in lon], dtype=np.float32)
    pos *= radius
    idx = []
    for i in range(n_lat):
        for j in range(n_lon):
            a, b = i * n_lon + j, i * n_lon + (j + 1) % n_lon
            c, d = a + n_lon, b + n_lon
            idx += [a, b, c, b, d, c]  # outward-facing (counter-clockwise seen from outside)
    return pos, np.array(idx, dtype=np.uint32)


def torus(n_major=160, n_minor=24, R=2.0, r=0.6):
    u = np.linspace(0, 2 * np.pi, n_major, endpoint=False)
    v = np.linspace(0, 2 * np.pi, n_minor, endpoint=False)
    pos = np.array([[(R + r * np.cos(b)) * np.cos(a), r * np.sin(b), (R + r * np.cos(b)) * np.sin(a)]
                    for a in u for b in v], dtype=np.float32)
    idx = []
    for j in range(n_minor):  # strips along the major circle: a meshlet of 64 triangles spans 72 degrees of it
        for i in range(n_major):
            a, b = i * n_minor + j, i * n_minor + (j + 1) % n_minor
            c, d = ((i + 1) % n_major) * n_minor + j, ((i + 1) % n_major) * n_minor + (j + 1) % n_minor
            idx += [a, b, c, b, d, c]
    return pos, np.array(idx, dtype=np.uint32)


MESHES = {"sphere": uv_sphere, "torus": torus}


def outward_winding(pos, idx):
    """Flip the winding if the generator's triangles face inwards (the tests want closed, outward-facing surfaces)."""
    tri = pos[idx.reshape(-1, 3)]
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    vol = np.einsum("ij,ij->i", tri[:, 0], n).sum()  # 6 x signed volume
    return idx if vol > 0 else idx.reshape(-1, 3)[:, ::-1].reshape(-1).copy()


# ----------------------------------------------------------------------------- BASELINE config 1 as a scene
def config1_spheres(n=8192):
    """The 8192 spheres of BASELINE.json configs[0] / SURVEY.md §8d C1 (splitmix64 streams of seed 69; the inputs of
    tests/golden/config1_spheres.npz)."""
    i = np.arange(n)
    return np.stack([rnd_range(69, 1, i, -128, 128), rnd_range(69, 2, i, -128, 128), rnd_range(69, 3, i, -128, 128),
                     rnd_range(69, 4, i, 0.25, 4.0)], axis=1).astype(np.float32)


def config1_scene(spheres, placement="model"):
    """The C1 spheres as one-meshlet entities for entity_cull: entity i draws mesh i whose
    bounding sphere is sphere i — placed either by the entity's model matrix (translation, mesh sphere at the
    origin: SURVEY §8d "model = translation") or by the mesh's own bounding-sphere centre under an identity model.
    Returns (entity_draw_buffer bytes, MeshInfo[], EntityData[], Meshlet[], Material[])."""
    n = len(spheres)
    draws = np.zeros(n, dtype=L.ENTITY_DRAW)
    draws["entity_index"] = draws["mesh_index"] = draws["visibility_offset"] = np.arange(n)
    mesh = np.zeros(n, dtype=L.MESH_INFO)
    ent = np.zeros(n, dtype=L.ENTITY_DATA)
    eye = mat4_cols(mat4_identity())
    ent["model_matrix"] = eye
    ent["normal_matrix"] = eye
    if placement == "model":
        ent["model_matrix"][:, 12:15] = spheres[:, :3]
        mesh["bounding_sphere"][:, 3] = spheres[:, 3]
    else:
        mesh["bounding_sphere"] = spheres
    mesh["lod_count"] = 1
    mesh["mesh_lods"][:, 0, 0] = np.arange(n)  # meshlet_offset
    mesh["mesh_lods"][:, 0, 1] = 1             # meshlet_count
    meshlets = np.zeros(n, dtype=L.MESHLET)
    meshlets["bounding_sphere"] = mesh["bounding_sphere"]
    meshlets["cone_cutoff"] = 127
    meshlets["vertex_count"], meshlets["triangle_count"] = 3, 1
    materials = np.zeros(1, dtype=L.MATERIAL)
    return L.entity_draw_buffer(draws), mesh, ent, meshlets, materials
