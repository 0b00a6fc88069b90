"""Generates tests/golden/spirv_cull.npz and spirv_cluster.npz (no argument), spirv_pyramid.npz (`pyramid`: depth_reduce.comp.spv
level by level), spirv_compact.npz (`compact`: active_cluster_compaction.comp.spv) and spirv_cull_dispatch_sizes.npz
(`dispatch`: the two cull shaders with MESHLET_DISPATCH_SIZE = 64 / 128, pass 0) and spirv_cull_contracted.npz
(`contracted`: the cull cases with Dot / matrix products / Length as fma chains): inputs and OUTPUTS OF THE
REFERENCE'S OWN COMPILED SHADERS.

Run in the build container only (it reads /root/reference/shaders/{entity_cull,meshlet_cull}.comp.spv,
forward/forward_depth_prepass.task.spv and light_cluster/{mark_active,light_culling}.comp.spv, which do not travel): `python tests/golden/make_spirv_vectors.py`.  The binaries are executed by oracle/spirv_vm.py; what the
interpreter decides itself is listed in its header (operation order inside Dot / MatrixTimesVector, log2, the ReduceMin
sampler, invocation order of atomics — the oracle's documented canonical choices, passed in from the oracle).  Every
other bit of the outputs is the reference binary's: which tests run, on what, in which arithmetic (including
spirv-opt's reciprocal, fused and factored operations), the visibility-bit protocol, LOD selection, record and command
contents.

Cases: occlusion passes 0 / 1 / 2 x perspective / orthographic, each on a seeded "plain" scene and on a "knife-edge"
scene — quantised geometry whose meshlet radii put the cone test within an ulp of equality and, in pass 2, a constant
depth buffer bisected to the closest-depth of a group of identical spheres — on which the round-1 arithmetic (IEEE
division by 127, unfused sums) provably differs from the binaries.
The tests that consume the file: tests/test_spirv_vectors_cpu.py (oracle), tests/test_spirv_vectors_gpu.py (product)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes as sc  # noqa: E402
from oracle import oracle, spirv_vm as vm  # noqa: E402
from orbit_amd import layouts as L  # noqa: E402

REF = "/root/reference/shaders/"
F = np.float32
W, H = 320, 180
ORTHO = dict(p00=1.0 / 16, p11=1.0 / 16, z_near=0.7, z_far=61.3)
RCP127 = np.frombuffer(np.array([0x3C010204], np.uint32).tobytes(), np.float32)[0]
_mods = {}


def module(name, S=32):
    if (name, S) not in _mods:
        _mods[name, S] = vm.Module(REF + name, spec={0: S})  # constant_id 0 = MESHLET_DISPATCH_SIZE / local_size_x
    return _mods[name, S]


def log2(x):
    return np.array([oracle.log2f(float(v)) for v in np.asarray(x).reshape(-1)], F).reshape(np.shape(x))


def model_view(scene, cam, k):
    ed = scene.entity_draws[k]
    M = np.asarray(scene.entities["model_matrix"][ed["entity_index"]], F).reshape(4, 4).T
    MV = (np.asarray(cam.view, np.float64) @ M.astype(np.float64)).astype(F)
    scale = F(np.sqrt(max((MV[:3, 0] ** 2).sum(), (MV[:3, 1] ** 2).sum(), (MV[:3, 2] ** 2).sum())))
    return ed, scene.mesh_infos[ed["mesh_index"]], MV, scale


def quantise(scene, rng):
    m = scene.meshlets
    n = len(m)
    m["bounding_sphere"][:, :3] = rng.integers(-12, 13, (n, 3)).astype(F) / 4
    m["bounding_sphere"][:, 3] = rng.choice(np.array([0, .125, .25, .5, 1, 2], F), n)
    m["cone_axis"][:] = rng.integers(-128, 128, (n, 3)).astype(np.int8)
    m["cone_cutoff"][:] = rng.integers(-128, 128, n).astype(np.int8)
    for k in range(len(scene.entities)):
        M = np.eye(4, dtype=F)
        perm, sg = rng.permutation(3), rng.choice([-1.0, 1.0], 3)
        R = np.zeros((3, 3), F)
        for i in range(3):
            R[i, perm[i]] = sg[i]
        M[:3, :3] = R * rng.choice(np.array([.5, 1, 2], F))
        M[:3, 3] = rng.integers(-20, 21, 3).astype(F) / 2
        scene.entities["model_matrix"][k] = sc.mat4_cols(M)
    mi = scene.mesh_infos
    mi["bounding_sphere"][:, :3] = rng.integers(-8, 9, (len(mi), 3)).astype(F) / 2
    mi["bounding_sphere"][:, 3] = rng.choice(np.array([1, 2, 4, 8], F), len(mi))


def cone_knife_edge(scene, cam, ortho):
    """Model radii that put dot(d, axis) >= fma(cutoff, |d|, r) within an ulp of equality."""
    ml = scene.meshlets
    for k in range(scene.entity_draw_count):
        _, mi, MV, scale = model_view(scene, cam, k)
        for lod in range(int(mi["lod_count"])):
            off, cnt = mi["mesh_lods"][lod]
            for j in range(int(off), int(off + cnt)):
                d = (MV[:3, :3] @ ml["bounding_sphere"][j, :3] + MV[:3, 3]).astype(F)
                if ortho:
                    d = np.array([0, 0, -1], F)
                a = (MV[:3, :3] @ (ml["cone_axis"][j].astype(F) * RCP127)).astype(F)
                r = F(d @ a) - F(ml["cone_cutoff"][j]) * RCP127 * F(np.sqrt(d @ d))
                if r > 0 and scale > 0:
                    ml["bounding_sphere"][j, 3] = F(r) / scale


def depth_knife_edge(scene, cam, ci, ortho, seed, rng):
    """Every meshlet of the largest entity becomes the same sphere (inexact radius x scale); returns a constant depth
    buffer at the ORACLE's closest-depth of that sphere (found by bisection), or one ulp above it."""
    best = max(range(scene.entity_draw_count),
               key=lambda k: int(scene.mesh_infos[scene.entity_draws[k]["mesh_index"]]["mesh_lods"][0][1]))
    ed = scene.entity_draws[best]
    mi = scene.mesh_infos[ed["mesh_index"]]
    M = np.eye(4, dtype=F)
    M[:3, :3] *= F(1.3)
    M[:3, 3] = (0.3, -0.2, -float(rng.integers(3, 9)) - 0.37)
    scene.entities["model_matrix"][ed["entity_index"]] = sc.mat4_cols(M)
    c0 = np.array([0.11, -0.07, 0.05, 0.0], F) * F(1 + seed % 7)
    c0[3] = F(1.7) + F(0.37) * F(seed % 5)
    mi["bounding_sphere"][:] = (0, 0, 0, 1.0)
    scene.mesh_infos[ed["mesh_index"]] = mi
    for lod in range(int(mi["lod_count"])):
        off, cnt = mi["mesh_lods"][lod]
        scene.meshlets["bounding_sphere"][int(off):int(off + cnt)] = c0
        scene.meshlets["cone_cutoff"][int(off):int(off + cnt)] = 127
        scene.meshlets["cone_axis"][int(off):int(off + cnt)] = (0, 0, 127)  # faces the camera: never cone-culled
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    ze, zm = np.zeros((scene.entity_draw_count + 31) // 32, np.uint32), np.zeros(scene.vis_words, np.uint32)
    entity = int(ed["entity_index"])

    def drawn(bits):
        dc = np.full((H, W), np.frombuffer(np.array([bits], np.uint32).tobytes(), F)[0], F)
        pyr, d = oracle.depth_reduce(dc, W, H)
        d0, _, _ = oracle.entity_cull(ci, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                      scene.entities, cap_d, ze, pyr, (d.width, d.height))
        c, _, _ = oracle.meshlet_cull(ci, d0, scene.meshlets, cap_c, scene.entities, scene.materials, zm, pyr,
                                      (d.width, d.height))
        return int((L.draw_buffer_commands(c)[1]["cmd_first_instance"] == entity).sum())
    lo, hi = 0, 0x3F800000
    base = drawn(lo)
    assert base > 0 and drawn(hi) < base, "the knife-edge group is not decided by the depth buffer"
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if drawn(mid) == base:
            lo = mid
        else:
            hi = mid
    return np.full((H, W), np.frombuffer(np.array([lo + (seed & 1)], np.uint32).tobytes(), F)[0], F)


def case(seed, occlusion_pass, ortho, knife, n=140, S=32, fused_dot=False):
    rng = np.random.default_rng(seed)
    scene = sc.make_scene(seed, n, meshlets_per_mesh=(1, 70), lods=3,
                          meshlet_layout="scattered")  # the layout the committed vectors were made with
    cam = sc.default_camera(rot=(0.2, 0.4))
    if knife:
        quantise(scene, rng)
        cam = sc.default_camera(position=(0.0, 0.0, float(rng.integers(4, 30))), rot=(1.0, 0.0))
        cone_knife_edge(scene, cam, ortho)
    kw = {}
    if occlusion_pass == 2:
        kw = ORTHO if ortho else dict(p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
    ci = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=occlusion_pass, projection_type=1 if ortho else 0, **kw)
    depth = sc.make_depth(9 + seed, W, H, cam)
    evis = rng.integers(0, 2 ** 32, (scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mvis = rng.integers(0, 2 ** 32, scene.vis_words, dtype=np.uint32)
    if knife and occlusion_pass == 2:
        depth = depth_knife_edge(scene, cam, ci, ortho, seed, rng)
        evis[:], mvis[:] = 0, 0
    pyr, dsc = oracle.depth_reduce(depth, W, H)
    ps = (dsc.width, dsc.height)
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    # registry: one index space for all storage buffers (set 0, binding 0), textures and samplers apart
    B = {}

    def buf(i, a):
        B[i] = np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()
    civ = np.array(ci).copy().reshape(())
    if occlusion_pass:
        civ["visibility_buffer"], civ["meshlet_visibility_buffer"], civ["depth_pyramid"] = 6, 7, 11
    buf(1, scene.entity_draw_buffer()), buf(2, scene.mesh_infos), buf(4, scene.entities), buf(5, civ.reshape(1))
    B[3] = np.zeros(L.DISPATCH_HEADER + 16 * cap_d, np.uint8)
    B[3][4:12] = np.array([1, 1], np.uint32).view(np.uint8)  # the clear of draw_gen.rs:356-363: {0, 1, 1}
    buf(6, evis.copy()), buf(7, mvis.copy()), buf(8, scene.meshlets), buf(10, scene.materials)
    B[9] = np.zeros(L.DRAW_HEADER + 28 * cap_c, np.uint8)
    kwm = dict(images={11: {"size": lambda lod: (max(ps[0] >> lod, 1), max(ps[1] >> lod, 1))}},
               samplers={k: {"id": k} for k in range(8)}, subgroup=32, fused_dot=fused_dot,
               sample=lambda img, smp, u, v, lod: np.array(
                   [oracle.hiz_sample(pyr, ps[0], ps[1], float(u), float(v), float(lod)), 0, 0, 0], F))
    vm.Machine(module("entity_cull.comp.spv", S), B, np.array([1, 2, 3, 4, 5], np.uint32).tobytes(), log2, **kwm).run(
        (scene.entity_draw_count + 255) // 256, batch=8)
    nrec = int(B[3][:4].view(np.uint32)[0])
    vm.Machine(module("meshlet_cull.comp.spv", S), B, np.array([3, 8, 9, 4, 5, 10], np.uint32).tobytes(), log2,
               **kwm).run(nrec, batch=256)
    ndraw = int(B[9][:4].view(np.uint32)[0])
    assert nrec <= cap_d and ndraw <= cap_c
    if S != 32:
        # other dispatch sizes (spec constant 0 = 64 / 128: what src/graphics/device.rs:369-372 derives on devices whose
        # mesh-shader workgroup is that wide): pass 0 only — the visibility-word protocol is consistent for S = 32 alone
        # (scene.rs:427, entity_cull.comp:222) — and no task-shader run: MeshTaskPayload.meshlet_indices has 32 entries
        # whatever S is (types.glsl:196-200), a record with more survivors writes past it
        assert occlusion_pass == 0
        return dict(cull_info=np.array(ci).reshape(1), entity_draws=scene.entity_draw_buffer(), mesh_infos=scene.mesh_infos,
                    entities=scene.entities, meshlets=scene.meshlets, materials=scene.materials,
                    caps=np.array([cap_d, cap_c], np.uint32), dispatch_size=np.array([S], np.uint32),
                    spv_dispatch=B[3][:L.DISPATCH_HEADER + 16 * nrec].copy(), spv_draw=B[9][:L.DRAW_HEADER + 28 * ndraw].copy())
    # the mesh-shading path: the task shader's binary on the same dispatch records (one workgroup per record), from
    # the same visibility words the meshlet stage started from
    T = dict(B)
    T[7] = np.ascontiguousarray(mvis).view(np.uint8).reshape(-1).copy()
    T[12], T[13] = np.zeros(64, np.uint8), np.zeros(64, np.uint8)  # vertex / meshlet-data buffers: not read by the task stage
    tpush = np.zeros(64 + 28, np.uint8)  # mat4 view_proj (unused by the task stage), then the seven registry indices
    tpush[64:].view(np.uint32)[:] = [3, 5, 12, 8, 13, 4, 10]
    tm = vm.Machine(module("forward/forward_depth_prepass.task.spv"), T, tpush.tobytes(), log2, **kwm)
    tm.run(nrec)
    task = np.zeros(nrec, L.MESH_TASK_RECORD)
    for k, (counts, pay) in enumerate(tm.emitted):
        assert counts[1:] == (1, 1)
        task[k]["task_mesh_count"], task[k]["entity_index"], task[k]["meshlet_offset"] = counts[0], int(pay[0]), int(pay[1])
        task[k]["meshlet_indices"] = np.array([int(x) & 0xFF for x in pay[2]], np.uint8)
    assert len(tm.emitted) == nrec
    return dict(spv_task_records=task, spv_task_mvis=T[7].view(np.uint32).copy(),cull_info=np.array(ci).reshape(1), entity_draws=scene.entity_draw_buffer(),
                mesh_infos=scene.mesh_infos, entities=scene.entities, meshlets=scene.meshlets,
                materials=scene.materials, evis_in=evis, mvis_in=mvis, pyramid=pyr,
                pyramid_size=np.array(ps, np.uint32), caps=np.array([cap_d, cap_c], np.uint32),
                spv_dispatch=B[3][:L.DISPATCH_HEADER + 16 * nrec].copy(), spv_draw=B[9][:L.DRAW_HEADER + 28 * ndraw].copy(),
                spv_evis=B[6].view(np.uint32).copy(), spv_mvis=B[7].view(np.uint32).copy())


# ------------------------------------------------------------------------------------------------ light clusters
CREF = REF + "light_cluster/"


def cluster_case(seed, width, height, n_lights, quantised, samples=1):
    """mark_active.comp.spv and light_culling.comp.spv (active_cluster_compaction.comp is integer-only and its list
    order is the atomics' — the oracle's list is handed to the last stage as input)."""
    cam = sc.default_camera(aspect=width / height)
    depth = sc.make_depth(seed, width, height, cam, n_occluders=40, ground=(seed % 2 == 0))
    if samples > 1:  # a multisampled depth buffer [h][w][samples] (cluster.rs:439-456)
        jit = sc.rnd_range(seed, 90, np.arange(width * height * samples), 0.9, 1.1).reshape(height, width, samples)
        depth = (depth[:, :, None] * jit).astype(F)
    tile, zsl, far = 8, 32, 200.0
    cx, cy = -(-width // tile), -(-height // tile)
    zs, zb = oracle.cluster_grid_info(cam.z_near, far, zsl)
    push = np.zeros((), dtype=L.MARK_ACTIVE_PUSH)
    push["cluster_count"], push["tile_size_px"], push["screen_size"] = (cx, cy, zsl), tile, (width, height)
    push["z_near"], push["z_far"], push["z_scale"], push["z_bias"] = cam.z_near, far, zs, zb
    push["depth_buffer_sample_count"] = samples
    info = np.zeros((), dtype=L.CLUSTER_CULL_INFO)
    info["world_to_view_matrix"] = sc.mat4_cols(cam.view)
    info["screen_to_view_matrix"] = sc.mat4_cols(np.linalg.inv(cam.proj.astype(np.float64)).astype(F))
    info["cluster_count"], info["tile_size_px"], info["screen_size"] = (cx, cy, zsl), tile, (width, height)
    info["z_near"], info["z_far"], info["global_light_count"] = cam.z_near, far, n_lights
    lights = sc.make_lights(seed, n_lights)
    if quantised is True:  # positions on a quarter-unit grid around the camera, few radii: exact ties in the sphere test
        rng = np.random.default_rng(seed)
        lights = lights.copy()
        lights["position"][:] = (rng.integers(-24, 25, (n_lights, 3)) / 4).astype(F)
        lights["outer_radius"][:] = rng.choice(np.array([0.5, 1, 2, 3, 5, 7.5], F), n_lights)
    total = cx * cy * zsl
    B = {1: np.zeros(4 * cx * cy, np.uint8), 2: np.zeros(8 * total, np.uint8)}
    pv = np.array(push).copy().reshape(())
    pv["depth_buffer"], pv["depth_slice_mask_buffer"], pv["depth_bounds_buffer"] = 7, 1, 2
    clipped = lambda x, y: (np.clip(y, 0, height - 1), np.clip(x, 0, width - 1))
    fetch = lambda x, y: np.stack([depth[clipped(x, y)]] + [np.zeros(len(x), F)] * 3, axis=1)
    fetch_ms = lambda x, y, k: np.stack([depth[clipped(x, y) + (k,)]] + [np.zeros(len(x), F)] * 3, axis=1)
    vm.Machine(vm.Module(CREF + "mark_active.comp.spv"), B, pv.tobytes(), log2,
               images={7: {"size": lambda lod: (width, height), "fetch": fetch, "fetch_ms": fetch_ms}},
               samplers={k: {"id": k} for k in range(8)},
               subgroup=32).run(((width + 7) // 8, (height + 7) // 8), batch=16)
    masks, bounds = B[1].view(np.uint32).copy(), B[2].view(np.uint32).reshape(-1, 2).copy()
    unique, _ = oracle.cluster_compact([cx, cy, zsl], masks, total)
    na = int(unique[12:16].view(np.uint32)[0])
    if quantised == "knife":
        # lights whose radius puts the sphere-box test of one active cluster within an ulp of equality: centre beyond
        # the box's max corner by an inexact offset, radius = sqrt of the squared distance (so r * r ~ sqr_dist)
        rng = np.random.default_rng(seed)
        lights = lights.copy()
        view = np.asarray(cam.view, np.float64)
        act = unique[16:16 + 4 * na].view(np.uint32)
        for k in range(n_lights):
            mn, mx = oracle.cluster_aabb(info, bounds, int(act[rng.integers(0, na)]))
            off = rng.uniform(0.05, 1.5, 3).astype(F)
            sel = rng.integers(0, 2, 3).astype(bool)  # some axes inside the slab (no contribution), at least one outside
            sel[rng.integers(0, 3)] = True
            centre = np.where(sel, mx + off, (mn + mx) * F(0.5)).astype(F)
            d = np.where(sel, centre - mx, F(0)).astype(F)
            lights["outer_radius"][k] = F(np.sqrt(np.float64(d[0]) ** 2 + np.float64(d[1]) ** 2 + np.float64(d[2]) ** 2))
            world = np.linalg.inv(view) @ np.array([centre[0], centre[1], centre[2], 1.0])
            lights["position"][k] = world[:3].astype(F)
    cap_l = 256 * na + 16
    iv = np.array(info).copy().reshape(())
    iv["unique_cluster_buffer"], iv["cluster_offset_image"], iv["light_index_buffer"] = 1, 9, 2
    iv["depth_bounds_buffer"], iv["global_light_list"] = 3, 4
    img = np.zeros((total, 2), np.uint32)

    def write(coord, texel):
        x, y, z = (int(v) for v in coord[:3])
        img[x + y * cx + z * cx * cy] = (int(texel[0]), int(texel[1]))
    B = {5: np.frombuffer(iv.tobytes(), np.uint8).copy(), 1: unique.copy(), 2: np.zeros(4 + 4 * cap_l, np.uint8),
         3: np.ascontiguousarray(bounds).view(np.uint8).reshape(-1).copy(),
         4: np.ascontiguousarray(lights).view(np.uint8).reshape(-1).copy()}
    vm.Machine(vm.Module(CREF + "light_culling.comp.spv"), B, np.array([5], np.uint32).tobytes(), log2,
               images={9: {"write3": write}}, samplers={}, subgroup=32).run((na + 255) // 256)
    nl = int(B[2][:4].view(np.uint32)[0])
    return dict(push=np.array(push).reshape(1), depth=depth, info=np.array(info).reshape(1), lights=lights,
                unique=unique[:16 + 4 * na].copy(), spv_masks=masks, spv_bounds=bounds,
                spv_light_list=B[2][:4 + 4 * nl].copy(), spv_offset_image=img)


def pyramid_depth(seed, sw, sh):
    """The depth buffer of a pyramid case, from its seed alone (so that a large case needs no stored input): a
    reverse-Z gradient, blocks of nearer geometry, noise, a quarter of the texels on a coarse grid (ties) and a few
    exact zeros / ones."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:sh, 0:sw]
    d = (0.05 + 0.6 * y / max(sh - 1, 1)).astype(F)
    for _ in range(24):
        w, h = int(rng.integers(4, max(5, sw // 3))), int(rng.integers(4, max(5, sh // 3)))
        x0, y0 = int(rng.integers(0, sw)), int(rng.integers(0, sh))
        d[y0:y0 + h, x0:x0 + w] = F(rng.uniform(0.0, 1.0))
    d = (d + rng.uniform(-0.01, 0.01, d.shape).astype(F)).astype(F)
    grid = rng.random(d.shape) < 0.25
    d[grid] = (np.round(d[grid] * F(32)) / F(32)).astype(F)
    d[rng.random(d.shape) < 0.001] = F(0)
    d[rng.random(d.shape) < 0.001] = F(1)
    return np.clip(d, 0, 1).astype(F)


def pyramid_case(seed, sw, sh):
    """depth_reduce.comp.spv dispatched level by level, as DepthPyramid::update does (draw_gen.rs:510-566): level 0
    from the depth buffer, level k from level k - 1, each dispatch ceil(w / 16) x ceil(h / 16) workgroups of the
    binary.  What the binary decides: which invocations write, the coordinate `(vec2(pos) + 0.5) / vec2(dimensions)`
    in its own arithmetic, one sample at LOD 0, one store.  The sample itself is the reduce-min sampler model the
    interpreter is handed (oracle_hiz_sample on the source image: driver-defined)."""
    depth = pyramid_depth(seed, sw, sh)
    npot = lambda v: 1 << max(int(v) - 1, 0).bit_length()
    w0, h0 = max(npot(sw) // 2, 1), max(npot(sh) // 2, 1)  # draw_gen.rs:458
    mips = int(np.floor(np.log2(max(w0, h0)))) + 1           # math.rs:18-20
    mod = module("depth_reduce.comp.spv")
    levels, src, src_w, src_h = [], np.ascontiguousarray(depth).reshape(-1), sw, sh
    for k in range(mips):
        w, h = max(w0 >> k, 1), max(h0 >> k, 1)
        dst = np.full((h, w), np.nan, F)  # every texel must be written exactly once

        def sample(img, smp, u, v, lod, src=src, src_w=src_w, src_h=src_h):
            assert float(lod) == 0.0
            return np.array([oracle.hiz_sample(src, src_w, src_h, float(u), float(v), 0.0), 0, 0, 0], F)

        def write(x, y, texel, dst=dst):
            assert np.isnan(dst[y, x]), "a texel written twice"
            dst[y, x] = texel[0]
        push = np.array([w, h, 1, 2], np.uint32).tobytes()  # dimensions, src_image, dst_image
        vm.Machine(mod, {}, push, log2, images={1: {"size": lambda lod, s=(src_w, src_h): s}, 2: {"write": write}},
                   samplers={k: {"id": k} for k in range(8)}, sample=sample, subgroup=32).run(
            ((w + 15) // 16, (h + 15) // 16), batch=64)
        assert not np.isnan(dst).any(), "a texel of the level was not written"
        levels.append(dst)
        src, src_w, src_h = dst.reshape(-1).copy(), w, h
    return depth, (w0, h0, mips), np.concatenate([l.reshape(-1) for l in levels])


def pyramid_main():
    """tests/golden/spirv_pyramid.npz: small cases with inputs and the binary's pyramids (odd and non-power-of-two
    screens, a screen smaller than a workgroup, one wider than high), and for the 1080p screen of BASELINE config 3
    — whose level 0 is the 1920x1080 -> 1024x1024 fractional-footprint step — the seed and the SHA-256 of the binary's
    pyramid (the input is pyramid_depth(seed, 1920, 1080); 13.9 MB of arrays stay out of the repository)."""
    import hashlib

    out = {}
    for seed, (sw, sh) in ((11, (200, 120)), (12, (97, 61)), (13, (33, 257)), (14, (16, 9)), (15, (1, 1)), (16, (64, 64))):
        depth, (w0, h0, mips), pyr = pyramid_case(seed, sw, sh)
        out[f"{sw}x{sh}/depth"], out[f"{sw}x{sh}/spv_pyramid"] = depth, pyr
        out[f"{sw}x{sh}/geometry"] = np.array([seed, sw, sh, w0, h0, mips], np.uint32)
        print("pyramid %4dx%-4d -> %4dx%-4d, %2d mips, %7d texels" % (sw, sh, w0, h0, mips, pyr.size))
    if "--no-1080p" not in sys.argv:
        depth, (w0, h0, mips), pyr = pyramid_case(17, 1920, 1080)
        out["1920x1080/geometry"] = np.array([17, 1920, 1080, w0, h0, mips], np.uint32)
        out["1920x1080/spv_pyramid_sha256"] = np.frombuffer(hashlib.sha256(pyr.tobytes()).digest(), np.uint8)
        out["1920x1080/depth_sha256"] = np.frombuffer(hashlib.sha256(depth.tobytes()).digest(), np.uint8)
        print("pyramid 1920x1080 -> %dx%d, %d mips, sha256 %s" % (w0, h0, mips, hashlib.sha256(pyr.tobytes()).hexdigest()))
    path = os.path.join(HERE, "spirv_pyramid.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


def compact_main():
    """tests/golden/spirv_compact.npz: active_cluster_compaction.comp.spv (integer-only; a 4x4x4-workgroup dispatch
    whose last workgroup writes the indirect-dispatch header) on the tile masks the mark_active binary produced for
    the five cluster scenes: the header {ceil(n / 256), 1, 1, n} and the list IN THE ORDER OF THE BINARY'S ATOMIC
    APPENDS as the interpreter schedules them (workgroup-major).  On hardware that order is nondeterministic: the
    list is compared as a multiset (the oracle's and the product's canonical order is ascending cluster index)."""
    st = np.load(os.path.join(HERE, "spirv_cluster.npz"))
    out = {}
    for name in ("s1", "s2", "s4", "s5", "s6"):
        masks = st[f"{name}/spv_masks"]
        cc = [int(v) for v in st[f"{name}/push"].view(L.MARK_ACTIVE_PUSH)[0]["cluster_count"]]
        total = cc[0] * cc[1] * cc[2]
        B = {1: masks.view(np.uint8).copy(), 2: np.zeros(16 + 4 * total, np.uint8)}
        vm.Machine(vm.Module(CREF + "active_cluster_compaction.comp.spv"), B, np.array(cc + [1, 2], np.uint32).tobytes(),
                   log2, subgroup=32).run(((cc[0] + 3) // 4, (cc[1] + 3) // 4, (cc[2] + 3) // 4), batch=64)
        n = int(B[2][12:16].view(np.uint32)[0])
        out[f"{name}/cluster_count"], out[f"{name}/masks"] = np.array(cc, np.uint32), masks
        out[f"{name}/spv_unique"] = B[2][:16 + 4 * n].copy()
        print("compaction %s: %d active clusters of %d" % (name, n, total))
    path = os.path.join(HERE, "spirv_compact.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


def contracted_main():
    """tests/golden/spirv_cull_contracted.npz: the twelve cull cases of spirv_cull.npz (same seeds, same inputs — asserted)
    through the same binaries with the ONE driver-defined point a GPU compiler most likely decides the other way: every
    OpDot / OpMatrixTimesVector / OpMatrixTimesMatrix / Length / Distance as an fma chain (acc = a0 b0; acc = fma(ak, bk,
    acc) in component order) instead of rounded products and sums — OrbitCaps.arith_profile = ORBIT_ARITH_CONTRACTED.
    Outputs only."""
    out, summary = {}, []
    committed = np.load(os.path.join(HERE, "spirv_cull.npz"))
    for occlusion_pass in (0, 1, 2):
        for ortho in (False, True):
            for knife in (False, True):
                seed = 3 + 7 * occlusion_pass + (2 if ortho else 0) + (1 if knife else 0)
                name = f"p{occlusion_pass}_{'ortho' if ortho else 'persp'}_{'knife' if knife else 'plain'}"
                c = case(seed, occlusion_pass, ortho, knife, fused_dot=True)
                for k in ("cull_info", "entity_draws", "mesh_infos", "entities", "meshlets", "materials", "caps", "evis_in",
                          "mvis_in", "pyramid"):
                    assert np.array_equal(np.ascontiguousarray(c[k]).view(np.uint8).reshape(-1),
                                          np.ascontiguousarray(committed[f"{name}/{k}"]).view(np.uint8).reshape(-1)), (name, k)
                differ = 0
                for k in ("spv_dispatch", "spv_draw", "spv_evis", "spv_mvis", "spv_task_records", "spv_task_mvis"):
                    out[f"{name}/{k}"] = c[k]
                    a, b = np.ascontiguousarray(c[k]).view(np.uint8).reshape(-1), np.ascontiguousarray(committed[f"{name}/{k}"]).view(np.uint8).reshape(-1)
                    differ += int(len(a) != len(b) or not np.array_equal(a, b))
                summary.append((name, int(c["spv_dispatch"][:4].view(np.uint32)[0]), int(c["spv_draw"][:4].view(np.uint32)[0]), differ))
    path = os.path.join(HERE, "spirv_cull_contracted.npz")
    np.savez_compressed(path, **out)
    for sm in summary:
        print("%-22s records %5d  draws %6d  outputs that differ from the canonical run: %d of 6" % sm)
    print(path, os.path.getsize(path), "bytes")


def dispatch_main():
    """tests/golden/spirv_cull_dispatch_sizes.npz: entity_cull.comp.spv + meshlet_cull.comp.spv with spec constant 0 = 64
    and 128, occlusion pass 0, perspective / orthographic x plain / knife-edge — and, per case, whether the binaries'
    command list equals the S = 32 binaries' on the same scene (the canonical (record, lane) order is the ascending
    meshlet order per entity whatever S cuts the records into).  Outputs only: the inputs are those of spirv_cull.npz's
    pass-0 cases (asserted)."""
    out = {}
    committed = np.load(os.path.join(HERE, "spirv_cull.npz"))
    for S in (64, 128):
        for ortho in (False, True):
            for knife in (False, True):
                seed = 3 + (2 if ortho else 0) + (1 if knife else 0)  # the seeds of spirv_cull.npz's pass-0 cases
                name = f"s{S}_p0_{'ortho' if ortho else 'persp'}_{'knife' if knife else 'plain'}"
                c = case(seed, 0, ortho, knife, S=S)
                c32 = case(seed, 0, ortho, knife, S=32)
                same = np.array_equal(c["spv_draw"], c32["spv_draw"])
                assert same, "the command list depends on the dispatch size?"
                # the inputs ARE spirv_cull.npz's pass-0 case of the same projection and kind: stored there, checked here
                base = f"p0_{'ortho' if ortho else 'persp'}_{'knife' if knife else 'plain'}"
                for k in ("cull_info", "entity_draws", "mesh_infos", "entities", "meshlets", "materials", "caps"):
                    assert np.array_equal(np.ascontiguousarray(c[k]).view(np.uint8).reshape(-1),
                                          np.ascontiguousarray(committed[f"{base}/{k}"]).view(np.uint8).reshape(-1)), (base, k)
                for k in ("dispatch_size", "spv_dispatch", "spv_draw"):
                    out[f"{name}/{k}"] = c[k]
                print("%-26s records %5d (S = 32: %5d)  draws %6d  commands equal to S = 32: %s" % (
                    name, int(c["spv_dispatch"][:4].view(np.uint32)[0]), int(c32["spv_dispatch"][:4].view(np.uint32)[0]),
                    int(c["spv_draw"][:4].view(np.uint32)[0]), same))
    path = os.path.join(HERE, "spirv_cull_dispatch_sizes.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "dispatch":
        return dispatch_main()
    if len(sys.argv) > 1 and sys.argv[1] == "contracted":
        return contracted_main()
    if len(sys.argv) > 1 and sys.argv[1] == "pyramid":
        return pyramid_main()
    if len(sys.argv) > 1 and sys.argv[1] == "compact":
        return compact_main()
    out, summary = {}, []
    cl = {}
    for seed, quantised, samples in ((1, False, 1), (2, True, 1), (4, True, 1), (5, False, 4), (6, "knife", 1)):
        c = cluster_case(seed, 96, 64, 120, quantised, samples)
        for k, v in c.items():
            cl[f"s{seed}/{k}"] = v
        print("cluster seed %d: active %d, light indices %d" % (seed, int(c["unique"][12:16].view(np.uint32)[0]),
                                                                 int(c["spv_light_list"][:4].view(np.uint32)[0])))
    np.savez_compressed(os.path.join(HERE, "spirv_cluster.npz"), **cl)
    for occlusion_pass in (0, 1, 2):
        for ortho in (False, True):
            for knife in (False, True):
                seed = 3 + 7 * occlusion_pass + (2 if ortho else 0) + (1 if knife else 0)
                name = f"p{occlusion_pass}_{'ortho' if ortho else 'persp'}_{'knife' if knife else 'plain'}"
                c = case(seed, occlusion_pass, ortho, knife)
                for k, v in c.items():
                    out[f"{name}/{k}"] = v
                summary.append((name, int(c["spv_dispatch"][:4].view(np.uint32)[0]), int(c["spv_draw"][:4].view(np.uint32)[0]),
                                int(c["spv_task_records"]["task_mesh_count"].sum())))
    path = os.path.join(HERE, "spirv_cull.npz")
    np.savez_compressed(path, **out)
    for s in summary:
        print("%-22s records %5d  draws %6d  task survivors %6d" % s)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
