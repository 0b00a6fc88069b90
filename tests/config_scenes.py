"""The workloads of BASELINE.json configs 2-4 exactly as SURVEY.md §8d defines their synthetic stand-ins — ONE
definition, used by the parity tests (tests/test_configs_gpu.py) and by the timing tool (tools/bench_configs.py), so
that what is timed is what is tested.

  C2  Sponza-class: 400 meshes / one entity each, ~6 000 meshlets LOD 0 in a 30 x 12 x 18 m box, 25 materials
      (alpha modes 80 / 15 / 5 %), alpha flag OPAQUE | MASKED, frustum + cone, occlusion_pass 0.
  C3  Bistro-class: 2 900 entities / ~60 000 meshlets in 150 x 40 x 150 m; 1920x1080; reverse-Z depth = K = 64 seeded
      occluder boxes + ground plane; the two-pass frame: pass 1 (bits of the previous frame) -> depth_reduce -> pass 2.
  C4  C3's depth + 10 000 point lights (intensity U[1,6], outer_r = sqrt(I / 0.25)), 240 x 135 x 32 clusters, 8-px
      tiles, far 200.
The real Sponza / Bistro glTFs are not in the reference or in the image."""
import numpy as np

import scenes as sc
from orbit_amd import layouts as L

SCREEN = (1920, 1080)
CLUSTERS = (240, 135, 32)
CLUSTER_FAR = 200.0
N_LIGHTS = 10_000


def camera():
    return sc.default_camera()  # src/app.rs:744-752


def config2_scene():
    return sc.make_scene(2, 400, n_meshes=400, meshlets_per_mesh=(4, 26), extent=(15.0, 6.0, 9.0))


def config3_scene():
    return sc.make_scene(3, 2900, n_meshes=2900, meshlets_per_mesh=(4, 38), extent=(75.0, 20.0, 75.0))


def config3_depth(cam):
    W, H = SCREEN
    return sc.make_depth(9, W, H, cam, n_occluders=64, ground=True)


def pass0_cull_info(cam):
    return sc.make_cull_info(cam.view, cam.planes)  # alpha flag OPAQUE | MASKED (forward.rs:276)


def two_pass_cull_infos(cam):
    ci1 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=1)
    ci2 = sc.make_cull_info(cam.view, cam.planes, occlusion_pass=2, p00=cam.p00, p11=cam.p11, z_near=cam.z_near)
    return ci1, ci2


def frame0_bits(oracle, scene, cam, depth):
    """The visibility words the timed / tested frame starts from: the oracle's result of frame 0 (all-zero bits going
    in, SURVEY.md §8a quirk 3: everything is 'newly visible' in pass 2) on the same depth."""
    W, H = SCREEN
    opyr, d = oracle.depth_reduce(depth, W, H)
    ci1, ci2 = two_pass_cull_infos(cam)
    ev = np.zeros((scene.entity_draw_count + 31) // 32, dtype=np.uint32)
    mv = np.zeros(scene.vis_words, dtype=np.uint32)
    cap_d, cap_c = scene.max_dispatches() + 8, scene.lod0_meshlets + 8
    disp, ev2, _ = oracle.entity_cull(ci2, scene.entity_draw_buffer(), scene.entity_draw_count, scene.mesh_infos,
                                      scene.entities, cap_d, ev, opyr, (d.width, d.height))
    _, mv2, _ = oracle.meshlet_cull(ci2, disp, scene.meshlets, cap_c, scene.entities, scene.materials, mv, opyr,
                                    (d.width, d.height))
    return ev2, mv2


def config4_inputs(oracle, cam):
    """(push block of mark_active, ClusterLightCullInfo, LightData[]) for C4 on C3's screen."""
    W, H = SCREEN
    cx, cy, cz = CLUSTERS
    zs, zb = oracle.cluster_grid_info(cam.z_near, CLUSTER_FAR, cz)
    push = np.zeros((), dtype=L.MARK_ACTIVE_PUSH)
    push["cluster_count"], push["tile_size_px"], push["screen_size"] = (cx, cy, cz), 8, (W, H)
    push["z_near"], push["z_far"], push["z_scale"], push["z_bias"] = cam.z_near, CLUSTER_FAR, zs, zb
    push["depth_buffer_sample_count"] = 1
    info = np.zeros((), dtype=L.CLUSTER_CULL_INFO)
    info["world_to_view_matrix"] = sc.mat4_cols(cam.view)
    info["screen_to_view_matrix"] = sc.mat4_cols(np.linalg.inv(cam.proj.astype(np.float64)).astype(np.float32))
    info["cluster_count"], info["tile_size_px"], info["screen_size"] = (cx, cy, cz), 8, (W, H)
    info["z_near"], info["z_far"], info["global_light_count"] = cam.z_near, CLUSTER_FAR, N_LIGHTS
    return push, info, sc.make_lights(4, N_LIGHTS)
