"""Byte layouts of every buffer on the cull path, as numpy structured dtypes.

These mirror ``include/orbit_abi.h`` (which mirrors the reference's std430 /
``repr(C)`` structs, ``shaders/include/types.glsl`` and the Rust twins cited per
dtype).  They are used to build inputs and to decode outputs; no arithmetic
happens here.
"""
import numpy as np

NONE = 0xFFFFFFFF  # u32::MAX, src/passes/draw_gen.rs:142-144
MAX_CULL_PLANES = 12  # src/passes/draw_gen.rs:206
MAX_MESH_LODS = 8  # shaders/include/types.glsl:4
MESHLET_DISPATCH_SIZE = 32  # src/graphics/device.rs:369-372 (canonical S)
MAX_LIGHTS_PER_CLUSTER = 256  # shaders/light_cluster/light_culling.comp:135

ENTITY_DRAW_HEADER = 4
DISPATCH_HEADER = 12
DRAW_HEADER = 4
COMPACT_HEADER = 16
LIGHT_INDEX_HEADER = 4

ALPHA_OPAQUE, ALPHA_MASKED, ALPHA_TRANSPARENT, ALPHA_ALL = 1, 2, 4, 7  # draw_gen.rs:630-641
LIGHT_TYPE_SKY, LIGHT_TYPE_DIRECTIONAL, LIGHT_TYPE_POINT = 0, 1, 2  # types.glsl:298-300

# types.glsl:112-116, src/scene.rs:127-133
ENTITY_DRAW = np.dtype([("entity_index", "<u4"), ("mesh_index", "<u4"), ("visibility_offset", "<u4")])
# types.glsl:75-78, src/scene.rs:120-125 (column-major mat4)
ENTITY_DATA = np.dtype([("model_matrix", "<f4", (16,)), ("normal_matrix", "<f4", (16,))])
# types.glsl:128-141, src/assets/mod.rs:18-28
MESH_INFO = np.dtype([
    ("bounding_sphere", "<f4", (4,)), ("aabb_min", "<f4", (4,)), ("aabb_max", "<f4", (4,)),
    ("vertex_offset", "<u4"), ("meshlet_data_offset", "<u4"), ("lod_count", "<u4"), ("_padding", "<u4"),
    ("mesh_lods", "<u4", (MAX_MESH_LODS, 2)),  # (meshlet_offset, meshlet_count)
])
# types.glsl:143-152, src/assets/mod.rs:111-122
MESHLET = np.dtype([
    ("bounding_sphere", "<f4", (4,)), ("cone_axis", "i1", (3,)), ("cone_cutoff", "i1"),
    ("vertex_offset", "<u4"), ("data_offset", "<u4"), ("material_index", "<u2"),
    ("vertex_count", "u1"), ("triangle_count", "u1"),
])
# types.glsl:92-110, src/assets/mod.rs:171-191
MATERIAL = np.dtype([
    ("base_color", "<f4", (4,)), ("emissive_factor", "<f4", (3,)), ("metallic_factor", "<f4"),
    ("roughness_factor", "<f4"), ("occlusion_factor", "<f4"), ("alpha_cutoff", "<f4"),
    ("texture_indices", "<u4", (5,)), ("alpha_mode", "<u4"), ("_padding", "<u4", (3,)),
])
# types.glsl:16-27, src/scene.rs:278-291
LIGHT = np.dtype([
    ("light_type", "<u4"), ("shadow_data_index", "<u4"), ("irradiance_map_index", "<u4"),
    ("prefiltered_map_index", "<u4"), ("color", "<f4", (3,)), ("intensity", "<f4"),
    ("position", "<f4", (3,)), ("inner_radius", "<f4"), ("direction", "<f4", (3,)), ("outer_radius", "<f4"),
])
# types.glsl:166-171
MESHLET_DISPATCH = np.dtype([("entity_index", "<u4"), ("meshlet_offset", "<u4"), ("meshlet_count", "<u4"),
                             ("visibility_offset", "<u4")])
# types.glsl:180-189, src/assets/mod.rs:98-109
VISIBLE_HEADER = 8  # {u32 count; u32 0}
# the same list at record granularity: {u32 records; u32 survivors} header, 12 B per dispatch record
VISIBLE_RECORD = np.dtype([("entity_index", "<u4"), ("meshlet_offset", "<u4"), ("mask", "<u4")])
MESH_TASK_RECORD = np.dtype([("task_mesh_count", "<u4"), ("entity_index", "<u4"), ("meshlet_offset", "<u4"),
                             ("meshlet_indices", "u1", (32,))])  # EmitMeshTasksEXT count + types.glsl:196-200
MESHLET_DRAW_COMMAND = np.dtype([
    ("cmd_index_count", "<u4"), ("cmd_instance_count", "<u4"), ("cmd_first_index", "<u4"),
    ("cmd_vertex_offset", "<i4"), ("cmd_first_instance", "<u4"), ("meshlet_vertex_offset", "<u4"),
    ("meshlet_index", "<u4"),
])
# GpuCullInfo, draw_gen.rs:208-237 == CullInfo, types.glsl:202-228
GPU_CULL_INFO = np.dtype([
    ("view_matrix", "<f4", (16,)), ("reprojection_matrix", "<f4", (16,)),
    ("cull_planes", "<f4", (MAX_CULL_PLANES, 4)), ("cull_plane_count", "<u4"),
    ("alpha_mode_flag", "<u4"), ("noskip_alphamode", "<u4"), ("occlusion_pass", "<u4"),
    ("visibility_buffer", "<u4"), ("meshlet_visibility_buffer", "<u4"), ("depth_pyramid", "<u4"),
    ("secondary_depth_pyramid", "<u4"), ("projection_type", "<u4"), ("p00_or_width_recipx2", "<f4"),
    ("p11_or_height_recipx2", "<f4"), ("z_near", "<f4"), ("z_far", "<f4"), ("lod_base", "<f4"),
    ("lod_step", "<f4"), ("min_mesh_lod", "<u4"), ("lod_target_pos_view_space", "<f4", (3,)),
    ("max_mesh_lod", "<u4"),
])
# types.glsl:251-255
CLUSTER_DEPTH_BOUNDS = np.dtype([("min_depth", "<u4"), ("max_depth", "<u4")])
# mark_active.comp:8-23
MARK_ACTIVE_PUSH = np.dtype([
    ("cluster_count", "<u4", (3,)), ("tile_size_px", "<u4"), ("screen_size", "<u4", (2,)),
    ("z_near", "<f4"), ("z_far", "<f4"), ("z_scale", "<f4"), ("z_bias", "<f4"), ("depth_buffer", "<u4"),
    ("depth_buffer_sample_count", "<u4"), ("depth_slice_mask_buffer", "<u4"), ("depth_bounds_buffer", "<u4"),
])
# ClusterCullInfo, cluster.rs:186-207 == light_culling.comp:8-26
CLUSTER_CULL_INFO = np.dtype([
    ("world_to_view_matrix", "<f4", (16,)), ("screen_to_view_matrix", "<f4", (16,)),
    ("cluster_count", "<u4", (3,)), ("tile_size_px", "<u4"), ("screen_size", "<u4", (2,)),
    ("z_near", "<f4"), ("z_far", "<f4"), ("unique_cluster_buffer", "<u4"), ("cluster_offset_image", "<u4"),
    ("light_index_buffer", "<u4"), ("depth_bounds_buffer", "<u4"), ("global_light_count", "<u4"),
    ("global_light_list", "<u4"), ("_padding", "<u4", (2,)),
])
# GpuClusterInfoBuffer, cluster.rs:322-335 == ClusterBuffer, types.glsl:42-53
GPU_CLUSTER_INFO_BUFFER = np.dtype([
    ("cluster_count", "<u4", (3,)), ("tile_size_px", "<u4"), ("screen_size", "<u4", (2,)),
    ("z_slice_count", "<u4"), ("z_scale", "<f4"), ("z_bias", "<f4"), ("luminance_cutoff", "<f4"),
    ("light_offset_image", "<u4"), ("light_index_list", "<u4"), ("tile_depth_slice_mask_buffer", "<u4"),
])

_SIZES = {
    "ENTITY_DRAW": (ENTITY_DRAW, 12), "ENTITY_DATA": (ENTITY_DATA, 128), "MESH_INFO": (MESH_INFO, 128),
    "MESHLET": (MESHLET, 32), "MATERIAL": (MATERIAL, 80), "LIGHT": (LIGHT, 64),
    "MESHLET_DISPATCH": (MESHLET_DISPATCH, 16), "MESHLET_DRAW_COMMAND": (MESHLET_DRAW_COMMAND, 28),
    "MESH_TASK_RECORD": (MESH_TASK_RECORD, 44), "VISIBLE_RECORD": (VISIBLE_RECORD, 12),
    "GPU_CULL_INFO": (GPU_CULL_INFO, 400), "CLUSTER_DEPTH_BOUNDS": (CLUSTER_DEPTH_BOUNDS, 8),
    "MARK_ACTIVE_PUSH": (MARK_ACTIVE_PUSH, 56), "CLUSTER_CULL_INFO": (CLUSTER_CULL_INFO, 192),
    "GPU_CLUSTER_INFO_BUFFER": (GPU_CLUSTER_INFO_BUFFER, 52),
}
for _name, (_dt, _sz) in _SIZES.items():
    assert _dt.itemsize == _sz, (_name, _dt.itemsize, _sz)
assert GPU_CULL_INFO.fields["cull_plane_count"][1] == 320
assert GPU_CULL_INFO.fields["occlusion_pass"][1] == 332
assert GPU_CULL_INFO.fields["lod_target_pos_view_space"][1] == 384
assert MATERIAL.fields["alpha_mode"][1] == 64
assert MESH_INFO.fields["mesh_lods"][1] == 64
assert CLUSTER_CULL_INFO.fields["global_light_count"][1] == 176


def entity_draw_buffer(draws: np.ndarray, count=None) -> np.ndarray:
    """EntityDrawBuffer bytes: u32 count @0, draws[] @4 (types.glsl:118-121)."""
    draws = np.ascontiguousarray(draws, dtype=ENTITY_DRAW)
    out = np.zeros(ENTITY_DRAW_HEADER + draws.nbytes, dtype=np.uint8)
    out[:4].view("<u4")[0] = len(draws) if count is None else count
    out[4:] = draws.view(np.uint8).reshape(-1)
    return out


def dispatch_buffer_records(buf: np.ndarray):
    """(header[3], records) view of a MeshletDispatchBuffer byte array."""
    buf = np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    head = buf[:12].view("<u4").copy()
    n = int(head[0])
    recs = buf[DISPATCH_HEADER:DISPATCH_HEADER + 16 * n].copy().view(MESHLET_DISPATCH)
    return head, recs


def draw_buffer_commands(buf: np.ndarray):
    """(count, commands) view of a MeshletDrawCommandBuffer byte array."""
    buf = np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    n = int(buf[:4].view("<u4")[0])
    cmds = buf[DRAW_HEADER:DRAW_HEADER + 28 * n].copy().view(MESHLET_DRAW_COMMAND)
    return n, cmds


def make_dispatch_buffer(records: np.ndarray, capacity=None) -> np.ndarray:
    """Bytes of a MeshletDispatchBuffer holding `records` (header {n,1,1})."""
    records = np.ascontiguousarray(records, dtype=MESHLET_DISPATCH)
    cap = len(records) if capacity is None else capacity
    out = np.zeros(DISPATCH_HEADER + 16 * cap, dtype=np.uint8)
    out[:12].view("<u4")[:] = (len(records), 1, 1)
    out[DISPATCH_HEADER:DISPATCH_HEADER + records.nbytes] = records.view(np.uint8).reshape(-1)
    return out
