//! Reference-side binding of the MI355X cull engine (`include/orbit_abi.h`).
//!
//! This is the file a maintainer of Thefefe/orbit adds as `src/passes/orbit_hip.rs`; it is NOT
//! compiled in this repository (no Rust toolchain in the build image) and contains no logic:
//! `extern "C"` declarations mirroring the header, `#[repr(C)]` structs that are byte-identical to
//! the existing `GpuCullInfo` / `ClusterCullInfo`, and the two-line bodies that replace the
//! Vulkan recording in `src/passes/draw_gen.rs` and `src/passes/cluster.rs`.
#![allow(non_camel_case_types, dead_code)]
use std::ffi::{c_char, c_void};

#[repr(C)]
pub struct OrbitCtx { _private: [u8; 0] }
#[repr(C)]
pub struct OrbitMeshletStream { _private: [u8; 0] }

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct OrbitCaps {
    pub max_entities: u32, pub max_dispatches: u32, pub max_draws: u32, pub max_lights: u32,
    pub max_clusters: u32, pub dispatch_size: u32, pub max_views: u32, pub validate_streams: u32, pub cull_path: u32,
    pub arith_profile: u32,
}

/// orbit_compute_clusters' arguments as a block (orbit_frame_late)
#[repr(C)]
pub struct OrbitClusterFrame {
    pub push: *const c_void, pub info: *const c_void, pub depth: *const f32, pub lights: *const c_void,
    pub tile_depth_slice_mask: *mut u32, pub depth_bounds: *mut c_void, pub unique_cluster_buffer: *mut c_void,
    pub light_index_buffer: *mut c_void, pub cluster_offset_image: *mut u32,
    pub index_capacity: u32, pub light_index_capacity: u32,
}
#[repr(C)]
pub struct OrbitFrameLate {
    pub pyramids: *const c_void, pub late_views: *const c_void, pub cascade_views: *const c_void,
    pub clusters: *const OrbitClusterFrame,
    pub pyramid_count: u32, pub late_view_count: u32, pub cascade_view_count: u32, pub _pad: u32,
}

/// push-constant order of shaders/entity_cull.comp:17-23 (== draw_gen.rs:372-376)
#[repr(C)]
pub struct OrbitEntityCullBufs {
    pub entity_draw_buffer: *const c_void, pub mesh_info_buffer: *const c_void,
    pub meshlet_dispatch_buffer: *mut c_void, pub entity_buffer: *const c_void,
    pub visibility_buffer: *mut u32, pub depth_pyramid: *const f32,
    pub depth_pyramid_size: [u32; 2], pub dispatch_capacity: u32, pub _pad: u32,
    /// null: `depth_pyramid` is the packed chain; else a DEVICE array of mip levels (separate per-mip images)
    pub depth_pyramid_levels: *const OrbitDepthPyramidLevel,
}

/// One view of orbit_cull_views (the frame's early pass and shadow cascades culled side by side)
#[repr(C)]
pub struct OrbitCullView {
    pub cull_info: *const c_void, pub entity: OrbitEntityCullBufs, pub meshlet: OrbitMeshletCullBufs,
    pub entity_draw_count: u32, pub skip_meshlet_stage: u32,
}

/// One mip of a pyramid that is not one packed buffer (linear-tiled image per level: INTEGRATION.md)
#[repr(C)] #[derive(Clone, Copy)]
pub struct OrbitDepthPyramidLevel { pub texels: *mut f32, pub row_pitch: u32, pub _pad: u32 }

/// One pyramid of orbit_depth_reduce_multi; exactly one of `pyramid` / `levels` (HOST array) is non-null
#[repr(C)]
pub struct OrbitDepthReduceItem {
    pub depth: *const f32, pub screen_width: u32, pub screen_height: u32, pub depth_row_pitch: u32, pub _pad: u32,
    pub pyramid: *mut f32, pub levels: *const OrbitDepthPyramidLevel,
}

/// push-constant order of shaders/meshlet_cull.comp:16-23 (== draw_gen.rs:426-431)
#[repr(C)]
pub struct OrbitMeshletCullBufs {
    pub meshlet_dispatch_buffer: *const c_void, pub meshlet_buffer: *const c_void,
    pub draw_commands_buffer: *mut c_void, pub entity_buffer: *const c_void,
    pub material_buffer: *const c_void, pub meshlet_visibility_buffer: *mut u32,
    pub depth_pyramid: *const f32, pub depth_pyramid_size: [u32; 2],
    pub dispatch_capacity: u32, pub draw_capacity: u32, pub material_count: u32, pub _pad: u32,
    pub depth_pyramid_levels: *const OrbitDepthPyramidLevel,
}

pub const ORBIT_MAX_PYRAMID_MIPS: usize = 16;
/// Geometry of the R32F mip chain `DepthPyramid::new` allocates (draw_gen.rs:456-494); buffer = 4 * total_texels bytes.
#[repr(C)] #[derive(Clone, Copy)]
pub struct OrbitDepthPyramidDesc {
    pub width: u32, pub height: u32, pub mip_levels: u32, pub total_texels: u32,
    pub mip_offset: [u32; ORBIT_MAX_PYRAMID_MIPS], pub mip_width: [u32; ORBIT_MAX_PYRAMID_MIPS],
    pub mip_height: [u32; ORBIT_MAX_PYRAMID_MIPS],
}

#[link(name = "orbit_cull")]
extern "C" {
    pub fn orbit_abi_version() -> u32;
    pub fn orbit_default_caps(caps: *mut OrbitCaps);
    pub fn orbit_ctx_create(device_id: i32, caps: *const OrbitCaps, out: *mut *mut OrbitCtx) -> i32;
    pub fn orbit_ctx_destroy(ctx: *mut OrbitCtx) -> i32;
    pub fn orbit_last_error(ctx: *const OrbitCtx) -> *const c_char;
    pub fn orbit_ctx_status(ctx: *mut OrbitCtx, stream: *mut c_void, sync: i32) -> i32;
    /// `cull_info` = `bytemuck::bytes_of(&GpuCullInfo)` (400 B, draw_gen.rs:208-237), a HOST pointer
    pub fn orbit_entity_cull(ctx: *mut OrbitCtx, cull_info: *const c_void, bufs: *const OrbitEntityCullBufs,
                             entity_draw_count: u32, stream: *mut c_void) -> i32;
    pub fn orbit_entity_cull_range(ctx: *mut OrbitCtx, cull_info: *const c_void, bufs: *const OrbitEntityCullBufs,
                                   draw_first: u32, draw_count: u32, stream: *mut c_void) -> i32;
    pub fn orbit_meshlet_cull(ctx: *mut OrbitCtx, cull_info: *const c_void, bufs: *const OrbitMeshletCullBufs,
                              stream: *mut c_void) -> i32;
    /// Host only: pyramid geometry for a screen size / from the mip-0 size the cull kernels see.
    pub fn orbit_depth_pyramid_desc(screen_w: u32, screen_h: u32, desc: *mut OrbitDepthPyramidDesc) -> i32;
    pub fn orbit_depth_pyramid_desc_from_mip0(mip0_w: u32, mip0_h: u32, desc: *mut OrbitDepthPyramidDesc) -> i32;
    /// Measurement hooks: HIP events around the dominant kernel on the launch stream.
    pub fn orbit_ctx_profile(ctx: *mut OrbitCtx, enable: i32) -> i32;
    pub fn orbit_ctx_profile_reserve(ctx: *mut OrbitCtx, pairs: u32, stream: *mut c_void) -> i32;
    pub fn orbit_ctx_profile_read(ctx: *mut OrbitCtx, avg_ms: *mut f32, launches: *mut u32) -> i32;
    pub fn orbit_cull_views(ctx: *mut OrbitCtx, views: *const OrbitCullView, count: u32, stream: *mut c_void) -> i32;
    pub fn orbit_depth_reduce(ctx: *mut OrbitCtx, depth: *const f32, screen_w: u32, screen_h: u32,
                              pyramid: *mut f32, stream: *mut c_void) -> i32;
    /// update_multiple_depth_pyramids::<C> (draw_gen.rs:569-628): up to 8 pyramids in one launch
    pub fn orbit_depth_reduce_multi(ctx: *mut OrbitCtx, items: *const OrbitDepthReduceItem, count: u32,
                                    stream: *mut c_void) -> i32;
    pub fn orbit_cluster_mark(ctx: *mut OrbitCtx, push: *const c_void, depth: *const f32, masks: *mut u32,
                              bounds: *mut c_void, stream: *mut c_void) -> i32;
    pub fn orbit_cluster_compact(ctx: *mut OrbitCtx, cluster_count: *const [u32; 3], masks: *const u32,
                                 unique: *mut c_void, index_capacity: u32, stream: *mut c_void) -> i32;
    pub fn orbit_cluster_assign(ctx: *mut OrbitCtx, info: *const c_void, unique: *const c_void, bounds: *const c_void,
                                lights: *const c_void, light_index_buffer: *mut c_void, light_index_capacity: u32,
                                cluster_offset_image: *mut u32, stream: *mut c_void) -> i32;
    /// Mesh-shading path: per dispatch record the EmitMeshTasksEXT count + MeshTaskPayload (44 B) that the task
    /// shaders (forward_depth_prepass.task, forward.task) compute; a task shader then only loads its record.
    pub fn orbit_meshlet_task_cull(ctx: *mut OrbitCtx, cull_info: *const c_void, bufs: *const c_void,
                                   task_records: *mut c_void, stream: *mut c_void) -> i32;
    /// compute_clusters (cluster.rs:368-397): the three stages in one call.
    pub fn orbit_compute_clusters(ctx: *mut OrbitCtx, push: *const c_void, info: *const c_void, depth: *const f32,
                                  lights: *const c_void, masks: *mut u32, bounds: *mut c_void, unique: *mut c_void,
                                  index_capacity: u32, light_index_buffer: *mut c_void, light_index_capacity: u32,
                                  cluster_offset_image: *mut u32, stream: *mut c_void) -> i32;
    /// Sharded engine: the shard range; SURVEY §8b's gather of 28-B command lists over the caller's ncclComm_t (one host sync).
    pub fn orbit_shard_range(entity_draw_count: u32, rank: u32, world: u32, begin: *mut u32, end: *mut u32);
    pub fn orbit_gather_visible(ctx: *mut OrbitCtx, nccl_comm: *mut c_void, rank: u32, world: u32,
                                local_draw_buffer: *const c_void, out_draw_buffer: *mut c_void, out_capacity: u32,
                                stream: *mut c_void) -> i32;
    /// The sharded engine's visible list: 12 B {entity_index, meshlet_offset, mask} per dispatch record, written by the
    /// evaluation launch itself; expanded on the receiving side.
    pub fn orbit_meshlet_cull_visible_records(ctx: *mut OrbitCtx, cull_info: *const c_void, bufs: *const c_void,
                                              record_buffer: *mut c_void, record_capacity: u32,
                                              stream: *mut c_void) -> i32;
    pub fn orbit_expand_visible_records(ctx: *mut OrbitCtx, record_buffer: *const c_void, meshlet_buffer: *const c_void,
                                        draw_commands_buffer: *mut c_void, draw_capacity: u32,
                                        stream: *mut c_void) -> i32;
    // derived meshlet streams: GpuAssets::add_mesh calls `update` for the range it wrote (assets/mod.rs:441-445),
    // every cull context binds the stream once
    pub fn orbit_meshlet_stream_create(ctx: *mut OrbitCtx, first_meshlet: u64, capacity: u64,
                                       out_stream: *mut *mut OrbitMeshletStream) -> i32;
    pub fn orbit_meshlet_stream_update(ctx: *mut OrbitCtx, ms: *mut OrbitMeshletStream, meshlet_buffer: *const c_void,
                                       first: u64, count: u64, stream: *mut c_void) -> i32;
    // GpuAssets::add_material calls `set_materials` after the upload (assets/mod.rs:520): the alpha classes
    pub fn orbit_meshlet_stream_set_materials(ctx: *mut OrbitCtx, ms: *mut OrbitMeshletStream, material_buffer: *const c_void,
                                              material_count: u32, stream: *mut c_void) -> i32;
    pub fn orbit_meshlet_stream_validate(ctx: *mut OrbitCtx, ms: *mut OrbitMeshletStream, meshlet_buffer: *const c_void,
                                         material_buffer: *const c_void, stream: *mut c_void) -> i32;
    pub fn orbit_meshlet_stream_destroy(ms: *mut OrbitMeshletStream) -> i32;
    // GpuAssets::add_mesh calls `update_meshes` for the MeshInfo it wrote (assets/mod.rs:18-28): the entity stage then
    // reads a 32-B side entry per mesh instead of the 128-B MeshInfo line
    pub fn orbit_meshlet_stream_update_meshes(ctx: *mut OrbitCtx, ms: *mut OrbitMeshletStream, mesh_info_buffer: *const c_void,
                                              first_mesh: u32, count: u32, stream: *mut c_void) -> i32;
    pub fn orbit_ctx_mesh_side_culls(ctx: *const OrbitCtx) -> u64;
    pub fn orbit_ctx_bind_meshlet_stream(ctx: *mut OrbitCtx, ms: *mut OrbitMeshletStream) -> i32;
    pub fn orbit_ctx_fused_culls(ctx: *const OrbitCtx) -> u64;
    /// The late half of a frame (app.rs:1151-1212) with its independent chains side by side: {pyramids -> late culls} on
    /// `stream`, {cascade culls} and {compute_clusters} on two streams of the context, forked and joined by events.
    pub fn orbit_frame_late(ctx: *mut OrbitCtx, frame: *const OrbitFrameLate, stream: *mut c_void) -> i32;
    pub fn orbit_ctx_meshlet_stream_culls(ctx: *const OrbitCtx) -> u64;
    pub fn orbit_ctx_meshlet_class_culls(ctx: *const OrbitCtx) -> u64;
    // exchange without a host in the step: IPC-mapped peer buffers + a device-signalled rank-ordered scatter (orbit_exchange_list)
    pub fn orbit_p2p_alloc(ctx: *mut OrbitCtx, bytes: u64, out_ptr: *mut *mut c_void, out_handle: *mut [u8; 64]) -> i32;
    pub fn orbit_p2p_free(ctx: *mut OrbitCtx, ptr: *mut c_void) -> i32;
    pub fn orbit_p2p_open(ctx: *mut OrbitCtx, handle: *const [u8; 64], out_peer_ptr: *mut *mut c_void) -> i32;
    pub fn orbit_p2p_close(ctx: *mut OrbitCtx, peer_ptr: *mut c_void) -> i32;
    pub fn orbit_meshlet_cull_records_and_commands(ctx: *mut OrbitCtx, cull_info: *const c_void,
                                                   bufs: *const OrbitMeshletCullBufs, record_buffer: *mut c_void,
                                                   record_capacity: u32, stream: *mut c_void) -> i32;
    pub fn orbit_exchange_list(ctx: *mut OrbitCtx, local_list: *const c_void, rank: u32, world: u32,
                               out_buffers: *const *mut c_void, ctrl_buffers: *const *mut c_void, out_capacity: u32,
                               header_bytes: u32, stride: u32, stream: *mut c_void) -> i32;
    /// A rank's whole cull (entity range + meshlet stage into the record list [+ its own commands]) as one call — one
    /// launch for pass 0 and up to 65 536 entity-draws.
    pub fn orbit_cull_shard(ctx: *mut OrbitCtx, cull_info: *const c_void, entity_bufs: *const OrbitEntityCullBufs,
                            draw_first: u32, draw_count: u32, meshlet_bufs: *const OrbitMeshletCullBufs,
                            record_buffer: *mut c_void, record_capacity: u32, with_commands: u32,
                            stream: *mut c_void) -> i32;
    pub fn orbit_ctx_shard_culls(ctx: *const OrbitCtx) -> u64;
    /// north_star's transport, device-only: ONE ncclAllGather of the ranks' fixed-capacity list segments, then one
    /// launch that compacts them into the contiguous rank-ordered list (no count on the host, no stream wait).
    pub fn orbit_allgather_list(ctx: *mut OrbitCtx, nccl_comm: *mut c_void, rank: u32, world: u32,
                                local_list: *const c_void, segment_capacity: u32, segments: *mut c_void,
                                out_list: *mut c_void, out_capacity: u32, header_bytes: u32, stride: u32,
                                stream: *mut c_void) -> i32;
    pub fn orbit_compact_segments(ctx: *mut OrbitCtx, segments: *const c_void, world: u32, segment_capacity: u32,
                                  out_list: *mut c_void, out_capacity: u32, header_bytes: u32, stride: u32,
                                  stream: *mut c_void) -> i32;
}

/// Turns a non-zero status into the panic the Vulkan path produced (assert!/unwrap, draw_gen.rs:247).
pub fn check(ctx: *const OrbitCtx, rc: i32) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(orbit_last_error(ctx)) }.to_string_lossy().into_owned();
        panic!("orbit_hip error {rc}: {msg}");
    }
}

// --- body swap in src/passes/draw_gen.rs::create_meshlet_dispatch_command (lines 354-377) -----------------
//
//     let gpu_cull_info_data = cull_info.to_gpu(context);                       // unchanged (draw_gen.rs:347)
//     context.add_pass(format!("{draw_commands_name}_entity_culling_hip"))
//         .with_dependency(meshlet_dispatch_buffer, AccessKind::ComputeShaderWrite)
//         .record_custom(move |_cmd, graph| {
//             let bufs = OrbitEntityCullBufs {
//                 entity_draw_buffer: graph.hip_ptr(scene.entity_draw_buffer),   // VK_KHR_external_memory_fd
//                 mesh_info_buffer: graph.hip_ptr(assets.mesh_info_buffer),      //   import, SURVEY.md §8f rank 1
//                 meshlet_dispatch_buffer: graph.hip_ptr_mut(meshlet_dispatch_buffer),
//                 entity_buffer: graph.hip_ptr(scene.entity_buffer),
//                 visibility_buffer: visibility.map_or(null_mut(), |b| graph.hip_ptr_mut(b) as *mut u32),
//                 depth_pyramid: pyramid.map_or(null(), |i| graph.hip_image_ptr(i)),
//                 depth_pyramid_size: pyramid_size, dispatch_capacity: MAX_MESHLET_DISPATCH_COUNT as u32, _pad: 0 };
//             check(ctx, unsafe { orbit_entity_cull(ctx, bytes_of(&gpu_cull_info_data).as_ptr() as _, &bufs,
//                                                   scene.entity_draw_count as u32, hip_stream) });
//         });
//     // the fill_buffer clears (draw_gen.rs:356-363) disappear: the library writes the {n,1,1} header itself.

// --- body swap in src/passes/draw_gen.rs::create_draw_commands (lines 239-325: both stages) -----------------------
//
// ONE pass, ONE library call; the library runs a scene of the renderer's size (<= 16 384 entity-draws per view) as ONE
// launch and a larger one as the launch chain, with the same buffers either way:
//
//     context.add_pass(format!("{draw_commands_name}_culling_hip"))
//         .with_dependency(meshlet_dispatch_buffer, AccessKind::ComputeShaderWrite)
//         .with_dependency(draw_commands_buffer, AccessKind::ComputeShaderWrite)
//         .record_custom(move |_cmd, graph| {
//             let view = OrbitCullView { cull_info: bytes_of(&gpu_cull_info_data).as_ptr() as _,
//                                        entity: entity_bufs(graph), meshlet: meshlet_bufs(graph),  // as above / below
//                                        entity_draw_count: scene.entity_draw_count as u32, skip_meshlet_stage: 0 };
//             check(ctx, unsafe { orbit_cull_views(ctx, &view, 1, hip_stream) });
//         });
//
// The frame's five independent culls (forward.rs:286-298 + shadow_renderer.rs:391-403) become one pass the same way:
// an array of five OrbitCullView and count = 5 — one launch for all of them.

// --- optional: derived meshlet streams, in src/assets/mod.rs ---------------------------------------------------
//
// GpuAssets::new (after meshlet_buffer is created, assets/mod.rs:272-276):
//     let mut ms = std::ptr::null_mut();
//     check(ctx, unsafe { orbit_meshlet_stream_create(ctx, 0, MAX_MESHLET_COUNT as u64, &mut ms) });
//     // once per cull context (frame slot): orbit_ctx_bind_meshlet_stream(slot_ctx, ms)
//
// GpuAssets::add_mesh, right after the upload (assets/mod.rs:441-445):
//     context.queue_write_buffer(&self.meshlet_buffer, meshlet_range.start * size_of::<GpuMeshlet>(), meshlet_bytes);
//     // ordered behind that write like any reader of the buffer (same HIP stream / the imported semaphore):
//     check(ctx, unsafe { orbit_meshlet_stream_update(ctx, ms, hip_ptr(&self.meshlet_buffer),
//                                                     meshlet_range.start as u64, meshlets.len() as u64, hip_stream) });
//
// GpuAssets::add_material, right after the upload (assets/mod.rs:520-526):
//     check(ctx, unsafe { orbit_meshlet_stream_set_materials(ctx, ms, hip_ptr(&self.materials_buffer),
//                                                            self.material_indices.len() as u32, hip_stream) });
//
// Nothing else changes: orbit_meshlet_cull keeps taking bufs.meshlet_buffer; when it is the pointer the stream was
// updated from, passes 0 and 2 read the stream (22 B per meshlet instead of 32 — 20.25 when bufs.material_buffer is
// the pointer of set_materials — no survivor payload, commands derived from the 2-B count chain).  A meshlet the
// stream does not cover latches ORBIT_E_RANGE (orbit_ctx_status) instead of being read out of range; a debug build of
// the renderer creates its contexts with caps.validate_streams = 1 and gets ORBIT_E_STALE where an update is missing.
