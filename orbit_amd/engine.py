"""Thin object wrapper over the C ABI for callers that hold torch tensors.

torch is plumbing here (device memory + streams); every method forwards raw
device pointers to ``liborbit_cull.so`` and enqueues on the current HIP stream.
Argument order follows the push-constant order of the reference shaders, as in
``include/orbit_abi.h``.
"""
import ctypes as C

import numpy as np

from . import _lib, layouts


def _ptr(t):
    """Device pointer of a torch tensor, or a raw integer address (used to pass a
    shard of a large buffer under its global indices: base = shard_ptr - first * stride)."""
    if t is None:
        return None
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())


def _stream(stream):
    import torch

    s = torch.cuda.current_stream() if stream is None else stream
    return C.c_void_p(s.cuda_stream)


def _host_bytes(x, nbytes):
    a = np.ascontiguousarray(x).view(np.uint8).reshape(-1)
    if a.nbytes != nbytes:
        raise ValueError(f"expected a {nbytes}-byte host block, got {a.nbytes}")
    return a


def depth_pyramid_desc(screen_width, screen_height):
    """DepthPyramid::new geometry (src/passes/draw_gen.rs:457-459)."""
    d = _lib.DepthPyramidDesc()
    _lib.check(_lib.load().orbit_depth_pyramid_desc(screen_width, screen_height, C.byref(d)))
    return d


def shard_range(entity_draw_count, rank, world):
    b, e = C.c_uint32(), C.c_uint32()
    _lib.load().orbit_shard_range(entity_draw_count, rank, world, C.byref(b), C.byref(e))
    return b.value, e.value


class MeshletStream:
    """Handle of an OrbitMeshletStream (include/orbit_abi.h, "Derived meshlet streams")."""

    def __init__(self, engine, meshlet_buffer, first, count, stream=None):
        self._engine, self._lib = engine, engine._lib
        self._h = C.c_void_p()
        self.first, self.capacity = int(first), int(count)
        _lib.check(self._lib.orbit_meshlet_stream_create(engine._ctx, self.first, self.capacity, C.byref(self._h)),
                   engine._ctx)
        self.update(meshlet_buffer, first, count, stream)

    def update(self, meshlet_buffer, first=None, count=None, stream=None):
        first = self.first if first is None else int(first)
        count = self.capacity if count is None else int(count)
        _lib.check(self._lib.orbit_meshlet_stream_update(self._engine._ctx, self._h, _ptr(meshlet_buffer), first, count,
                                                         _stream(stream)), self._engine._ctx)

    def set_materials(self, material_buffer, material_count, stream=None):
        """orbit_meshlet_stream_set_materials: alpha classes from material_buffer[material_index].alpha_mode
        (None forgets them).  Culls whose material buffer is this one then read no material index."""
        _lib.check(self._lib.orbit_meshlet_stream_set_materials(self._engine._ctx, self._h, _ptr(material_buffer),
                                                                int(material_count), _stream(stream)), self._engine._ctx)

    def update_meshes(self, mesh_info_buffer, first_mesh, count, stream=None):
        """orbit_meshlet_stream_update_meshes: 32-B side entries of meshes [first_mesh, first_mesh + count) of
        `mesh_info_buffer` (None forgets the table); entity culls of that buffer then read them instead of the MeshInfos."""
        _lib.check(self._lib.orbit_meshlet_stream_update_meshes(self._engine._ctx, self._h, _ptr(mesh_info_buffer),
                                                                int(first_mesh), int(count), _stream(stream)),
                   self._engine._ctx)

    def validate(self, meshlet_buffer, material_buffer=None, stream=None):
        """orbit_meshlet_stream_validate: ORBIT_E_STALE is latched (Engine.status raises) if the stream differs."""
        _lib.check(self._lib.orbit_meshlet_stream_validate(self._engine._ctx, self._h, _ptr(meshlet_buffer),
                                                           _ptr(material_buffer), _stream(stream)), self._engine._ctx)

    def close(self):
        """orbit_meshlet_stream_destroy.  The library refuses while a context still has the stream bound; the engine
        this handle came from is unbound here, other engines are the caller's to unbind first."""
        if getattr(self, "_h", None) is not None and self._h.value:
            if getattr(self._engine, "_meshlet_stream", None) is self and getattr(self._engine, "_ctx", None):
                self._engine.bind_meshlet_stream(None)
            _lib.check(self._lib.orbit_meshlet_stream_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PreparedShardCull:
    """Engine.prepare_cull_shard: orbit_cull_shard's argument blocks, kept alive and reusable frame after frame."""

    def __init__(self, engine, cull_info, entity_draw_buffer, mesh_info_buffer, meshlet_dispatch_buffer, entity_buffer,
                 draw_first, draw_count, dispatch_capacity, meshlet_buffer, material_buffer, record_buffer, record_capacity,
                 draw_commands_buffer, draw_capacity, material_count, visibility_buffer, meshlet_visibility_buffer,
                 depth_pyramid, depth_pyramid_size, stream):
        self.cull_info = _host_bytes(cull_info, 400).copy()  # the call reads THIS buffer: update it in place per frame
        e, m = _lib.EntityCullBufs(), _lib.MeshletCullBufs()
        e.entity_draw_buffer = _ptr(entity_draw_buffer)
        e.mesh_info_buffer = _ptr(mesh_info_buffer)
        e.meshlet_dispatch_buffer = m.meshlet_dispatch_buffer = _ptr(meshlet_dispatch_buffer)
        e.entity_buffer = m.entity_buffer = _ptr(entity_buffer)
        e.visibility_buffer = _ptr(visibility_buffer)
        e.depth_pyramid = m.depth_pyramid = _ptr(depth_pyramid)
        e.depth_pyramid_size[0], e.depth_pyramid_size[1] = depth_pyramid_size
        m.depth_pyramid_size[0], m.depth_pyramid_size[1] = depth_pyramid_size
        e.dispatch_capacity = m.dispatch_capacity = dispatch_capacity
        m.meshlet_buffer = _ptr(meshlet_buffer)
        m.draw_commands_buffer = _ptr(draw_commands_buffer)
        m.material_buffer = _ptr(material_buffer)
        m.meshlet_visibility_buffer = _ptr(meshlet_visibility_buffer)
        m.draw_capacity = draw_capacity
        m.material_count = material_count
        self._fn, self._ctx = engine._lib.orbit_cull_shard, engine._ctx
        self._cip, self._ep, self._mp = self.cull_info.ctypes.data_as(C.c_void_p), C.byref(e), C.byref(m)
        self._rec, self._st = _ptr(record_buffer), _stream(stream)
        self._args = (int(draw_first), int(draw_count), int(record_capacity), 1 if draw_commands_buffer is not None else 0)
        self._keep = (e, m, stream, engine)  # the blocks (and the stream object) live as long as the call does

    def set_cull_info(self, cull_info):
        self.cull_info[:] = _host_bytes(cull_info, 400)

    def __call__(self):
        first, count, cap, with_cmds = self._args
        rc = self._fn(self._ctx, self._cip, self._ep, first, count, self._mp, self._rec, cap, with_cmds, self._st)
        if rc != _lib.OK:
            _lib.check(rc, self._ctx)


class Engine:
    """One ``OrbitCtx``: scan scratch sized from ``caps`` on one gfx950 device."""

    def __init__(self, device_index=0, _library=None, **caps):
        lib = _lib.load() if _library is None else _library  # _library: a variant build (tools/ab_libs.py)
        c = _lib.Caps()
        lib.orbit_default_caps(C.byref(c))
        for k, v in caps.items():
            if not hasattr(c, k):
                raise TypeError(f"unknown capacity {k}")
            setattr(c, k, int(v))
        self.caps = c
        self._ctx = C.c_void_p()
        _lib.check(lib.orbit_ctx_create(device_index, C.byref(c), C.byref(self._ctx)))
        self._lib = lib

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.orbit_ctx_destroy(self._ctx)  # unbinds its meshlet stream
            self._ctx = C.c_void_p()
            self._meshlet_stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def status(self, stream=None, sync=True):
        """Raises OrbitError for a device-latched condition (capacity overflow, timeout)."""
        _lib.check(self._lib.orbit_ctx_status(self._ctx, _stream(stream), 1 if sync else 0), self._ctx)

    def profile(self, enable=True, every=1):
        """HIP-event timing of the dominant kernel of meshlet_cull (measurement hook); every n-th call is timed."""
        _lib.check(self._lib.orbit_ctx_profile(self._ctx, int(every) if enable else 0), self._ctx)

    def profile_reserve(self, pairs, stream=None):
        """orbit_ctx_profile_reserve: the hook's event pairs created (and recorded once on `stream`) before a timed region."""
        _lib.check(self._lib.orbit_ctx_profile_reserve(self._ctx, int(pairs), _stream(stream)), self._ctx)

    def profile_read(self):
        ms, n = C.c_float(), C.c_uint32()
        _lib.check(self._lib.orbit_ctx_profile_read(self._ctx, C.byref(ms), C.byref(n)), self._ctx)
        return ms.value, n.value

    # -- entity_cull: create_meshlet_dispatch_command body (draw_gen.rs:327-380)
    def entity_cull(self, cull_info, entity_draw_buffer, mesh_info_buffer, meshlet_dispatch_buffer, entity_buffer,
                    entity_draw_count, dispatch_capacity, visibility_buffer=None, depth_pyramid=None,
                    depth_pyramid_size=(0, 0), draw_first=None, stream=None, depth_pyramid_levels=None):
        ci = _host_bytes(cull_info, 400)
        b = _lib.EntityCullBufs()
        b.entity_draw_buffer = _ptr(entity_draw_buffer)
        b.mesh_info_buffer = _ptr(mesh_info_buffer)
        b.meshlet_dispatch_buffer = _ptr(meshlet_dispatch_buffer)
        b.entity_buffer = _ptr(entity_buffer)
        b.visibility_buffer = _ptr(visibility_buffer)
        b.depth_pyramid = _ptr(depth_pyramid)
        b.depth_pyramid_size[0], b.depth_pyramid_size[1] = depth_pyramid_size
        b.dispatch_capacity = dispatch_capacity
        b.depth_pyramid_levels = _ptr(depth_pyramid_levels)  # DEVICE array of OrbitDepthPyramidLevel, or None
        if draw_first is None:
            rc = self._lib.orbit_entity_cull(self._ctx, ci.ctypes.data_as(C.c_void_p), C.byref(b), entity_draw_count,
                                             _stream(stream))
        else:
            rc = self._lib.orbit_entity_cull_range(self._ctx, ci.ctypes.data_as(C.c_void_p), C.byref(b), draw_first,
                                                   entity_draw_count, _stream(stream))
        _lib.check(rc, self._ctx)

    # -- several views side by side (orbit_cull_views)
    def cull_views(self, views, stream=None):
        """views: list of dicts with the arguments of entity_cull + meshlet_cull for one view each:
        cull_info, entity_draw_buffer, mesh_info_buffer, meshlet_dispatch_buffer, entity_buffer, entity_draw_count,
        dispatch_capacity, meshlet_buffer, draw_commands_buffer, material_buffer, draw_capacity and optionally
        visibility_buffer, meshlet_visibility_buffer, depth_pyramid, depth_pyramid_size, material_count,
        skip_meshlet_stage."""
        arr, keep = self.prepare_views(views)
        self.cull_views_prepared(arr, stream)
        del keep

    def cull_views_prepared(self, arr, stream=None):
        """orbit_cull_views on a view array built once by prepare_views (per-frame callers keep it)."""
        _lib.check(self._lib.orbit_cull_views(self._ctx, arr, len(arr), _stream(stream)), self._ctx)

    @staticmethod
    def prepare_views(views):
        """-> (ctypes OrbitCullView array, objects that must outlive it)."""
        arr = (_lib.CullView * len(views))()
        keep = []
        for v, a in zip(views, arr):
            ci = _host_bytes(v["cull_info"], 400)
            keep.append(ci)
            a.cull_info = ci.ctypes.data
            e, m = a.entity, a.meshlet
            e.entity_draw_buffer = _ptr(v["entity_draw_buffer"])
            e.mesh_info_buffer = _ptr(v["mesh_info_buffer"])
            e.meshlet_dispatch_buffer = m.meshlet_dispatch_buffer = _ptr(v["meshlet_dispatch_buffer"])
            e.entity_buffer = m.entity_buffer = _ptr(v["entity_buffer"])
            e.visibility_buffer = _ptr(v.get("visibility_buffer"))
            e.depth_pyramid = m.depth_pyramid = _ptr(v.get("depth_pyramid"))
            ps = v.get("depth_pyramid_size", (0, 0))
            e.depth_pyramid_size[0], e.depth_pyramid_size[1] = ps
            m.depth_pyramid_size[0], m.depth_pyramid_size[1] = ps
            e.dispatch_capacity = m.dispatch_capacity = v["dispatch_capacity"]
            m.meshlet_buffer = _ptr(v.get("meshlet_buffer"))
            m.draw_commands_buffer = _ptr(v.get("draw_commands_buffer"))
            m.material_buffer = _ptr(v.get("material_buffer"))
            m.meshlet_visibility_buffer = _ptr(v.get("meshlet_visibility_buffer"))
            m.draw_capacity = v.get("draw_capacity", 0)
            m.material_count = v.get("material_count", 0)
            a.entity_draw_count = v["entity_draw_count"]
            a.skip_meshlet_stage = 1 if v.get("skip_meshlet_stage") else 0
        return arr, keep

    # -- meshlet_cull: create_meshlet_draw_commands body (draw_gen.rs:382-435)
    def meshlet_cull(self, cull_info, meshlet_dispatch_buffer, meshlet_buffer, draw_commands_buffer, entity_buffer,
                     material_buffer, dispatch_capacity, draw_capacity, meshlet_visibility_buffer=None,
                     depth_pyramid=None, depth_pyramid_size=(0, 0), material_count=0, stream=None, task_records=None,
                     depth_pyramid_levels=None, record_buffer=None, record_capacity=None):
        ci = _host_bytes(cull_info, 400)
        b = _lib.MeshletCullBufs()
        b.meshlet_dispatch_buffer = _ptr(meshlet_dispatch_buffer)
        b.meshlet_buffer = _ptr(meshlet_buffer)
        b.draw_commands_buffer = _ptr(draw_commands_buffer)
        b.entity_buffer = _ptr(entity_buffer)
        b.material_buffer = _ptr(material_buffer)
        b.meshlet_visibility_buffer = _ptr(meshlet_visibility_buffer)
        b.depth_pyramid = _ptr(depth_pyramid)
        b.depth_pyramid_size[0], b.depth_pyramid_size[1] = depth_pyramid_size
        b.dispatch_capacity = dispatch_capacity
        b.draw_capacity = draw_capacity
        b.material_count = material_count
        b.depth_pyramid_levels = _ptr(depth_pyramid_levels)
        if record_buffer is not None and record_capacity is not None:
            # sharded engine: the record list AND this rank's own commands (draw_commands_buffer / draw_capacity)
            _lib.check(self._lib.orbit_meshlet_cull_records_and_commands(
                self._ctx, ci.ctypes.data_as(C.c_void_p), C.byref(b), _ptr(record_buffer), record_capacity,
                _stream(stream)), self._ctx)
            return
        if record_buffer is not None:  # sharded engine: the visible list at record granularity (12 B per record)
            _lib.check(self._lib.orbit_meshlet_cull_visible_records(self._ctx, ci.ctypes.data_as(C.c_void_p), C.byref(b),
                                                                    _ptr(record_buffer), draw_capacity, _stream(stream)),
                       self._ctx)
            return
        if task_records is not None:
            _lib.check(self._lib.orbit_meshlet_task_cull(self._ctx, ci.ctypes.data_as(C.c_void_p), C.byref(b),
                                                         _ptr(task_records), _stream(stream)), self._ctx)
            return
        _lib.check(self._lib.orbit_meshlet_cull(self._ctx, ci.ctypes.data_as(C.c_void_p), C.byref(b),
                                                _stream(stream)), self._ctx)

    def meshlet_task_cull(self, cull_info, meshlet_dispatch_buffer, meshlet_buffer, task_records, entity_buffer,
                          material_buffer, dispatch_capacity, **kw):
        """Mesh-shading path (forward_depth_prepass.task): one 44-B OrbitMeshTaskRecord per dispatch record."""
        self.meshlet_cull(cull_info, meshlet_dispatch_buffer, meshlet_buffer, None, entity_buffer, material_buffer,
                          dispatch_capacity, 0, task_records=task_records, **kw)

    # -- DepthPyramid::update (draw_gen.rs:510-566)
    def depth_reduce(self, depth, screen_width, screen_height, pyramid, stream=None):
        _lib.check(self._lib.orbit_depth_reduce(self._ctx, _ptr(depth), screen_width, screen_height, _ptr(pyramid),
                                                _stream(stream)), self._ctx)

    def depth_reduce_multi(self, items, stream=None):
        """orbit_depth_reduce_multi (update_multiple_depth_pyramids::<C>, draw_gen.rs:569-628): one launch pair for up
        to 8 pyramids.  items: dicts with depth, width, height and either pyramid (packed chain) or levels (list of
        (tensor-or-pointer, row_pitch) per mip: separate per-mip images); optional depth_row_pitch."""
        arr, keep = self.prepare_depth_reduce_items(items)
        _lib.check(self._lib.orbit_depth_reduce_multi(self._ctx, arr, len(items), _stream(stream)), self._ctx)

    def prepare_depth_reduce_items(self, items):
        """-> (ctypes OrbitDepthReduceItem array, objects that must outlive it); items as for depth_reduce_multi."""
        arr = (_lib.DepthReduceItem * len(items))()
        keep = []
        for it, a in zip(items, arr):
            a.depth = _ptr(it["depth"])
            a.screen_width, a.screen_height = it["width"], it["height"]
            a.depth_row_pitch = it.get("depth_row_pitch", 0)
            a.pyramid = _ptr(it.get("pyramid"))
            if it.get("levels") is not None:
                lv = (_lib.DepthPyramidLevel * len(it["levels"]))()
                for l, (t, pitch) in zip(lv, it["levels"]):
                    l.texels, l.row_pitch = _ptr(t), pitch
                keep.append(lv)
                a.levels = lv
        return arr, keep

    def prepare_frame_late(self, pyramids=(), late_views=(), cascade_views=(), clusters=None):
        """orbit_frame_late's descriptor, built once (a per-frame caller keeps it): `pyramids` as for depth_reduce_multi,
        `late_views` / `cascade_views` as for cull_views, `clusters` a dict with compute_clusters' arguments (push, info,
        depth, lights, tile_depth_slice_mask, depth_bounds, unique_cluster_buffer, index_capacity, light_index_buffer,
        light_index_capacity, cluster_offset_image).  -> (descriptor, objects that must outlive it)."""
        f, keep = _lib.FrameLate(), []
        if pyramids:
            arr, k = self.prepare_depth_reduce_items(list(pyramids))
            f.pyramids, f.pyramid_count = arr, len(arr)
            keep += [arr, k]
        for name, views in (("late_views", late_views), ("cascade_views", cascade_views)):
            if views:
                arr, k = self.prepare_views(list(views))
                setattr(f, name, arr)
                setattr(f, name[:-1] + "_count", len(arr))
                keep += [arr, k]
        if clusters is not None:
            c = _lib.ClusterFrame()
            pc = _host_bytes(clusters["push"], layouts.MARK_ACTIVE_PUSH.itemsize).copy()
            ib = _host_bytes(clusters["info"], layouts.CLUSTER_CULL_INFO.itemsize).copy()
            c.push, c.info = pc.ctypes.data, ib.ctypes.data
            for k in ("depth", "lights", "tile_depth_slice_mask", "depth_bounds", "unique_cluster_buffer",
                      "light_index_buffer", "cluster_offset_image"):
                setattr(c, k, _ptr(clusters.get(k)))
            c.index_capacity, c.light_index_capacity = clusters["index_capacity"], clusters["light_index_capacity"]
            f.clusters = C.pointer(c)
            keep += [c, pc, ib]
        return f, keep

    def frame_late(self, frame, stream=None):
        """orbit_frame_late on a descriptor of prepare_frame_late: {pyramids -> late culls} || {cascade culls} ||
        {compute_clusters}, forked behind `stream` and joined into it."""
        _lib.check(self._lib.orbit_frame_late(self._ctx, C.byref(frame), _stream(stream)), self._ctx)

    def meshlet_cull_visible_records(self, cull_info, meshlet_dispatch_buffer, meshlet_buffer, record_buffer,
                                     entity_buffer, material_buffer, dispatch_capacity, record_capacity, **kw):
        """orbit_meshlet_cull_visible_records: {records, survivors} + 12-B {entity_index, meshlet_offset, mask} per
        dispatch record, in record order (mask 0: no survivor), written by the evaluation launch itself."""
        self.meshlet_cull(cull_info, meshlet_dispatch_buffer, meshlet_buffer, None, entity_buffer, material_buffer,
                          dispatch_capacity, record_capacity, record_buffer=record_buffer, **kw)

    def meshlet_cull_records_and_commands(self, cull_info, meshlet_dispatch_buffer, meshlet_buffer, record_buffer,
                                          draw_commands_buffer, entity_buffer, material_buffer, dispatch_capacity,
                                          record_capacity, draw_capacity, **kw):
        """orbit_meshlet_cull_records_and_commands: one evaluation, the 12-B record list and the 28-B commands."""
        self.meshlet_cull(cull_info, meshlet_dispatch_buffer, meshlet_buffer, draw_commands_buffer, entity_buffer,
                          material_buffer, dispatch_capacity, draw_capacity, record_buffer=record_buffer,
                          record_capacity=record_capacity, **kw)

    def cull_shard(self, cull_info, entity_draw_buffer, mesh_info_buffer, meshlet_dispatch_buffer, entity_buffer,
                   draw_first, draw_count, dispatch_capacity, meshlet_buffer, material_buffer, record_buffer,
                   record_capacity, draw_commands_buffer=None, draw_capacity=0, material_count=0,
                   visibility_buffer=None, meshlet_visibility_buffer=None, depth_pyramid=None,
                   depth_pyramid_size=(0, 0), stream=None):
        """orbit_cull_shard: a rank's whole cull — entity range, meshlet stage, record list (and, with
        draw_commands_buffer, the rank's own 28-B commands) — as one call; one launch (+ the emit) for pass 0 and up to
        65 536 entity-draws."""
        ci = _host_bytes(cull_info, 400)
        e, m = _lib.EntityCullBufs(), _lib.MeshletCullBufs()
        e.entity_draw_buffer = _ptr(entity_draw_buffer)
        e.mesh_info_buffer = _ptr(mesh_info_buffer)
        e.meshlet_dispatch_buffer = m.meshlet_dispatch_buffer = _ptr(meshlet_dispatch_buffer)
        e.entity_buffer = m.entity_buffer = _ptr(entity_buffer)
        e.visibility_buffer = _ptr(visibility_buffer)
        e.depth_pyramid = m.depth_pyramid = _ptr(depth_pyramid)
        e.depth_pyramid_size[0], e.depth_pyramid_size[1] = depth_pyramid_size
        m.depth_pyramid_size[0], m.depth_pyramid_size[1] = depth_pyramid_size
        e.dispatch_capacity = m.dispatch_capacity = dispatch_capacity
        m.meshlet_buffer = _ptr(meshlet_buffer)
        m.draw_commands_buffer = _ptr(draw_commands_buffer)
        m.material_buffer = _ptr(material_buffer)
        m.meshlet_visibility_buffer = _ptr(meshlet_visibility_buffer)
        m.draw_capacity = draw_capacity
        m.material_count = material_count
        _lib.check(self._lib.orbit_cull_shard(self._ctx, ci.ctypes.data_as(C.c_void_p), C.byref(e), draw_first, draw_count,
                                              C.byref(m), _ptr(record_buffer), record_capacity,
                                              1 if draw_commands_buffer is not None else 0, _stream(stream)), self._ctx)

    def prepare_cull_shard(self, cull_info, entity_draw_buffer, mesh_info_buffer, meshlet_dispatch_buffer, entity_buffer,
                           draw_first, draw_count, dispatch_capacity, meshlet_buffer, material_buffer, record_buffer,
                           record_capacity, draw_commands_buffer=None, draw_capacity=0, material_count=0,
                           visibility_buffer=None, meshlet_visibility_buffer=None, depth_pyramid=None,
                           depth_pyramid_size=(0, 0), stream=None):
        """cull_shard with its argument blocks built ONCE: returns a PreparedShardCull — calling it enqueues the shard
        cull on `stream` (a per-frame caller whose buffers do not change — bench.py's step loop — spends its host time in
        the library, not in filling ctypes structs: ~10 us per call).  The 400-B CullInfo is the object's own copy:
        write the next frame's into `prepared.cull_info` (a uint8[400] numpy view, `prepared.set_cull_info(ci)`) before
        the call.  `stream` is part of what is prepared and must be given (the stream current at prepare time is not
        the stream current at call time)."""
        if stream is None:
            raise ValueError("prepare_cull_shard: pass the stream the prepared call enqueues on")
        return PreparedShardCull(self, cull_info, entity_draw_buffer, mesh_info_buffer, meshlet_dispatch_buffer, entity_buffer,
                                 draw_first, draw_count, dispatch_capacity, meshlet_buffer, material_buffer, record_buffer,
                                 record_capacity, draw_commands_buffer, draw_capacity, material_count, visibility_buffer,
                                 meshlet_visibility_buffer, depth_pyramid, depth_pyramid_size, stream)

    def mesh_side_culls(self):
        """orbit_ctx_mesh_side_culls: entity culls of the bound stream that were handed its mesh side table."""
        return int(self._lib.orbit_ctx_mesh_side_culls(self._ctx))

    def shard_culls(self):
        """orbit_ctx_shard_culls: orbit_cull_shard calls of this context that took the one launch."""
        return int(self._lib.orbit_ctx_shard_culls(self._ctx))

    def expand_visible_records(self, record_buffer, meshlet_buffer, draw_commands_buffer, draw_capacity, stream=None):
        """orbit_expand_visible_records: record list -> MeshletDrawCommandBuffer, in list order."""
        _lib.check(self._lib.orbit_expand_visible_records(self._ctx, _ptr(record_buffer), _ptr(meshlet_buffer),
                                                          _ptr(draw_commands_buffer), draw_capacity, _stream(stream)),
                   self._ctx)

    # -- exchange without a host in the step (orbit_p2p_* / orbit_exchange_list)
    def p2p_alloc(self, nbytes):
        """(device pointer, 64-byte IPC handle) of an exchange buffer peers can map."""
        ptr, handle = C.c_void_p(), (C.c_uint8 * 64)()
        _lib.check(self._lib.orbit_p2p_alloc(self._ctx, int(nbytes), C.byref(ptr), handle), self._ctx)
        return int(ptr.value), bytes(handle)

    def p2p_free(self, ptr):
        _lib.check(self._lib.orbit_p2p_free(self._ctx, C.c_void_p(ptr)), self._ctx)

    def p2p_open(self, handle):
        ptr, h = C.c_void_p(), (C.c_uint8 * 64).from_buffer_copy(handle)
        _lib.check(self._lib.orbit_p2p_open(self._ctx, h, C.byref(ptr)), self._ctx)
        return int(ptr.value)

    def p2p_close(self, ptr):
        _lib.check(self._lib.orbit_p2p_close(self._ctx, C.c_void_p(ptr)), self._ctx)

    def meshlet_stream(self, meshlet_buffer, first, count, stream=None):
        """orbit_meshlet_stream_create + _update: the derived streams of meshlets [first, first + count) of
        `meshlet_buffer` (a device pointer / tensor indexed by GLOBAL meshlet index, like bufs.meshlet_buffer).
        Bind the result with bind_meshlet_stream; call .update() after writing meshlets."""
        return MeshletStream(self, meshlet_buffer, first, count, stream)

    def bind_meshlet_stream(self, ms):
        """orbit_ctx_bind_meshlet_stream (None unbinds).  The engine keeps the stream object alive."""
        _lib.check(self._lib.orbit_ctx_bind_meshlet_stream(self._ctx, ms._h if ms is not None else None), self._ctx)
        self._meshlet_stream = ms

    def fused_culls(self):
        """Views this context has culled through the one-launch path of orbit_cull_views (include/orbit_abi.h)."""
        return int(self._lib.orbit_ctx_fused_culls(self._ctx))

    def meshlet_stream_culls(self):
        """orbit_ctx_meshlet_stream_culls: meshlet culls launched from a bound stream so far."""
        return int(self._lib.orbit_ctx_meshlet_stream_culls(self._ctx))

    def meshlet_class_culls(self):
        """orbit_ctx_meshlet_class_culls: ... of which evaluated from the stream's alpha classes."""
        return int(self._lib.orbit_ctx_meshlet_class_culls(self._ctx))

    def exchange_list(self, local_list, rank, world, out_buffers, ctrl_buffers, out_capacity, header_bytes, stride,
                      stream=None):
        """orbit_exchange_list: the rank-ordered all-gather of the ranks' lists with counts and completion signalled on
        the device (no collective, no host round trip; capturable).  `out_buffers` / `ctrl_buffers`: lists of device
        pointers (ints or tensors), rank r's output buffer / control block as mapped in this process."""
        outs = (C.c_void_p * world)(*[_ptr(b) for b in out_buffers])
        ctrls = (C.c_void_p * world)(*[_ptr(b) for b in ctrl_buffers])
        _lib.check(self._lib.orbit_exchange_list(self._ctx, _ptr(local_list), rank, world, outs, ctrls, out_capacity,
                                                 header_bytes, stride, _stream(stream)), self._ctx)

    def prepare_exchange_list(self, local_list, rank, world, out_buffers, ctrl_buffers, out_capacity, header_bytes, stride,
                              stream=None):
        """exchange_list with its pointer arrays built once: returns a function of no arguments (see prepare_cull_shard)."""
        outs = (C.c_void_p * world)(*[_ptr(b) for b in out_buffers])
        ctrls = (C.c_void_p * world)(*[_ptr(b) for b in ctrl_buffers])
        fn, ctx, lst, st = self._lib.orbit_exchange_list, self._ctx, _ptr(local_list), _stream(stream)
        keep = (outs, ctrls, local_list, stream)

        def call():
            rc = fn(ctx, lst, rank, world, outs, ctrls, out_capacity, header_bytes, stride, st)
            if rc != _lib.OK:
                _lib.check(rc, ctx)
            return keep

        return call

    def allgather_list(self, nccl_comm, rank, world, local_list, segment_capacity, segments, out_list, out_capacity,
                       header_bytes, stride, stream=None):
        """orbit_allgather_list: ONE ncclAllGather of the ranks' fixed-capacity list segments + the compaction launch; no
        count is read back, no stream is synchronised (`nccl_comm`: the ncclComm_t as an integer / c_void_p)."""
        comm = nccl_comm if isinstance(nccl_comm, C.c_void_p) else C.c_void_p(int(nccl_comm))
        _lib.check(self._lib.orbit_allgather_list(self._ctx, comm, rank, world, _ptr(local_list), segment_capacity,
                                                  _ptr(segments), _ptr(out_list), out_capacity, header_bytes, stride,
                                                  _stream(stream)), self._ctx)

    def compact_segments(self, segments, world, segment_capacity, out_list, out_capacity, header_bytes, stride,
                         stream=None):
        """orbit_compact_segments: `world` gathered segments {count | header | segment_capacity items} -> the contiguous
        rank-ordered list {total | header | items}."""
        _lib.check(self._lib.orbit_compact_segments(self._ctx, _ptr(segments), world, segment_capacity, _ptr(out_list),
                                                    out_capacity, header_bytes, stride, _stream(stream)), self._ctx)

    def gather_visible(self, nccl_comm, rank, world, local_draw_buffer, out_draw_buffer, out_capacity, stream=None):
        """Rank-ordered all-gather of the visible lists over the caller's RCCL communicator
        (`nccl_comm`: the ncclComm_t as an integer / c_void_p).  One stream sync (message sizes)."""
        comm = nccl_comm if isinstance(nccl_comm, C.c_void_p) else C.c_void_p(int(nccl_comm))
        _lib.check(self._lib.orbit_gather_visible(self._ctx, comm, rank, world, _ptr(local_draw_buffer),
                                                  _ptr(out_draw_buffer), out_capacity, _stream(stream)), self._ctx)

    def compute_clusters(self, push, info, depth, lights, tile_depth_slice_mask, depth_bounds, unique_cluster_buffer,
                         index_capacity, light_index_buffer, light_index_capacity, cluster_offset_image, stream=None):
        """compute_clusters (cluster.rs:368-397): mark -> compact -> assign in one call."""
        pc = _host_bytes(push, layouts.MARK_ACTIVE_PUSH.itemsize)
        ib = _host_bytes(info, layouts.CLUSTER_CULL_INFO.itemsize)
        _lib.check(self._lib.orbit_compute_clusters(
            self._ctx, pc.ctypes.data_as(C.c_void_p), ib.ctypes.data_as(C.c_void_p), _ptr(depth), _ptr(lights),
            _ptr(tile_depth_slice_mask), _ptr(depth_bounds), _ptr(unique_cluster_buffer), index_capacity,
            _ptr(light_index_buffer), light_index_capacity, _ptr(cluster_offset_image), _stream(stream)), self._ctx)

    # -- compute_clusters stages (cluster.rs:399-591)
    def cluster_mark(self, push, depth, tile_depth_slice_mask, depth_bounds, stream=None):
        pc = _host_bytes(push, layouts.MARK_ACTIVE_PUSH.itemsize)
        _lib.check(self._lib.orbit_cluster_mark(self._ctx, pc.ctypes.data_as(C.c_void_p), _ptr(depth),
                                                _ptr(tile_depth_slice_mask), _ptr(depth_bounds), _stream(stream)),
                   self._ctx)

    def cluster_compact(self, cluster_count, tile_depth_slice_mask, unique_cluster_buffer, index_capacity,
                        stream=None):
        cc = (C.c_uint32 * 3)(*[int(v) for v in cluster_count])
        _lib.check(self._lib.orbit_cluster_compact(self._ctx, C.byref(cc), _ptr(tile_depth_slice_mask),
                                                   _ptr(unique_cluster_buffer), index_capacity, _stream(stream)),
                   self._ctx)

    def cluster_assign(self, info, unique_cluster_buffer, depth_bounds, lights, light_index_buffer,
                       light_index_capacity, cluster_offset_image, stream=None):
        ib = _host_bytes(info, layouts.CLUSTER_CULL_INFO.itemsize)
        _lib.check(self._lib.orbit_cluster_assign(self._ctx, ib.ctypes.data_as(C.c_void_p),
                                                  _ptr(unique_cluster_buffer), _ptr(depth_bounds), _ptr(lights),
                                                  _ptr(light_index_buffer), light_index_capacity,
                                                  _ptr(cluster_offset_image), _stream(stream)), self._ctx)
