"""Test helper: an Engine whose entity_cull + meshlet_cull pairs run as ONE call of orbit_cull_views on the one-launch
path (orbit_amd/csrc/cull_fused.hip; OrbitCaps.cull_path = 2).  The whole parity suite of tests/test_gpu_parity.py
runs once more through it, so every scenario there is also a bit-exactness check of the one-launch cull against the
oracle."""
from orbit_amd.engine import Engine


class FusedEngine(Engine):
    """entity_cull runs at once (tests that look at the dispatch records alone see them) and is remembered; the
    meshlet_cull that follows on the same dispatch buffer puts the visibility bits back to what they were before the
    entity stage wrote them (pass 2) and runs BOTH stages again as one launch, over the same buffers."""

    def __init__(self, *a, **kw):
        kw.setdefault("cull_path", 2)
        super().__init__(*a, **kw)
        self._pending = None

    def entity_cull(self, cull_info, entity_draw_buffer, mesh_info_buffer, meshlet_dispatch_buffer, entity_buffer,
                    entity_draw_count, dispatch_capacity, visibility_buffer=None, depth_pyramid=None,
                    depth_pyramid_size=(0, 0), draw_first=None, stream=None, depth_pyramid_levels=None):
        self._pending = None
        whole = draw_first is None and entity_draw_count > 0 and depth_pyramid_levels is None \
            and not isinstance(meshlet_dispatch_buffer, int)
        if whole:
            import torch

            with torch.cuda.stream(stream) if stream is not None else _null():
                saved = None if visibility_buffer is None else visibility_buffer.clone()
            self._pending = dict(cull_info=cull_info, entity_draw_buffer=entity_draw_buffer,
                                 mesh_info_buffer=mesh_info_buffer, meshlet_dispatch_buffer=meshlet_dispatch_buffer,
                                 entity_buffer=entity_buffer, entity_draw_count=entity_draw_count,
                                 dispatch_capacity=dispatch_capacity, visibility_buffer=visibility_buffer,
                                 depth_pyramid=depth_pyramid, depth_pyramid_size=depth_pyramid_size, _saved=saved)
        super().entity_cull(cull_info, entity_draw_buffer, mesh_info_buffer, meshlet_dispatch_buffer, entity_buffer,
                            entity_draw_count, dispatch_capacity, visibility_buffer=visibility_buffer,
                            depth_pyramid=depth_pyramid, depth_pyramid_size=depth_pyramid_size, draw_first=draw_first,
                            stream=stream, depth_pyramid_levels=depth_pyramid_levels)

    def meshlet_cull(self, cull_info, meshlet_dispatch_buffer, meshlet_buffer, draw_commands_buffer, entity_buffer,
                     material_buffer, dispatch_capacity, draw_capacity, meshlet_visibility_buffer=None,
                     depth_pyramid=None, depth_pyramid_size=(0, 0), material_count=0, stream=None, task_records=None,
                     depth_pyramid_levels=None, record_buffer=None, record_capacity=None):
        p, self._pending = self._pending, None
        plain = task_records is None and record_buffer is None and depth_pyramid_levels is None
        if p is None or not plain or p["meshlet_dispatch_buffer"] is not meshlet_dispatch_buffer \
                or p["dispatch_capacity"] != dispatch_capacity or isinstance(meshlet_buffer, int):
            return super().meshlet_cull(cull_info, meshlet_dispatch_buffer, meshlet_buffer, draw_commands_buffer,
                                        entity_buffer, material_buffer, dispatch_capacity, draw_capacity,
                                        meshlet_visibility_buffer=meshlet_visibility_buffer, depth_pyramid=depth_pyramid,
                                        depth_pyramid_size=depth_pyramid_size, material_count=material_count,
                                        stream=stream, task_records=task_records,
                                        depth_pyramid_levels=depth_pyramid_levels, record_buffer=record_buffer,
                                        record_capacity=record_capacity)
        import torch

        saved = p.pop("_saved")
        with torch.cuda.stream(stream) if stream is not None else _null():
            if saved is not None:
                p["visibility_buffer"].copy_(saved)
        view = dict(p, meshlet_buffer=meshlet_buffer, draw_commands_buffer=draw_commands_buffer,
                    material_buffer=material_buffer, draw_capacity=draw_capacity,
                    meshlet_visibility_buffer=meshlet_visibility_buffer, material_count=material_count)
        if depth_pyramid is not None:
            view.update(depth_pyramid=depth_pyramid, depth_pyramid_size=depth_pyramid_size)
        before = self.fused_culls()
        self.cull_views([view], stream=stream)
        assert self.fused_culls() == before + 1, "the one-launch path was not taken"


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
