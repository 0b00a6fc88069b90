"""Test helper: an Engine whose meshlet culls run from a derived meshlet stream (include/orbit_abi.h, "Derived meshlet
streams").  The whole parity suite of tests/test_gpu_parity.py runs a second time through it, so every scenario
there is also a bit-exactness check of the stream path against the oracle."""
from orbit_amd.engine import Engine, _host_bytes


class StreamEngine(Engine):
    """Before every meshlet cull the stream of the call's meshlet buffer is (re)built and bound.  Rebuilt every time:
    the caching allocator hands the same address to different test scenes."""

    def __init__(self, *a, classes=False, **kw):
        """classes: the stream also derives its alpha classes from the call's material buffer
        (orbit_meshlet_stream_set_materials) and the culls must then evaluate from them."""
        super().__init__(*a, **kw)
        self._ms, self._ms_key, self._classes = None, None, classes

    def _set_materials(self, material_buffer, stream):
        if not self._classes or self._ms is None:
            return
        if material_buffer is None or isinstance(material_buffer, int):
            self._ms.set_materials(None, 0, stream)
        else:
            n = material_buffer.numel() * material_buffer.element_size() // 80
            self._ms.set_materials(material_buffer, n, stream)

    def meshlet_cull(self, cull_info, meshlet_dispatch_buffer, meshlet_buffer, *a, **kw):
        material_buffer = a[2] if len(a) > 2 else kw.get("material_buffer")
        if isinstance(meshlet_buffer, int):  # a shard under global indices: extent unknown here -> plain path
            self.bind_meshlet_stream(None)
        else:
            count = meshlet_buffer.numel() * meshlet_buffer.element_size() // 32
            key = (meshlet_buffer.data_ptr(), count)
            if count == 0:
                self.bind_meshlet_stream(None)
            elif key != self._ms_key:
                self.bind_meshlet_stream(None)
                if self._ms is not None:
                    self._ms.close()
                self._ms, self._ms_key = self.meshlet_stream(meshlet_buffer, 0, count, kw.get("stream")), key
                self._set_materials(material_buffer, kw.get("stream"))
                self.bind_meshlet_stream(self._ms)
            else:
                self._set_materials(material_buffer, kw.get("stream"))  # before the update: it derives its range's classes
                self._ms.update(meshlet_buffer, stream=kw.get("stream"))
                self.bind_meshlet_stream(self._ms)
        before, cbefore = self.meshlet_stream_culls(), self.meshlet_class_culls()
        import torch

        capturing = torch.cuda.is_current_stream_capturing()
        if self._classes and not capturing:  # the library trusts the classes once the launch that derived them has finished
            torch.cuda.synchronize()
        super().meshlet_cull(cull_info, meshlet_dispatch_buffer, meshlet_buffer, *a, **kw)
        # the library must have taken the stream path wherever the header says it does (passes 0 and 2)
        occlusion_pass = int(_host_bytes(cull_info, 400)[332:336].view("<u4")[0])
        expect = self._ms is not None and not isinstance(meshlet_buffer, int) and count != 0 and occlusion_pass != 1
        assert (self.meshlet_stream_culls() - before == 1) == expect, "stream path taken / not taken unexpectedly"
        if self._classes and expect and not capturing and material_buffer is not None and not isinstance(material_buffer, int):
            assert self.meshlet_class_culls() - cbefore == 1, "the alpha classes were not used"
