"""Host-side scene producers (orbit_amd/host/orbit_scene.hpp): the generational arena, the best-fit free-list
allocator that hands out meshlet-visibility words, and SceneData::update_scene which fills EntityDrawBuffer,
EntityData[] and LightData[] — the inputs of entity_cull / meshlet_cull / light_cluster.

The arena scenarios are the situations the reference's own unit tests walk through (src/collections/arena.rs:388-505:
insert/get/len, remove twice, slot reuse after removal, has_index, iteration order), written against this binding;
the allocator is checked against an independent pure-Python restatement of freelist_alloc.rs on random traces.
"""
import numpy as np
import pytest

from orbit_amd import layouts as L
from orbit_amd import scene as S


# ----------------------------------------------------------------------------- Arena
def test_arena_insert_get_len():
    a = S.Arena()
    assert len(a) == 0
    idx = [a.insert(v) for v in (0, 1, 2)]
    assert len(a) == 3
    assert [a.get(i) for i in (idx[1], idx[0], idx[2])] == [1, 0, 2]
    assert [i.key() for i in idx] == [(1, 0), (1, 1), (1, 2)]  # first generation is 1, slots in insertion order


def test_arena_remove_is_idempotent_and_counts():
    a = S.Arena()
    i0, i1, i2 = (a.insert(v) for v in (0, 1, 2))
    for i, v, left in ((i1, 1, 2), (i0, 0, 1), (i2, 2, 0)):
        assert a.remove(i) == v and len(a) == left
        assert a.remove(i) is None and len(a) == left


def test_arena_slot_reuse_bumps_generation():
    a = S.Arena()
    i0, i1, i2 = (a.insert(v) for v in (0, 1, 2))
    assert a.remove(i1) == 1
    i3 = a.insert(3)
    assert i3.key() == (2, 1)          # most recently freed slot, next generation
    assert a.get(i1) is None            # the stale index no longer resolves
    assert a.remove(i3) == 3
    assert a.remove(i0) == 0
    i4 = a.insert(4)                    # LIFO free list: slot 0 was freed last
    assert i4.key() == (2, 0)
    i5 = a.insert(5)
    assert i5.key() == (3, 1)
    assert a.remove(i2) == 2 and a.remove(i2) is None


def test_arena_has_index_and_iteration_order():
    a = S.Arena()
    idx = [a.insert(v) for v in range(5)]
    a.remove(idx[1])
    a.remove(idx[3])
    assert [a.has_index(i) for i in idx] == [True, False, True, False, True]
    assert a.iter() == [(idx[0].key(), 0), (idx[2].key(), 2), (idx[4].key(), 4)]
    j = a.insert(7)                     # reuses slot 3 (freed last); iteration stays in slot order
    assert a.iter() == [(idx[0].key(), 0), (idx[2].key(), 2), (j.key(), 7), (idx[4].key(), 4)]


# ----------------------------------------------------------------------------- FreeListAllocator
class PyFreeList:
    """Independent restatement of freelist_alloc.rs + arena.rs slot policy in plain Python (lists and dicts)."""

    def __init__(self, size):
        self.entries = []      # slot -> [generation, block or None, next_free]
        self.first_free = None
        self._insert(dict(free=True, start=0, end=size, prev=None, next=None))

    def _insert(self, block):
        if self.first_free is not None:
            slot = self.first_free
            e = self.entries[slot]
            self.first_free = e[2]
            e[0] += 1
            e[1], e[2] = block, None
            return (e[0], slot)
        self.entries.append([1, block, None])
        return (1, len(self.entries) - 1)

    def _get(self, idx):
        if idx is None or idx[1] >= len(self.entries):
            return None
        e = self.entries[idx[1]]
        return e[1] if e[1] is not None and e[0] == idx[0] else None

    def _remove(self, idx):
        e = self.entries[idx[1]]
        b = e[1]
        e[1], e[2] = None, self.first_free
        self.first_free = idx[1]
        return b

    def allocate(self, size):
        best = None
        for slot, e in enumerate(self.entries):
            b = e[1]
            if b is not None and b["free"] and b["end"] - b["start"] >= size:
                if best is None or b["end"] - b["start"] < best[1]:
                    best = ((e[0], slot), b["end"] - b["start"])
        if best is None:
            return None
        fi = best[0]
        fb = self._get(fi)
        if fb["end"] - fb["start"] == size:
            fb["free"] = False
            return fi, (fb["start"], fb["end"])
        start = fb["start"]
        ni = self._insert(dict(free=False, start=start, end=start + size, prev=fb["prev"], next=fi))
        fb["start"] += size
        fb["prev"] = ni
        pb = self._get(self.entries[ni[1]][1]["prev"])
        if pb is not None:
            pb["next"] = ni   # the forward link the reference leaves stale (see the deviation test below)
        return ni, (start, start + size)

    def deallocate(self, idx):
        b = self._get(idx)
        if b is None:
            return
        pf = b["prev"] if (self._get(b["prev"]) or {}).get("free") else None
        nf = b["next"] if (self._get(b["next"]) or {}).get("free") else None
        b["free"] = True
        if pf is not None:
            pb = self._remove(pf)
            pp = self._get(pb["prev"])
            if pp is not None:
                pp["next"] = idx
            b["prev"], b["start"] = pb["prev"], pb["start"]
        if nf is not None:
            nb = self._remove(nf)
            nn = self._get(nb["next"])
            if nn is not None:
                nn["prev"] = idx
            b["next"], b["end"] = nb["next"], nb["end"]

    def blocks(self):
        out = [(e[1]["start"], e[1]["end"], e[1]["free"]) for e in self.entries if e[1] is not None]
        return sorted(out)


def test_freelist_without_frees_is_a_running_prefix_sum():
    """with no frees the allocator hands out [sum of earlier sizes, +size) — the visibility_offset
    sequence of the scene (scene.rs:422-431)."""
    rng = np.random.default_rng(3)
    sizes = rng.integers(1, 40, 500)
    a = S.FreeListAllocator(int(sizes.sum()) + 7)
    at = 0
    for s in sizes:
        idx, (b, e) = a.allocate(int(s))
        assert (b, e) == (at, at + int(s))
        at = e
    assert a.allocate(8) is None and a.allocate(7) is not None  # exact fit of the tail, then exhausted
    assert a.allocate(1) is None


def test_freelist_best_fit_and_first_of_equals():
    a = S.FreeListAllocator(100)
    blocks = [a.allocate(s) for s in (10, 5, 10, 5, 10, 60)]   # fills [0,100) exactly
    assert [b[1] for b in blocks] == [(0, 10), (10, 15), (15, 25), (25, 30), (30, 40), (40, 100)]
    a.deallocate(blocks[1][0])   # hole [10,15)
    a.deallocate(blocks[3][0])   # hole [25,30)
    a.deallocate(blocks[5][0])   # hole [40,100)
    assert a.allocate(5)[1] == (10, 15)     # two holes of 5: the first in arena slot order
    assert a.allocate(3)[1] == (25, 28)     # best fit: the 5-hole, not the 60-hole; split at the front
    assert a.allocate(2)[1] == (28, 30)     # exact fit of the remainder
    assert a.allocate(61) is None
    assert a.allocate(60)[1] == (40, 100)


def test_freelist_coalesces_with_live_neighbours():
    a = S.FreeListAllocator(64)
    x = a.allocate(16)
    y = a.allocate(16)
    z = a.allocate(32)              # exact fit of the tail: the list is x, y, z with correct links z<->tail gone
    a.deallocate(z[0])
    a.deallocate(y[0])              # y's next link -> z (free): merged into [16,64)
    assert a.blocks() == [(0, 16, False), (16, 64, True)]
    assert a.allocate(48)[1] == (16, 64)
    assert x[1] == (0, 16)


def test_freelist_forward_link_is_kept_correct():
    """freelist_alloc.rs:52-71 links a split-off block in front of the free block but never updates the PREVIOUS
    block's next_index; freeing that previous block would then merge it with the free tail across the live block in
    between.  The scene never deallocates visibility words (scene.rs:422-431), so the reference cannot reach that
    state; this restatement keeps the link correct (documented deviation), so frees behave."""
    a = S.FreeListAllocator(100)
    x = a.allocate(10)
    a.allocate(10)
    a.deallocate(x[0])
    assert a.blocks() == [(0, 10, True), (10, 20, False), (20, 100, True)]
    assert a.allocate(11)[1] == (20, 31) and a.allocate(10)[1] == (0, 10)


def _check_invariants(blocks, size):
    at = 0
    prev_free = False
    for b, e, free in blocks:
        assert b == at and e > b, blocks          # contiguous cover of [0, size), no overlap, no empty block
        assert not (free and prev_free), blocks   # free neighbours are always coalesced
        at, prev_free = e, free
    assert at == size


@pytest.mark.parametrize("seed", range(12))
def test_freelist_matches_python_restatement_on_random_traces(seed):
    rng = np.random.default_rng(100 + seed)
    size = int(rng.integers(50, 400))
    a, m = S.FreeListAllocator(size), PyFreeList(size)
    live = []
    for step in range(300):
        if live and rng.random() < 0.45:
            k = int(rng.integers(len(live)))
            ia, im = live.pop(k)
            a.deallocate(ia)
            m.deallocate(im)
        else:
            s = int(rng.integers(1, 24))
            ra, rm = a.allocate(s), m.allocate(s)
            assert (ra is None) == (rm is None), (seed, step)
            if ra is not None:
                assert ra[0].key() == rm[0] and ra[1] == rm[1], (seed, step)
                live.append((ra[0], rm[0]))
        assert a.blocks() == m.blocks(), (seed, step)
        _check_invariants(a.blocks(), size)


# ----------------------------------------------------------------------------- SceneData::update_scene
def _mesh_infos(counts):
    mi = np.zeros(len(counts), dtype=L.MESH_INFO)
    mi["lod_count"] = 1
    off = 0
    for i, c in enumerate(counts):
        mi["mesh_lods"][i, 0] = (off, c)
        off += c
    return mi


def test_update_scene_fills_the_upload_caches():
    counts = [70, 1, 32, 33, 256]
    mi = _mesh_infos(counts)
    sc = S.SceneData()
    sc.add_entity(position=(1, 2, 3), mesh=0, name="a")
    sc.add_entity(position=(0, 0, 0), light=dict(kind=S.POINT, intensity=4.0, param=0.1, color=(1, 0.5, 0.25)))
    sc.add_entity(position=(4, 5, 6), scale=(2, 2, 2), mesh=3)
    sc.add_entity(mesh=4, light=dict(kind=S.POINT, intensity=1.0))          # mesh and light on one entity
    sc.add_entity()                                                          # neither
    sc.add_entity(mesh=1)
    sc.update_scene(mi, luminance_cutoff=0.25)
    d = sc.entity_draw_cache()
    # instance_index counts mesh entities only (:421); words = ceil(lod-0 meshlets / 32) in entity order (:425-427)
    assert d["entity_index"].tolist() == [0, 1, 2, 3]
    assert d["mesh_index"].tolist() == [0, 3, 4, 1]
    assert d["visibility_offset"].tolist() == [0, 3, 5, 13]
    e = sc.entity_data_cache()
    assert len(e) == 4
    m0 = e["model_matrix"][0].reshape(4, 4)          # column-major: rows of this array are columns
    assert m0[3].tolist() == [1, 2, 3, 1] and np.array_equal(m0[:3, :3], np.eye(3, dtype=np.float32))
    m1 = e["model_matrix"][1].reshape(4, 4)
    assert m1[3].tolist() == [4, 5, 6, 1] and np.array_equal(m1[:3, :3], 2 * np.eye(3, dtype=np.float32))
    n1 = e["normal_matrix"][1].reshape(4, 4)         # inverse-transpose of the upper 3x3, identity elsewhere
    assert np.allclose(n1[:3, :3], 0.5 * np.eye(3)) and n1[3].tolist() == [0, 0, 0, 1]
    li = sc.light_data_cache()
    assert len(li) == 2
    assert li["light_type"].tolist() == [L.LIGHT_TYPE_POINT] * 2 and li["shadow_data_index"].tolist() == [0xFFFFFFFF] * 2
    assert li["position"][0].tolist() == [0, 0, 0] and li["color"][0].tolist() == [1, 0.5, 0.25]
    assert li["inner_radius"][0] == np.float32(0.1)
    assert li["outer_radius"][0] == np.sqrt(np.float32(4.0) / np.float32(0.25))     # scene.rs:273-275
    assert li["outer_radius"][1] == np.float32(2.0)
    # the buffer bytes: u32 count, then 12-B draws (scene.rs:470-481)
    raw = sc.entity_draw_buffer_bytes()
    assert raw[:4].view("<u4")[0] == 4 and raw.nbytes == 4 + 12 * 4


def test_update_scene_keeps_allocated_words_across_frames():
    mi = _mesh_infos([64, 40, 10])
    sc = S.SceneData()
    a = sc.add_entity(mesh=0)
    sc.add_entity(mesh=1)
    sc.update_scene(mi)
    first = sc.entity_draw_cache()["visibility_offset"].tolist()
    assert first == [0, 2]
    sc.add_entity(mesh=2)                      # a later entity gets the next words; earlier ranges are kept (:422-424)
    sc.set_transform(a, (9, 9, 9))
    sc.update_scene(mi, frame_index=1)
    assert sc.entity_draw_cache()["visibility_offset"].tolist() == [0, 2, 4]
    assert sc.entity_data_cache()["model_matrix"][0][12:15].tolist() == [9, 9, 9]


def test_update_scene_directional_shadow_index_and_direction():
    sc = S.SceneData()
    h = np.float32(np.sqrt(0.5))
    # 90 degrees about +x: forward (0,0,-1) -> (0,1,0); direction = -forward
    sc.add_entity(orientation=(h, 0, 0, h), light=dict(kind=S.DIRECTIONAL, param=0.6, cast_shadows=True), name="sun")
    sc.add_entity(light=dict(kind=S.DIRECTIONAL, cast_shadows=False))
    sc.add_entity(light=dict(kind=S.DIRECTIONAL, cast_shadows=True))
    sc.update_scene(_mesh_infos([1]), frame_index=1)
    li = sc.light_data_cache()
    assert np.allclose(li["direction"][0], [0, -1, 0], atol=1e-6)
    assert np.allclose(li["direction"][1], [0, 0, 1], atol=1e-6)
    # shadow_index + MAX_SHADOW_COMMANDS * frame_index (:463-464); lights without a shadow keep u32::MAX
    assert li["shadow_data_index"].tolist() == [256, 0xFFFFFFFF, 257]
    assert sc.shadow_command_count() == 2
    assert li["inner_radius"][0] == np.float32(0.6) and li["outer_radius"][0] == 0


def test_update_scene_panics_like_the_reference():
    from orbit_amd.passes import Panic

    sc = S.SceneData(visibility_chunk_count=4)
    sc.add_entity(mesh=0)
    sc.add_entity(mesh=0)
    with pytest.raises(Panic, match="exhausted"):       # allocate(..).unwrap() on None, scene.rs:427
        sc.update_scene(_mesh_infos([96]))
    sc2 = S.SceneData()
    sc2.add_entity(mesh=5)
    with pytest.raises(Panic, match="out of range"):    # mesh_infos[mesh] index panic, scene.rs:425
        sc2.update_scene(_mesh_infos([1]))
