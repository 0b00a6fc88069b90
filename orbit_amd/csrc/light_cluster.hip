// light_cluster.hip — clustered light binning for gfx950: mark active clusters,
// ordered compaction, light assignment.
//
// Restates shaders/light_cluster/{mark_active,active_cluster_compaction,
// light_culling}.comp + cluster_common.glsl (reference repo) under the host
// protocol of src/passes/cluster.rs:368-591.  MI355X-first structure:
//   mark    : one WAVE owns one screen tile, so the per-tile slice mask and the
//             per-cluster depth bounds are wave reductions written once — the
//             reference issues 3 global atomics per pixel (mark_active.comp:33-55)
//             into zero-filled buffers; here every word is written exactly once,
//             which also subsumes the zero-fill (cluster.rs:439-456).
//   compact : prefix-sum append in ascending linear cluster index (canonical
//             order), header written last (active_cluster_compaction.comp:33-43).
//   assign  : lights are transformed to view space ONCE (the reference redoes
//             world_to_view x position per cluster per light,
//             light_culling.comp:111) and cluster AABBs are computed once;
//             then three conservative levels — all lights against the union
//             box of 256 consecutive active clusters, those candidates
//             against the union box of a block's 8 clusters, the survivors
//             (lane = candidate) against the single boxes: hit ballots become
//             popcounts and kept hits in the count launch; the write launch
//             allocates ranges in compacted order (chunk sums + the counts in
//             front within the chunk) and copies the kept lists, testing
//             again only the few clusters that found more than were kept.
//   Four launches for the assignment, six for orbit_compute_clusters (whose
//   mark launch also takes the compaction's counts): DESIGN.md 4.4.
#include "handoff.h"

namespace orbit {

namespace {

// ---------------------------------------------------------------- mark_active
// mark_active.comp:27-57.  Lane L (< cz) accumulates the bounds of slice L.  A wave takes kMarkTiles consecutive tiles
// and issues the first 64 depth loads of all of them before it evaluates any: the kernel is a short chain of
// load -> log2 -> ballots -> stores per tile, so memory-level parallelism per wave is what shortens it.
constexpr uint32_t kMarkTiles = 4;
constexpr uint32_t kMarkWaves = 16; // 64 tiles per block: a quarter of the counting atomics of four-wave blocks

__global__ __launch_bounds__(kMarkWaves * 64) void cluster_mark_kernel(const ClusterMarkParams p) {
    __shared__ uint32_t s_zc[kMarkWaves][32]; // counting: per wave and z slice, how many of the wave's tiles are active in it
    // A tile's 32 depth bounds belong to 32 planes of the bounds array (slice-major): stored by the tile's wave they are
    // 32 eight-byte requests to 32 lines per tile, and the CU's one write port made 1.6 us of the launch out of them
    // (profiles/r06_notes.md).  They go through LDS instead: a plane's 64 entries of the block's 64 consecutive tiles are
    // 512 contiguous bytes, one wave's store.  (Row stride 65 entries: the waves' column writes touch every bank once.)
    constexpr uint32_t kBlockTiles = kMarkWaves * kMarkTiles;
    __shared__ uint2 s_bounds[32][kBlockTiles + 1];
    __shared__ uint32_t s_masks[kBlockTiles];
    const OrbitMarkActivePush &pc = p.pc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t cx = pc.cluster_count[0], cy = pc.cluster_count[1], cz = pc.cluster_count[2];
    // (wave-uniform: the tile walk below stays in scalar registers)
    const uint32_t tile0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * kMarkWaves + wave) * kMarkTiles));
    uint32_t zc = 0; // lane = z slice
    if (tile0 < cx * cy) { // wave-uniform
    const uint32_t ts = pc.tile_size_px, sc = pc.depth_buffer_sample_count;
    const uint32_t W = pc.screen_size[0], H = pc.screen_size[1];
    const uint32_t items = ts * ts * sc;
    // sample `it` of tile `tile`: its address, or none (mark_active.comp:40)
    auto sample = [&](uint32_t tile, uint32_t it, size_t &index) -> bool {
        // (divisors the compiler cannot see through: their reciprocals are not hoisted into every wave's prologue)
        uint32_t dcx = cx, dsc = sc, dts = ts;
        asm volatile("" : "+s"(dcx), "+s"(dsc), "+s"(dts));
        const uint32_t tx = tile % dcx, ty = tile / dcx;
        const uint32_t pix = it / dsc, smp = it % dsc;
        const uint32_t px = tx * ts + pix % dts, py = ty * ts + pix / dts;
        index = ((size_t)py * W + px) * sc + smp;
        return tile < cx * cy && it < items && px < W && py < H;
    };
    // The first 64 samples of every tile — all of them at the reference's 8-px tiles and one sample — by the same
    // arithmetic with the divisions taken out: a lane's place inside a tile is the same in every tile (once per wave,
    // shifts where tile size and sample count allow), a tile's corner is wave-uniform and walks along the row.
    // (A u32 division is ~25 vector instructions, a third of them quarter-rate multiplies; `sample` above cost this
    // launch, which is bound by its vector instructions, three of them per tile and lane.)
    uint32_t lpix = (uint32_t)lane, lsmp = 0u;
    if (sc != 1u) lpix = (uint32_t)lane / sc, lsmp = (uint32_t)lane % sc;
    uint32_t lpx, lpy;
    if ((ts & (ts - 1u)) == 0u && ts != 0u) lpx = lpix & (ts - 1u), lpy = lpix >> __builtin_ctz(ts);
    else lpx = lpix % ts, lpy = lpix / ts;
    const size_t lane_index = ((size_t)lpy * W + lpx) * sc + lsmp;
    const bool lane_in = (uint32_t)lane < items;
    uint32_t tx = tile0 % cx, ty = tile0 / cx; // (scalar)
    float first[kMarkTiles];
    bool first_valid[kMarkTiles];
#pragma unroll
    for (uint32_t t = 0; t < kMarkTiles; t++) {
        // ((py W + px) sc + smp with py = ty ts + lpy, px = tx ts + lpx: the tile's corner + the lane's place)
        const size_t index = ((size_t)(ty * ts) * W + tx * ts) * sc + lane_index;
        first_valid[t] = tile0 + t < cx * cy && lane_in && tx * ts + lpx < W && ty * ts + lpy < H;
        first[t] = first_valid[t] ? p.depth[index] : 0.0f;
        if (++tx == cx) tx = 0u, ty++;
    }
#pragma unroll
    for (uint32_t t = 0; t < kMarkTiles; t++) {
        const uint32_t tile = tile0 + t;
        if (tile >= cx * cy) break; // wave-uniform
        uint32_t acc_min = 0, acc_max = 0, tile_mask = 0;
        for (uint32_t base = 0; base < items; base += 64) {
            bool valid = first_valid[t];
            float d = first[t];
            if (base != 0) { // (larger tiles or several samples: the general form, its divisions set up in here only)
                size_t index;
                valid = sample(tile, base + (uint32_t)lane, index);
                d = valid ? p.depth[index] : 0.0f;
            }
            uint32_t slice = 0xFFFFFFFFu, bmin = 0, bmax = 0, bit = 0;
            if (valid) {
                const float linear_z = pc.z_near / d;                       // :28
                // cluster_common.glsl:18-20 as compiled (mark_active.comp.spv): one fused operation — through the
                // hardware log2 where that provably gives the canonical slice (orbit_device.h depth_slice)
                slice = depth_slice(linear_z, pc.z_scale, pc.z_bias);
                bit = shl1(slice);                                          // :30
                const float inv = 1.0f - d;                                 // :33
                // the sign / payload of a NaN produced by arithmetic is implementation-defined: canonical quiet NaN
                bmin = inv != inv ? 0x7fc00000u : __float_as_uint(inv);
                bmax = __float_as_uint(d);                                  // :34
            }
            const uint32_t chunk_mask = wave_reduce_or(bit);
            tile_mask |= chunk_mask;
            uint32_t todo = chunk_mask;
            if (cz < 32u) todo &= (1u << cz) - 1u; // :31 slice < cluster_count.z
            while (todo) {
                const uint32_t s = (uint32_t)__builtin_ctz(todo);
                todo &= todo - 1u;
                const bool mine = valid && slice == s;
                const uint32_t m1 = wave_reduce_max(mine ? bmin : 0u);
                const uint32_t m2 = wave_reduce_max(mine ? bmax : 0u);
                if ((uint32_t)lane == s) {
                    acc_min = max(acc_min, m1);
                    acc_max = max(acc_max, m2);
                }
            }
        }
        if ((uint32_t)lane < cz) s_bounds[lane][wave * kMarkTiles + t] = make_uint2(acc_min, acc_max);
        if (lane == 0) s_masks[wave * kMarkTiles + t] = tile_mask;
        zc += (uint32_t)lane < cz ? (tile_mask >> (lane & 31)) & 1u : 0u; // cluster_common.glsl:31-34
    }
    }
    if (p.chunk_counts != nullptr && lane < 32) s_zc[wave][lane] = zc;
    __syncthreads();
    {   // the block's outputs, every word of them once (tiles past the end were never filled in and are not stored)
        static_assert(sizeof(OrbitClusterDepthBounds) == sizeof(uint2), "bounds entry");
        const uint32_t tiles = cx * cy, block_tile0 = blockIdx.x * kBlockTiles;
        const uint32_t tb = (uint32_t)lane, tile = block_tile0 + tb; // kBlockTiles == 64: a wave per plane and round
        static_assert(kBlockTiles == 64, "one wave stores one plane's entries of the block");
        if (tile < tiles) {
#pragma unroll
            for (uint32_t slice = (uint32_t)wave; slice < 32u; slice += kMarkWaves) {
                if (slice >= cz) break; // wave-uniform
                const uint2 b = s_bounds[slice][tb];
                OrbitClusterDepthBounds o;
                o.min_depth = b.x;
                o.max_depth = b.y;
                p.bounds[tile + (size_t)slice * tiles] = o;
            }
            if (wave == 0) p.masks[tile] = s_masks[tb];
        }
    }
    if (p.chunk_counts == nullptr) return; // (uniform)
    // the block's 64 consecutive tiles lie in one chunk of 1024: one atomic per z slice the block is active in — a
    // handful, the tiles of a block see neighbouring depths — on the word of (slice, chunk).  (Atomics on one 128-B line
    // queue in its L2 channel at ~4 ns each: four-wave blocks, 16 tiles, cost the launch 2 us at config 4.)
    static_assert(1024 % (kMarkWaves * kMarkTiles) == 0, "a block's tiles share a compaction chunk");
    if (threadIdx.x < 32) {
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t w = 0; w < kMarkWaves; w++) sum += s_zc[w][threadIdx.x];
        const uint32_t nchunks = (cx * cy + 1023u) / 1024u, chunk = blockIdx.x * (kMarkWaves * kMarkTiles) / 1024u;
        if (sum != 0u)
            (void)__hip_atomic_fetch_add(p.chunk_counts + threadIdx.x * nchunks + chunk, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// -------------------------------------------------- active_cluster_compaction
// One 1024-thread block per (chunk of 1024 tiles, z slice).  The emit launch sums the counts of the preceding
// (slice, chunk) slots itself — at most a few thousand words from L2 — instead of a scan launch in between; block
// (0, 0) sums all of them for the header.
template <bool EMIT>
__global__ __launch_bounds__(1024) void cluster_compact_kernel(const ClusterCompactParams p) {
    __shared__ uint32_t smem[1024 / 64 + 1];
    const uint32_t tiles = p.cc[0] * p.cc[1];
    const uint32_t z = blockIdx.y, chunk = blockIdx.x;
    // (most (slice, chunk) pairs hold no active cluster: their emit blocks leave on the one word that says so)
    if (EMIT && (blockIdx.x | blockIdx.y) != 0u && p.chunk_counts[z * gridDim.x + chunk] == 0u) return;
    const uint32_t i = chunk * 1024 + threadIdx.x;
    const uint32_t active = (i < tiles && (p.masks[i] & shl1(z))) ? 1u : 0u; // cluster_common.glsl:31-34
    uint32_t block_total;
    const uint32_t ex = block_exclusive_scan<16>(active, smem, &block_total);
    const uint32_t slot_index = z * gridDim.x + chunk;
    if (!EMIT) {
        if (threadIdx.x == 0) p.chunk_counts[slot_index] = block_total;
    } else {
        uint32_t *idx = reinterpret_cast<uint32_t *>(p.unique + ORBIT_COMPACT_HEADER);
        const bool first = blockIdx.x == 0 && blockIdx.y == 0;
        const uint32_t slots = gridDim.x * gridDim.y, upto = first ? slots : slot_index;
        uint32_t part = 0;
        for (uint32_t k = threadIdx.x; k < upto; k += 1024) part += p.chunk_counts[k];
        part = wave_reduce_add(part);
        __syncthreads(); // smem of the scan above is free again
        if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = part;
        __syncthreads();
        uint32_t preceding = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) preceding += smem[w];
        const uint32_t total = preceding; // meaningful in block (0, 0) only
        const uint32_t pos = (first ? 0u : preceding) + ex;
        if (active && pos < p.index_capacity) idx[pos] = i + z * tiles; // cluster_common.glsl:1-4
        if (first && threadIdx.x == 0) {
            const uint32_t n = min(total, p.index_capacity);
            uint32_t *hdr = reinterpret_cast<uint32_t *>(p.unique);
            hdr[0] = (n + 255u) / 256u; // active_cluster_compaction.comp:38-41
            hdr[1] = 1u;
            hdr[2] = 1u;
            hdr[3] = n;
            if (total > p.index_capacity) latch_status(p.status, ORBIT_E_CAPACITY);
        }
    }
}

// -------------------------------------------------------------- light_culling
__device__ __forceinline__ void light_prepare_body(const ClusterAssignParams &p, uint32_t block) {
    const uint32_t i = block * 256 + threadIdx.x;
    const uint32_t n = p.info.global_light_count;
    bool point = false;
    if (i < n) {
        const OrbitLightData &l = p.lights[i];
        point = l.light_type == ORBIT_LIGHT_TYPE_POINT;
        const float *m = p.info.world_to_view_matrix;
        const float x = l.position[0], y = l.position[1], z = l.position[2];
        float4 v;
        v.x = ((m[0] * x + m[4] * y) + m[8] * z) + m[12] * 1.0f; // light_culling.comp:111
        v.y = ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * 1.0f;
        v.z = ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * 1.0f;
        v.w = l.outer_radius * l.outer_radius;                     // sphere.w * sphere.w, :103
        p.view_lights[i] = v;
    }
    const uint64_t b = __ballot(point);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_i = i - lane;
    if (lane == 0 && wave_i < n) p.light_flags[wave_i / 32] = (uint32_t)b;
    if (lane == 32 && wave_i + 32 < n) p.light_flags[wave_i / 32 + 1] = (uint32_t)(b >> 32);
}

struct Aabb3 {
    float mn[3], mx[3];
};

// screen_to_view, light_culling.comp:34-48
__device__ __forceinline__ void screen_to_view(const OrbitClusterCullInfo &in, float sx, float sy, float out[3]) {
    const float tx = sx / (float)in.screen_size[0], ty = sy / (float)in.screen_size[1];
    const float c0 = tx * 2.0f - 1.0f, c1 = (1.0f - ty) * 2.0f - 1.0f, c2 = 1.0f, c3 = 1.0f;
    const float *m = in.screen_to_view_matrix;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = ((m[0 + r] * c0 + m[4 + r] * c1) + m[8 + r] * c2) + m[12 + r] * c3;
    out[0] = v[0] / v[3];
    out[1] = v[1] / v[3];
    out[2] = v[2] / v[3];
}

// line_intersection_to_z_plane with a = eye = 0, light_culling.comp:50-60
__device__ __forceinline__ void line_z(const float b[3], float zd, float out[3]) {
    const float ab0 = b[0] - 0.0f, ab1 = b[1] - 0.0f, ab2 = b[2] - 0.0f;
    const float dna = (0.0f * 0.0f + 0.0f * 0.0f) + -1.0f * 0.0f;
    const float dnab = (0.0f * ab0 + 0.0f * ab1) + -1.0f * ab2;
    const float t = (zd - dna) / dnab;
    out[0] = 0.0f + t * ab0;
    out[1] = 0.0f + t * ab1;
    out[2] = 0.0f + t * ab2;
}

// compute_cluster_volume, light_culling.comp:62-90
__device__ __forceinline__ Aabb3 cluster_aabb(const ClusterAssignParams &p, uint32_t cluster_index) {
    const OrbitClusterCullInfo &in = p.info;
    const uint32_t cx = in.cluster_count[0], cy = in.cluster_count[1];
    uint32_t idx = cluster_index;
    const uint32_t z = idx / (cx * cy);
    idx -= z * cx * cy;
    const uint32_t y = idx / cx;
    idx -= y * cx;
    const uint32_t x = idx;
    const float minx = (float)(x * in.tile_size_px), miny = (float)(y * in.tile_size_px);
    const float maxx = gmin(minx + (float)in.tile_size_px, (float)in.screen_size[0]);
    const float maxy = gmin(miny + (float)in.tile_size_px, (float)in.screen_size[1]);
    float minv[3], maxv[3];
    screen_to_view(in, minx, miny, minv);
    screen_to_view(in, maxx, maxy, maxv);
    const OrbitClusterDepthBounds db = p.bounds[cluster_index];
    const float min_depth = 1.0f - __uint_as_float(db.min_depth); // :72
    const float max_depth = __uint_as_float(db.max_depth);        // :73
    const float cnear = in.z_near / max_depth, cfar = in.z_near / min_depth;
    float q[4][3];
    line_z(minv, cnear, q[0]);
    line_z(minv, cfar, q[1]);
    line_z(maxv, cnear, q[2]);
    line_z(maxv, cfar, q[3]);
    Aabb3 a;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        a.mn[i] = gmin(gmin(q[0][i], q[1][i]), gmin(q[2][i], q[3][i]));
        a.mx[i] = gmax(gmax(q[0][i], q[1][i]), gmax(q[2][i], q[3][i]));
    }
    return a;
}

// aabb_sphere_test, light_culling.comp:92-104 (l.w already holds r*r)
__device__ __forceinline__ bool sphere_hits(const Aabb3 &a, const float4 l) {
    // as compiled (light_culling.comp.spv): sqr_dist = fma(d, d, sqr_dist) per term
    float sq = 0.0f;
    if (l.x < a.mn[0]) sq = __builtin_fmaf(a.mn[0] - l.x, a.mn[0] - l.x, sq);
    if (l.x > a.mx[0]) sq = __builtin_fmaf(l.x - a.mx[0], l.x - a.mx[0], sq);
    if (l.y < a.mn[1]) sq = __builtin_fmaf(a.mn[1] - l.y, a.mn[1] - l.y, sq);
    if (l.y > a.mx[1]) sq = __builtin_fmaf(l.y - a.mx[1], l.y - a.mx[1], sq);
    if (l.z < a.mn[2]) sq = __builtin_fmaf(a.mn[2] - l.z, a.mn[2] - l.z, sq);
    if (l.z > a.mx[2]) sq = __builtin_fmaf(l.z - a.mx[2], l.z - a.mx[2], sq);
    return sq <= l.w;
}

// Executed-test accounting (SURVEY §8d prices the assignment in sphere-box tests, not bytes): -DORBIT_TRIAGE builds
// only, one atomic per wave at the end of its work; the product library compiles none of it.
#ifdef ORBIT_TRIAGE
#define ORBIT_COUNT_TESTS(slot, n)                                                                  \
    do {                                                                                            \
        if (p.debug_tests && (threadIdx.x & 63) == 0 && (n) != 0) atomicAdd(p.debug_tests + (slot), (unsigned long long)(n)); \
    } while (0)
#define ORBIT_MAX_TESTS(slot, n)                                                                    \
    do {                                                                                            \
        if (p.debug_tests && threadIdx.x == 0) atomicMax(p.debug_tests + (slot), (unsigned long long)(n)); \
    } while (0)
#else
#define ORBIT_COUNT_TESTS(slot, n) ((void)0)
#define ORBIT_MAX_TESTS(slot, n) ((void)0)
#endif

constexpr uint32_t kLightTile = 1024;  // coarse candidates a block filters per step; survivors live in LDS (20 KB)
// 4 waves x 2 clusters = 8 clusters per block (round 3; rounds 1-2 ran 4 x 4: the block's size is the size of the
// launch's tail, DESIGN.md 4.4 — 16 -> 8 clusters per block took the count launch from 29.8 to 23.4 us)
constexpr int kClustersPerWave = 2;    // cluster AABBs held in scalar registers per wave
constexpr int kAssignWaves = 4;
constexpr int kClustersPerBlock = kClustersPerWave * kAssignWaves;
constexpr uint32_t kGroupClusters = 256; // coarse level: one candidate list per 256 consecutive active clusters
constexpr uint32_t kNotPoint = 1u << 31; // candidate index flag: not a point light -> in every cluster (:116-118)

// The assignment is a three-level cull of the (cluster, light) pairs, every level conservative for the next:
//   coarse : a block owns 256 consecutive entries of the compacted list.  It computes their AABBs once
//            (compute_cluster_volume, :62-90 — cached for the later launches), takes the UNION of the boxes and
//            filters ALL lights against it: wave w takes the w-th quarter of the lights and appends the survivors,
//            in light order, to its own segment of the group's candidate list (no barrier in the filter).
//   filter : a block of the count / write launches owns 16 consecutive entries (neighbours along x); wave w
//            filters coarse segment w against the union of those 16 boxes into an LDS segment.
//   test   : lane = candidate.  A wave holds the AABBs of its 4 clusters in SGPRs, tests 64 candidates per step
//            and turns the hit ballots into counts (WRITE = false, light_culling.comp:128-135) or into ascending
//            index writes at the scanned offset (WRITE = true, :138-147).
// A union contains every member box, and aabb_sphere_test is monotone in the box (each clamp distance, each square
// and each partial sum can only shrink when the box grows, also in floating point), so no light that hits a member
// box is ever filtered out; a union over boxes that contain a NaN does not filter at all.  Segment order == light
// order, so the per-cluster lists come out ascending.  The reference runs one thread per cluster over all lights,
// re-reading each 64-B light twice and redoing world_to_view x position per (cluster, light).
struct BoxUnion {
    Aabb3 box;
    bool any_nan;
};

__device__ __forceinline__ void union_add(BoxUnion &u, const float lo[3], const float hi[3]) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        u.any_nan = u.any_nan || lo[i] != lo[i] || hi[i] != hi[i];
        u.box.mn[i] = lo[i] < u.box.mn[i] ? lo[i] : u.box.mn[i];
        u.box.mx[i] = hi[i] > u.box.mx[i] ? hi[i] : u.box.mx[i];
    }
}

constexpr uint32_t kCoarseSplit = 4;                    // blocks that share the lights of one group
constexpr uint32_t kCoarseSegs = kCoarseSplit * 4;      // = waves per group = segments of its candidate list

// coarse, part 1: AABBs of 256 consecutive active clusters (cached for every later launch) and their union.
// (The launch's first `prepare_blocks` blocks transform the lights instead: two independent pieces of work, one launch.)
__global__ __launch_bounds__(256) void cluster_aabb_kernel(const ClusterAssignParams p, uint32_t prepare_blocks) {
    __shared__ float s_box[4][8];
    if (blockIdx.x < prepare_blocks) { // block-uniform
        light_prepare_body(p, blockIdx.x);
        return;
    }
    const uint32_t *hdr = reinterpret_cast<const uint32_t *>(p.unique);
    const uint32_t n = min(hdr[3], p.max_clusters);
    if (blockIdx.x == prepare_blocks) { // the launch's housekeeping block
        // what the count launch accumulates into: the number of its heavy blocks (and, per group below, the chunks' sums);
        // the chunk counts a counting mark launch left (the compaction has read them)
        if (threadIdx.x == 0) p.total[0] = 0u;
        for (uint32_t k = threadIdx.x; k < p.zero_count; k += 256u) p.zero_words[k] = 0u;
        // The order the count launch takes the groups in: longest candidate lists first.  That launch lasts as long as
        // its last block, a block's time grows with its group's list, and in compacted order the long ones come
        // wherever the lights are (config 4: the one group with 1 099 candidates started 9 us into the launch and ended
        // it at 22).  A scheduling hint only — any permutation gives the same lists — so it is taken from what is at
        // hand now: the segment counts the PREVIOUS assignment's coarse launch left (this one's rewrites them two
        // launches on).  Three classes by the group's total, stable within a class; one packed scan places all three.
        __shared__ uint32_t s_scan[4 + 1];
        const uint32_t groups = (n + kGroupClusters - 1u) / kGroupClusters;
        auto cls = [&](uint32_t g) -> uint32_t { // 0: > 512 candidates, 1: > 128, 2: the rest
            const uint4 *c = reinterpret_cast<const uint4 *>(p.coarse_counts + (size_t)g * kCoarseSegs);
            uint32_t t = 0;
#pragma unroll
            for (uint32_t j = 0; j < kCoarseSegs / 4; j++) t += (c[j].x + c[j].y) + (c[j].z + c[j].w);
            return t > 512u ? 0u : t > 128u ? 1u : 2u;
        };
        uint32_t base[3] = {0u, 0u, 0u}; // where each class begins, then how far it has been filled
        const uint32_t c_first = threadIdx.x < groups ? cls(threadIdx.x) : 3u; // (kept for the first round below)
        {
            uint32_t c0 = c_first == 0u ? 1u : 0u, c1 = c_first == 1u ? 1u : 0u;
            for (uint32_t g = 256u + threadIdx.x; g < groups; g += 256u) {
                const uint32_t c = cls(g);
                c0 += c == 0u ? 1u : 0u, c1 += c == 1u ? 1u : 0u;
            }
            uint32_t t0, t1;
            (void)block_exclusive_scan<4>(c0, s_scan, &t0);
            (void)block_exclusive_scan<4>(c1, s_scan, &t1);
            base[1] = t0, base[2] = t0 + t1;
        }
        for (uint32_t g0 = 0; g0 < groups; g0 += 256u) { // block-uniform trip count
            const uint32_t g = g0 + threadIdx.x;
            const uint32_t c = g0 == 0u ? c_first : (g < groups ? cls(g) : 3u);
            uint32_t tot; // 10 bits per class: at most 256 flags each
            const uint32_t ex = block_exclusive_scan<4>(c < 3u ? 1u << (10u * c) : 0u, s_scan, &tot);
            if (c < 3u) p.group_order[base[c] + ((ex >> (10u * c)) & 1023u)] = g;
#pragma unroll
            for (uint32_t k = 0; k < 3u; k++) base[k] += (tot >> (10u * k)) & 1023u;
        }
        return;
    }
    // grid-stride over the groups that exist (n is only known on the device; a grid sized by the capacity would
    // spend its time launching empty blocks)
    for (uint32_t grp = blockIdx.x - prepare_blocks - 1u; grp * kGroupClusters < n; grp += gridDim.x - prepare_blocks - 1u) {
    const uint32_t first = grp * kGroupClusters;
    static_assert(kScanChunk % kGroupClusters == 0, "a chunk of the offsets' sums is whole groups");
    if (threadIdx.x == 0 && first % kScanChunk == 0u) p.block_sums[(size_t)(first / kScanChunk) * kChunkSumStride] = 0u;
    const uint32_t *unique_idx = reinterpret_cast<const uint32_t *>(p.unique + ORBIT_COMPACT_HEADER);
    const uint32_t u = first + threadIdx.x;
    const float inf = __uint_as_float(0x7f800000u);
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf}; // absent cluster: the empty box, neutral in the union
    if (u < n) {
        const Aabb3 box = cluster_aabb(p, unique_idx[u]);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            lo[i] = box.mn[i], hi[i] = box.mx[i];
            p.aabb[6 * (size_t)u + i] = box.mn[i];
            p.aabb[6 * (size_t)u + 3 + i] = box.mx[i];
        }
    }
    // union over the block: butterfly over the lanes, then over the four waves.  With a NaN anywhere the filter is
    // bypassed (flag), so what the selects make of a NaN operand does not matter.
    bool nan = false;
#pragma unroll
    for (int i = 0; i < 3; i++) nan = nan || lo[i] != lo[i] || hi[i] != hi[i];
    const bool wave_nan = __ballot(nan) != 0ull;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float ol = __shfl_xor(lo[i], d, 64), oh = __shfl_xor(hi[i], d, 64);
            lo[i] = ol < lo[i] ? ol : lo[i];
            hi[i] = oh > hi[i] ? oh : hi[i];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 3; i++) s_box[wave][i] = lo[i], s_box[wave][3 + i] = hi[i];
        s_box[wave][6] = wave_nan ? 1.0f : 0.0f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        BoxUnion un;
        un.any_nan = false;
#pragma unroll
        for (int i = 0; i < 3; i++) un.box.mn[i] = inf, un.box.mx[i] = -inf;
        for (int w = 0; w < 4; w++) {
            union_add(un, &s_box[w][0], &s_box[w][3]);
            un.any_nan = un.any_nan || s_box[w][6] != 0.0f;
        }
        float *g = p.group_box + 8 * (size_t)grp;
#pragma unroll
        for (int i = 0; i < 3; i++) g[i] = un.box.mn[i], g[3 + i] = un.box.mx[i];
        g[6] = un.any_nan ? 1.0f : 0.0f;
    }
    __syncthreads(); // s_box is rewritten by the next group
    }
}

// coarse, part 2: kCoarseSplit blocks per group; wave s of a group (s < kCoarseSegs) filters lights
// [s * seg, (s + 1) * seg) against the group's union box into segment s of the group's list, in light order.
__global__ __launch_bounds__(256) void cluster_coarse_kernel(const ClusterAssignParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t *hdr = reinterpret_cast<const uint32_t *>(p.unique);
    const uint32_t n = min(hdr[3], p.max_clusters);
    for (uint32_t vb = blockIdx.x; (vb / kCoarseSplit) * kGroupClusters < n; vb += gridDim.x) { // as in cluster_aabb_kernel
    const uint32_t group = vb / kCoarseSplit;
    const float *g = p.group_box + 8 * (size_t)group;
    BoxUnion un;
#pragma unroll
    for (int i = 0; i < 3; i++) un.box.mn[i] = g[i], un.box.mx[i] = g[3 + i];
    un.any_nan = g[6] != 0.0f;
    const uint32_t nl = p.info.global_light_count, seg = p.coarse_seg;
    const uint32_t sidx = (vb % kCoarseSplit) * 4 + wave;
    uint32_t *out = p.coarse + ((size_t)group * kCoarseSegs + sidx) * seg;
    float4 *out_l = p.coarse_lights + ((size_t)group * kCoarseSegs + sidx) * seg; // the candidate itself beside its index
    uint32_t cnt = 0;
    [[maybe_unused]] uint32_t tested = 0;
    const uint32_t begin = sidx * seg, end = min(begin + seg, nl);
    for (uint32_t b0 = begin; b0 < end; b0 += 256) { // 4 loads per lane issued together
        float4 l[4];
        uint32_t pt = 0;
#pragma unroll
        for (uint32_t r = 0; r < 4; r++) {
            const uint32_t li = b0 + r * 64u + (uint32_t)lane;
            const uint32_t gi = li < end ? li : 0u;
            l[r] = p.view_lights[gi];
            pt |= ((p.light_flags[gi >> 5] >> (gi & 31)) & 1u) << r;
        }
#pragma unroll
        for (uint32_t r = 0; r < 4; r++) {
            const uint32_t li = b0 + r * 64u + (uint32_t)lane;
            const bool point = (pt >> r) & 1u;
            const bool pass = li < end && (!point || un.any_nan || sphere_hits(un.box, l[r]));
            const uint64_t m = __ballot(pass);
            if (pass) {
                const uint32_t pos = cnt + lane_prefix(m);
                out[pos] = li | (point ? 0u : kNotPoint);
                out_l[pos] = l[r];
            }
            cnt += (uint32_t)__popcll(m);
#ifdef ORBIT_TRIAGE
            tested += (uint32_t)__popcll(__ballot(li < end && point && !un.any_nan));
#endif
        }
    }
    ORBIT_COUNT_TESTS(0, tested);
    ORBIT_COUNT_TESTS(1, cnt);
    if (lane == 0) p.coarse_counts[group * kCoarseSegs + sidx] = cnt;
    }
}

// Offsets without a scan launch: the count launch adds every wave's (capped) counts to the sum of its chunk of
// kScanChunk clusters and lists the blocks that hold a cluster with more than kHitCache lights ("heavy").  A cluster's
// offset is then the sum of the chunks before its own plus the counts of its chunk's clusters before it — at most a
// thousand words each, read by whoever needs the offset.  The write launch is two kinds of blocks: the first
// `place_blocks` take 256 clusters each (thread = cluster), compute the offsets and copy the lists of the clusters
// whose hits the count launch kept (all but a few); the others stride over the heavy blocks and test those clusters
// again (WRITE = true below).
__device__ __forceinline__ void cluster_place_body(const ClusterAssignParams &p, uint32_t place_blocks) {
    __shared__ uint32_t s_scan[4 + 1];
    __shared__ uint32_t s_part[4];
    const uint32_t *hdr = reinterpret_cast<const uint32_t *>(p.unique);
    const uint32_t *unique_idx = reinterpret_cast<const uint32_t *>(p.unique + ORBIT_COMPACT_HEADER);
    uint32_t *out = reinterpret_cast<uint32_t *>(p.light_index_buffer + ORBIT_LIGHT_INDEX_HEADER);
    const uint32_t n = min(hdr[3], p.max_clusters);
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    static_assert(kScanChunk % 256u == 0, "a piece of 256 clusters lies in one chunk");
    for (uint32_t piece = blockIdx.x; piece * 256u < max(n, 1u); piece += place_blocks) { // (n = 0: block 0 writes the header)
        const uint32_t u = piece * 256u + t;
        const bool in = u < n;
        const uint32_t cnt = in ? p.counts[u] : 0u;
        const uint32_t ci = in ? unique_idx[u] : 0u;
        uint4 h[kPlaceDirect / 4];
#pragma unroll
        for (uint32_t j = 0; j < kPlaceDirect / 4; j++) // the first 64 B whatever the count says: no load waits for another
            h[j] = reinterpret_cast<const uint4 *>(p.hit_cache + (size_t)min(u, p.max_clusters) * kHitCache)[j];
        const uint32_t chunk = piece * 256u / kScanChunk;
        uint32_t part = 0;
        for (uint32_t k = t; k < chunk; k += 256u) part += p.block_sums[(size_t)k * kChunkSumStride];
        uint32_t cv[kScanChunk / 256u - 1u]; // the counts of the chunk's clusters in front of the piece, loaded together
#pragma unroll
        for (uint32_t j = 0; j < kScanChunk / 256u - 1u; j++) {
            const uint32_t v = chunk * kScanChunk + j * 256u + t;
            cv[j] = v < piece * 256u ? p.counts[v] : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < kScanChunk / 256u - 1u; j++) part += cv[j];
        part = wave_reduce_add(part);
        if (lane == 0) s_part[wave] = part;
        uint32_t piece_total;
        const uint32_t ex = block_exclusive_scan<4>(cnt, s_scan, &piece_total); // (its barriers order s_part as well)
        const uint32_t before = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        const uint32_t off = before + ex;
        // the kept hits, kPlaceDirect at a time (thread = cluster).  The first round's loads were issued above; a further
        // round is taken only by waves that hold a cluster that long (a round trip of their own: a few waves per launch)
        const bool kept = in && cnt <= kHitCache;
#pragma unroll
        for (uint32_t r = 0; r < kHitCache / kPlaceDirect; r++) {
            if (r != 0u) {
                if (__ballot(kept && cnt > r * kPlaceDirect) == 0ull) break; // wave-uniform
#pragma unroll
                for (uint32_t j = 0; j < kPlaceDirect / 4; j++)
                    h[j] = reinterpret_cast<const uint4 *>(p.hit_cache + (size_t)min(u, p.max_clusters) * kHitCache + r * kPlaceDirect)[j];
            }
            const uint32_t w[kPlaceDirect] = {h[0].x, h[0].y, h[0].z, h[0].w, h[1].x, h[1].y, h[1].z, h[1].w,
                                              h[2].x, h[2].y, h[2].z, h[2].w, h[3].x, h[3].y, h[3].z, h[3].w};
#pragma unroll
            for (uint32_t k = 0; k < kPlaceDirect; k++)
                if (kept && r * kPlaceDirect + k < cnt && (uint64_t)off + r * kPlaceDirect + k < p.light_index_capacity)
                    out[off + r * kPlaceDirect + k] = w[k]; // :138-147
        }
        if (kept) { // imageStore(cluster_offset_image, cluster_id, (offset, count)), :149-150
            p.offset_image[2 * (size_t)ci + 0] = off;
            p.offset_image[2 * (size_t)ci + 1] = cnt;
        }
        if (t == 0 && piece * 256u + 256u >= n) { // the last piece: light_count = the sum of all counts (atomicAdd total, :136)
            const uint32_t total = before + piece_total;
            *reinterpret_cast<uint32_t *>(p.light_index_buffer) = total;
            if (total > p.light_index_capacity) latch_status(p.status, ORBIT_E_CAPACITY);
        }
        __syncthreads(); // s_part is rewritten by the next piece
    }
}

// ---- the count launch's item for ONE wave (see cluster_assign_kernel): eight clusters, lane c < 8 holds cluster c
constexpr uint32_t kWaveSpec = 8;              // candidates per coarse segment this path handles (fetched speculatively)
constexpr uint32_t kWaveClusters = kClustersPerBlock;
struct WaveItem {
    uint32_t ci;       // lanes 0..7: the cluster's index in the grid
    float lo[3], hi[3]; // ... its AABB
    uint32_t tag[2];   // round r: entry (lane & 7) of segment 8 r + (lane >> 3) — segments ascending = light order
    float4 l[2];
};

__device__ __forceinline__ void wave_item_loads(const ClusterAssignParams &p, WaveItem &w, uint32_t block_first,
                                                const uint32_t *unique_idx, const uint32_t *cbase, const float4 *clbase, int lane) {
    static_assert(kWaveClusters == 8 && kCoarseSegs == 16 && kWaveSpec == 8, "lane = (segment of the round, entry) = (lane >> 3, lane & 7)");
    const uint32_t ul = min(block_first + ((uint32_t)lane & (kWaveClusters - 1)), p.max_clusters - 1u);
    w.ci = unique_idx[ul];
#pragma unroll
    for (int i = 0; i < 3; i++) w.lo[i] = p.aabb[6 * (size_t)ul + i], w.hi[i] = p.aabb[6 * (size_t)ul + 3 + i];
#pragma unroll
    for (uint32_t r = 0; r < 2; r++) {
        const uint32_t slot = (r * 8u + ((uint32_t)lane >> 3)) * p.coarse_seg + ((uint32_t)lane & 7u);
        w.tag[r] = cbase[slot];
        w.l[r] = clbase[slot];
    }
}

__device__ __forceinline__ void wave_item(const ClusterAssignParams &p, const WaveItem &w, uint32_t block_first, uint32_t n,
                                          uint32_t seg_counts, float4 *s_cand, uint32_t *s_cand_idx, float *s_box, int lane) {
    const bool owns = lane < (int)kWaveClusters && block_first + (uint32_t)lane < n;
    const float inf = __uint_as_float(0x7f800000u);
    // the eight boxes: absent cluster = the empty box, neutral in the union; union by a butterfly over lanes 0..7
    float lo[3], hi[3];
    bool nan = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        lo[i] = owns ? w.lo[i] : inf, hi[i] = owns ? w.hi[i] : -inf;
        nan = nan || lo[i] != lo[i] || hi[i] != hi[i];
    }
    BoxUnion un;
    un.any_nan = (__ballot(nan) & 0xFFull) != 0ull; // (then the filter is bypassed: what the selects make of a NaN does not matter)
    // every cluster's box where the test loop reads it with uniform addresses (LDS broadcasts; 48 scalar registers
    // held across the loop spilled), and the union by a butterfly over lanes 0..7
    if (lane < (int)kWaveClusters) {
#pragma unroll
        for (int i = 0; i < 3; i++) s_box[lane * 8 + i] = lo[i], s_box[lane * 8 + 4 + i] = hi[i];
    }
#pragma unroll
    for (int d = (int)kWaveClusters / 2; d >= 1; d >>= 1) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float ol = __shfl_xor(lo[i], d, 64), oh = __shfl_xor(hi[i], d, 64);
            lo[i] = ol < lo[i] ? ol : lo[i];
            hi[i] = oh > hi[i] ? oh : hi[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        un.box.mn[i] = __uint_as_float(__builtin_amdgcn_readfirstlane((int)__float_as_uint(lo[i])));
        un.box.mx[i] = __uint_as_float(__builtin_amdgcn_readfirstlane((int)__float_as_uint(hi[i])));
    }
    // filter: two rounds of 8 segments x 8 entries, survivors appended to the wave's LDS list in lane order = light order
    uint32_t total = 0;
    [[maybe_unused]] uint32_t n_filter = 0, n_test = 0;
#pragma unroll
    for (uint32_t r = 0; r < 2; r++) {
        const uint32_t cnt = (uint32_t)__shfl((int)seg_counts, (int)(r * 8u + ((uint32_t)lane >> 3)), 64);
        const bool have = ((uint32_t)lane & 7u) < cnt;
        const bool point = (w.tag[r] & kNotPoint) == 0u;
        const bool pass = have && (!point || un.any_nan || sphere_hits(un.box, w.l[r]));
        const uint64_t m = __ballot(pass);
#ifdef ORBIT_TRIAGE
        n_filter += (uint32_t)__popcll(__ballot(have && point && !un.any_nan));
#endif
        if (pass) {
            const uint32_t pos = total + lane_prefix(m);
            s_cand[pos] = w.l[r];
            s_cand_idx[pos] = w.tag[r];
        }
        total += (uint32_t)__popcll(m);
    }
    wave_lds_fence();
    // test: lane = candidate, against the eight boxes in turn (a rolled loop: unrolled, the compiler hoists all 48 box
    // words out of the candidate loop and spills); lane c keeps cluster c's count; the first kHitCache hits of a
    // cluster are kept for the write launch
    uint32_t mine = 0;
    for (uint32_t b0 = 0; b0 < total; b0 += 64u) { // (at most two trips)
        const uint32_t k = b0 + (uint32_t)lane;
        const bool in_list = k < total;
        const float4 cl = s_cand[in_list ? k : 0u];
        const uint32_t ctag = s_cand_idx[in_list ? k : 0u];
        const bool point = (ctag & kNotPoint) == 0u;
#ifdef ORBIT_TRIAGE
        n_test += (uint32_t)__popcll(__ballot(in_list && point)) * kWaveClusters;
#endif
#pragma unroll 1
        for (uint32_t c = 0; c < kWaveClusters; c++) {
            if (block_first + c >= n) break; // wave-uniform
            Aabb3 a;
#pragma unroll
            for (int i = 0; i < 3; i++) a.mn[i] = s_box[c * 8u + i], a.mx[i] = s_box[c * 8u + 4u + i];
            const uint64_t m = __ballot(in_list && (!point || sphere_hits(a, cl))); // :108-119
            if (m == 0ull) continue; // wave-uniform
            const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)c);
            if (before < kHitCache && ((m >> lane) & 1ull)) {
                const uint32_t slot = before + lane_prefix(m);
                if (slot < kHitCache) p.hit_cache[(size_t)(block_first + c) * kHitCache + slot] = ctag & ~kNotPoint;
            }
            mine += (uint32_t)lane == c ? (uint32_t)__popcll(m) : 0u;
        }
    }
    ORBIT_COUNT_TESTS(2, n_filter);
    ORBIT_COUNT_TESTS(3, n_test);
    const uint32_t capped = owns ? min(mine, ORBIT_MAX_LIGHTS_PER_CLUSTER) : 0u; // :135
    if (owns) p.counts[block_first + (uint32_t)lane] = capped;
    const uint32_t block_total = wave_reduce_add(capped);
    const bool heavy = __ballot(capped > kHitCache) != 0ull;
    if (lane == 0) { // as the four-wave path does at its end
        if (block_total != 0u)
            (void)__hip_atomic_fetch_add(p.block_sums + (size_t)(block_first / kScanChunk) * kChunkSumStride, block_total,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (heavy)
            p.block_base[__hip_atomic_fetch_add(p.total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = block_first / kClustersPerBlock;
    }
}

#ifndef ORBIT_ASSIGN_WAVES // five waves per SIMD: five blocks per CU resident (96 registers; the count launch is its blocks' latencies)
#define ORBIT_ASSIGN_WAVES __attribute__((amdgpu_waves_per_eu(5)))
#endif
template <bool WRITE>
__global__ __launch_bounds__(kAssignWaves * 64) ORBIT_ASSIGN_WAVES
void cluster_assign_kernel(const ClusterAssignParams p, uint32_t place_blocks) {
    if (WRITE && blockIdx.x < place_blocks) { // block-uniform
        cluster_place_body(p, place_blocks);
        return;
    }
    __shared__ float4 s_cand[kLightTile];       // a segment per wave: view-space light ...
    __shared__ uint32_t s_cand_idx[kLightTile]; // ... and its index | kNotPoint
    __shared__ float s_wbox[kAssignWaves][8];   // per wave: union of its clusters' boxes (6) and its any-NaN flag
    __shared__ float s_ibox[kWaveClusters][8];  // one-wave items: the eight clusters' boxes.  Wave 0's own words: waves 1-3
                                                // leave a one-wave item without a barrier and may be writing s_wbox for the
                                                // NEXT (four-wave) item while wave 0 still tests this one
    __shared__ uint32_t s_wave_cnt[kAssignWaves];
    __shared__ uint32_t s_cpre[kCoarseSegs + 1]; // prefix of the group's segment counts
    __shared__ uint32_t s_ccnt[kCoarseSegs];     // the counts themselves
    __shared__ uint32_t s_fast;
    __shared__ uint32_t s_wtot[kClustersPerBlock]; // count launch: the clusters' capped counts
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef ORBIT_TRIAGE_STAMPS // (a build of its own, tools/cluster_tests.py: the counters' atomics would dominate the stamps)
    // block life stamps of the COUNT launch (wall clock, 100 MHz): debug_tests[16 + 2 b] = begin, [.. + 1] = end of
    // block b; the end stamp is taken behind a wait for the block's own stores (tools/cluster_tests.py)
    const unsigned long long t_begin = wall_clock64();
    unsigned long long dbg_filtered = 0, dbg_ctotal = 0;
    unsigned long long dbg_phase = 0; // the first item's phases, 16 bits each, in 10-ns ticks from t_begin (thread 0's view):
    uint32_t dbg_phases = 0;          // setup loads consumed | boxes united (barrier) | candidates filtered (barrier) | tested
#define ORBIT_PHASE_STAMP()                                                                                          \
    do {                                                                                                             \
        if (dbg_phases < 4u) dbg_phase |= ((wall_clock64() - t_begin) & 0xFFFFull) << (16u * dbg_phases), dbg_phases++; \
    } while (0)
#else
#define ORBIT_PHASE_STAMP() ((void)0)
#endif
    const uint32_t *hdr = reinterpret_cast<const uint32_t *>(p.unique);
    const uint32_t *unique_idx = reinterpret_cast<const uint32_t *>(p.unique + ORBIT_COMPACT_HEADER);
    uint32_t *out = reinterpret_cast<uint32_t *>(p.light_index_buffer + ORBIT_LIGHT_INDEX_HEADER);
    // Little arithmetic (config 4 executes 0.4 M tests here) in a great many short waves: what the launch costs is the
    // instructions its waves issue and the dependent round trips of a block.  The header word n is loaded
    // first but consumed only after the first item's loads that need no other load's result are in flight (indices are
    // clamped to the buffers instead of compared with n).  A group whose coarse segments are all short is one wave's work
    // (wave_item); in the four-wave form the first 16 entries of every segment are fetched before the segment counts are
    // known, which serves groups whose segments hold no more; up to 64 per segment take one fetched step, longer lists
    // the stepped path.
    const uint32_t n_raw = hdr[3];
    uint32_t n = 0;
    bool first_item = true;
    constexpr uint32_t kSeg = kLightTile / kAssignWaves; // LDS candidates per wave and step
    constexpr uint32_t kLoads = kSeg / 64u;              // coarse candidates per lane and step
    static_assert(kLoads * kAssignWaves == kCoarseSegs, "wave w owns coarse segments kLoads * w ..");
    // grid-stride over the blocks of kClustersPerBlock (8) active clusters that exist (see cluster_aabb_kernel); WRITE: over
    // the heavy blocks the count launch listed (p.block_base, p.total[0] of them; past the list: a block beyond every n)
    const uint32_t heavy_n = WRITE ? min(p.total[0], p.max_clusters / kClustersPerBlock + 1u) : 0u;
    uint32_t heavy_i = WRITE ? blockIdx.x - place_blocks : 0u;
    const uint32_t beyond = (p.max_clusters / kClustersPerBlock + 1u) * kClustersPerBlock; // >= n, inside the scratch padding
    auto heavy_at = [&](uint32_t i) { return i < heavy_n ? min(p.block_base[i] * kClustersPerBlock, beyond) : beyond; };
    // count launch: the blocks of the groups with the longest candidate lists first (p.group_order, cluster_coarse_kernel):
    // linear item L = rank of the group * 32 + block of the group.  An entry past the groups that exist is an old one:
    // a valid group index, used for nothing once n is known.
    constexpr uint32_t kGroupBlocks = kGroupClusters / kClustersPerBlock;
    uint32_t item = blockIdx.x;
    auto count_at = [&](uint32_t L) {
        return p.group_order[min(L / kGroupBlocks, p.max_clusters / kGroupClusters)] * kGroupClusters + (L % kGroupBlocks) * kClustersPerBlock;
    };
    uint32_t next_first = WRITE ? 0u : count_at(item);
    for (uint32_t block_first = WRITE ? heavy_at(heavy_i) : next_first;;
         block_first = WRITE ? heavy_at(heavy_i += gridDim.x - place_blocks) : next_first) {
    if (!WRITE) {
        if (!first_item && item >= (n + kGroupClusters - 1u) / kGroupClusters * kGroupBlocks) break;
        item += gridDim.x;
        next_first = count_at(item); // (in flight while this item is worked on)
        if (!first_item && block_first >= n) continue; // the last group's blocks beyond n
    } else if (!first_item && block_first >= n) {
        break;
    }
    const uint32_t u0 = block_first + wave * kClustersPerWave;
    const uint32_t group = block_first / kGroupClusters;
    const uint32_t *cbase = p.coarse + (size_t)group * kCoarseSegs * p.coarse_seg;
    const float4 *clbase = p.coarse_lights + (size_t)group * kCoarseSegs * p.coarse_seg; // (the lights, slot for slot)
    // ---- round trip 1: everything addressed by the block's position alone (count launch: before it is known whether
    // the four waves will need it — a group with long lists is the launch's critical path, one round trip shorter so)
    const uint32_t ul = min(u0 + ((uint32_t)lane & (kClustersPerWave - 1)), p.max_clusters - 1u); // lanes 0 .. kClustersPerWave - 1 count
    uint32_t my_ci = unique_idx[ul];
    Aabb3 box;
#pragma unroll
    for (int i = 0; i < 3; i++) box.mn[i] = p.aabb[6 * (size_t)ul + i], box.mx[i] = p.aabb[6 * (size_t)ul + 3 + i];
    if (!WRITE) {
        // Most blocks are light work — ten coarse candidates in their group at the median, a hundred at the 90th
        // percentile, three behind the block's own filter (config 4) — and the launch is bound by the instructions its
        // waves issue, not by their loads: four waves that each set up, unite boxes, filter a quarter of ten candidates
        // and walk four LDS segments to test two clusters spend four times the instructions of one wave doing all of
        // it.  So when no segment of the group holds more than kWaveSpec candidates, wave 0 takes the block's eight
        // clusters alone and the other waves leave the item (no barrier anywhere on this path).  Longer lists take the
        // four waves together, below.
        if (first_item) {
            n = min(n_raw, p.max_clusters);
            first_item = false;
            if (item - gridDim.x >= (n + kGroupClusters - 1u) / kGroupClusters * kGroupBlocks) break;
            if (block_first >= n) continue;
        }
        const uint32_t seg_counts = p.coarse_counts[(size_t)group * kCoarseSegs + ((uint32_t)lane & (kCoarseSegs - 1))];
        if (__ballot(seg_counts > kWaveSpec) == 0ull) { // wave-uniform, and the same in all four waves
            if (wave == 0) {
                ORBIT_PHASE_STAMP();
                // (fetched only now: issued beside the counts these loads were measured no faster — 62.8 against 62.0 us
                // for the chain — and their registers, live across the branch, spill under this kernel's budget)
                WaveItem wi;
                wave_item_loads(p, wi, block_first, unique_idx, cbase, clbase, lane);
                wave_item(p, wi, block_first, n, seg_counts, s_cand, s_cand_idx, &s_ibox[0][0], lane);
                ORBIT_PHASE_STAMP();
            }
            continue;
        }
    }
    const uint32_t cc = p.coarse_counts[(size_t)group * kCoarseSegs + ((uint32_t)lane & (kCoarseSegs - 1))];
    uint32_t my_limit = 0, my_offset = 0;
    if (WRITE) {
        my_limit = p.counts[ul];
        // exclusive prefix of the capped counts: the sums of the chunks before the block's (its 8 clusters share a
        // chunk), the counts of the chunk's clusters before the wave's first one, the wave's own clusters before the lane's
        static_assert(kScanChunk % kClustersPerBlock == 0, "a block's clusters share a scan chunk");
        const uint32_t chunk = min(block_first, p.max_clusters) / kScanChunk;
        uint32_t part = 0;
        for (uint32_t k = (uint32_t)lane; k < chunk; k += 64u) part += p.block_sums[(size_t)k * kChunkSumStride];
        uint32_t cv[kScanChunk / 64u]; // (loaded together: a loop of load-and-add is sixteen round trips in a row)
#pragma unroll
        for (uint32_t j = 0; j < kScanChunk / 64u; j++) {
            const uint32_t v = chunk * kScanChunk + j * 64u + (uint32_t)lane;
            cv[j] = v < min(u0, p.max_clusters) ? p.counts[v] : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < kScanChunk / 64u; j++) part += cv[j];
        const uint32_t mine = lane < kClustersPerWave ? my_limit : 0u;
        my_offset = wave_reduce_add(part) + (wave_inclusive_scan(mine) - mine);
    }
    uint32_t tag[kLoads];
    float4 l[kLoads];
    // speculative: the first kSpec (16) entries of the wave's four segments, one per lane (lane = segment * 16 + entry)
    // — a block's group has ten candidates at the median and a hundred at the 90th percentile (config 4), a few per
    // segment; groups with a longer segment fetch theirs once the counts are known
    constexpr uint32_t kSpec = 64u / kLoads;
    const uint32_t spec_seg = wave * kLoads + (uint32_t)lane / kSpec, spec_entry = (uint32_t)lane % kSpec;
    if (!WRITE) { // (in the first of the four register sets the other paths fill)
        tag[0] = cbase[spec_seg * p.coarse_seg + spec_entry];
        l[0] = clbase[spec_seg * p.coarse_seg + spec_entry];
    }
    if (first_item) { // (WRITE: the count launch's first item found n above)
        n = min(n_raw, p.max_clusters);
        first_item = false;
        if (block_first >= n) break;
    }
    const bool owns = lane < kClustersPerWave && u0 + lane < n;
    if (wave == 0) { // exclusive prefix of the 16 segment counts (wave scan); fast = no segment longer than a wave
        const uint32_t v = lane < (int)kCoarseSegs ? cc : 0u;
        const uint32_t inc = wave_inclusive_scan(v);
        if (lane < (int)kCoarseSegs) s_cpre[lane] = inc - v, s_ccnt[lane] = v;
        if (lane == (int)kCoarseSegs - 1) s_cpre[kCoarseSegs] = inc;
        const bool lng = __ballot(v > 64u) != 0ull, mid = __ballot(v > kSpec) != 0ull;
        if (lane == 0) s_fast = lng ? 0u : mid ? 1u : 2u; // 2: every segment within the speculative fetch
    }
    ORBIT_PHASE_STAMP();
    // absent cluster: the empty box, neutral in the union
    {
        const float inf = __uint_as_float(0x7f800000u);
#pragma unroll
        for (int i = 0; i < 3; i++) box.mn[i] = owns ? box.mn[i] : inf, box.mx[i] = owns ? box.mx[i] : -inf;
    }
    // the wave's boxes into scalar registers; their union over the wave (lanes 0 .. kClustersPerWave - 1 hold them), then over the block.
    // With a NaN anywhere the filter is bypassed (flag), so what the selects make of a NaN operand does not matter.
    float mn[kClustersPerWave][3], mx[kClustersPerWave][3];
    uint32_t offset[kClustersPerWave], limit[kClustersPerWave], count[kClustersPerWave];
    bool valid[kClustersPerWave];
#pragma unroll
    for (int c = 0; c < kClustersPerWave; c++) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            mn[c][i] = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(box.mn[i]), c));
            mx[c][i] = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(box.mx[i]), c));
        }
        offset[c] = (uint32_t)__builtin_amdgcn_readlane((int)my_offset, c);
        limit[c] = (uint32_t)__builtin_amdgcn_readlane((int)my_limit, c);
        count[c] = 0;
        valid[c] = u0 + c < n;
    }
    {
        bool nan = false;
        float lo[3], hi[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            lo[i] = box.mn[i], hi[i] = box.mx[i];
            nan = nan || lo[i] != lo[i] || hi[i] != hi[i];
        }
        const bool wave_nan = (__ballot(nan) & ((1ull << kClustersPerWave) - 1ull)) != 0ull;
#pragma unroll
        for (int d = kClustersPerWave / 2; d >= 1; d >>= 1) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const float ol = __shfl_xor(lo[i], d, 64), oh = __shfl_xor(hi[i], d, 64);
                lo[i] = ol < lo[i] ? ol : lo[i];
                hi[i] = oh > hi[i] ? oh : hi[i];
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 3; i++) s_wbox[wave][i] = lo[i], s_wbox[wave][3 + i] = hi[i];
            s_wbox[wave][6] = wave_nan ? 1.0f : 0.0f;
        }
    }
    __syncthreads();
    ORBIT_PHASE_STAMP();
    BoxUnion un;
    un.any_nan = false;
    {
        const float inf = __uint_as_float(0x7f800000u);
#pragma unroll
        for (int i = 0; i < 3; i++) un.box.mn[i] = inf, un.box.mx[i] = -inf;
    }
#pragma unroll
    for (int w = 0; w < kAssignWaves; w++) {
        union_add(un, &s_wbox[w][0], &s_wbox[w][3]);
        un.any_nan = un.any_nan || s_wbox[w][6] != 0.0f;
    }
    const uint32_t ctotal = s_cpre[kCoarseSegs];
    const bool fast = !WRITE && s_fast != 0u, spec = !WRITE && s_fast == 2u; // block-uniform

    [[maybe_unused]] uint32_t n_filter = 0, n_test = 0;
    ORBIT_MAX_TESTS(7, ctotal);

    bool have[kLoads];
    // One step: the wave filters its (up to) kSeg candidates {tag, l, have} into LDS segment `wave`, in light order
    // (filter_step); then, behind a barrier, every wave tests its clusters against the four segments in order
    // (test_step).
    uint32_t filter_rounds = kLoads; // (block-uniform; 1: the speculative fetch, whose one round holds all four segments)
    auto filter_step = [&]() {
        uint32_t seg_n = 0; // wave-uniform
#pragma unroll
        for (uint32_t r = 0; r < kLoads; r++) {
            if (r >= filter_rounds) break;
            const bool point = (tag[r] & kNotPoint) == 0u;
            const bool pass = have[r] && (!point || un.any_nan || sphere_hits(un.box, l[r]));
            const uint64_t m = __ballot(pass);
#ifdef ORBIT_TRIAGE
            n_filter += (uint32_t)__popcll(__ballot(have[r] && point && !un.any_nan));
#endif
            if (pass) {
                const uint32_t pos = wave * kSeg + seg_n + lane_prefix(m);
                s_cand[pos] = l[r];
                s_cand_idx[pos] = tag[r];
            }
            seg_n += (uint32_t)__popcll(m);
        }
        if (lane == 0) s_wave_cnt[wave] = seg_n;
    };
    // A wave with many candidates (a hot spot: a thousand lights over one block's clusters) is a serial loop of
    // sphere-box tests at one wave per SIMD, and the launch lasts as long as its slowest block: the eight tests of a
    // batch are straight-line code (eight independent chains, no branch between them), the ordered writes follow.
    auto test_step = [&]() {
#ifdef ORBIT_TRIAGE_STAMPS
        dbg_filtered += s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
        dbg_ctotal = ctotal;
#endif
        // the four waves' segments back to back (= light order) as ONE list: a block with twenty survivors spread over
        // four segments tests one batch of 64 lanes, not four
        static_assert(kAssignWaves == 4, "three prefixes place a candidate in its segment");
        const uint32_t end0 = s_wave_cnt[0], end1 = end0 + s_wave_cnt[1], end2 = end1 + s_wave_cnt[2], cnt = end2 + s_wave_cnt[3];
        {
            for (uint32_t b0 = 0; b0 < cnt; b0 += 64) {
                const uint32_t ci = b0 + lane;
                const bool in_list = ci < cnt;
                const uint32_t seg = (ci >= end0 ? 1u : 0u) + (ci >= end1 ? 1u : 0u) + (ci >= end2 ? 1u : 0u);
                const uint32_t at = in_list ? seg * kSeg + (ci - (seg == 0u ? 0u : seg == 1u ? end0 : seg == 2u ? end1 : end2)) : 0u;
                const float4 cl = s_cand[at];
                const uint32_t ctag = s_cand_idx[at];
                const bool point = (ctag & kNotPoint) == 0u;
                uint64_t hits[kClustersPerWave];
#pragma unroll
                for (int c = 0; c < kClustersPerWave; c++) {
                    Aabb3 a;
#pragma unroll
                    for (int i = 0; i < 3; i++) a.mn[i] = mn[c][i], a.mx[i] = mx[c][i];
                    hits[c] = __ballot(in_list && (!point || sphere_hits(a, cl))); // :108-119
                }
#ifdef ORBIT_TRIAGE
                n_test += (uint32_t)__popcll(__ballot(in_list && point)) * (uint32_t)kClustersPerWave;
#endif
#pragma unroll
                for (int c = 0; c < kClustersPerWave; c++) {
                    const uint64_t m = hits[c];
                    if (!valid[c] || m == 0ull) continue; // wave-uniform
                    if (WRITE && limit[c] <= kHitCache) continue; // (its list was copied by a placement block)
                    if (WRITE || count[c] < kHitCache) {  // wave-uniform: the counting launch keeps the first few only
                        if ((m >> lane) & 1ull) {
                            const uint32_t k = count[c] + lane_prefix(m);
                            if (WRITE) {
                                if (k < limit[c] && offset[c] + k < p.light_index_capacity) // while (counter < light_count), :140
                                    out[offset[c] + k] = ctag & ~kNotPoint;
                            } else if (k < kHitCache) {
                                p.hit_cache[(size_t)(u0 + c) * kHitCache + k] = ctag & ~kNotPoint;
                            }
                        }
                    }
                    count[c] += (uint32_t)__popcll(m);
                }
            }
        }
    };
    if (fast) {
        if (spec) {
            // no second round trip: the candidates came with their lights (an entry past its segment's count is a stale
            // one of an earlier call: never used).  One filter round: lanes in segment order = light order.
            have[0] = spec_entry < s_ccnt[spec_seg];
            filter_rounds = 1u;
        } else {
            // segments of up to 64: one step, fetched now that the counts are known (index and light, slot for slot)
#pragma unroll
            for (uint32_t r = 0; r < kLoads; r++) {
                have[r] = (uint32_t)lane < s_ccnt[wave * kLoads + r];
                const uint32_t slot = (wave * kLoads + r) * p.coarse_seg + (have[r] ? (uint32_t)lane : 0u); // (< 16 segments of caps.max_lights / 16)
                tag[r] = cbase[slot];
                l[r] = clbase[slot];
            }
        }
        filter_step();
        __syncthreads(); // all segments complete
        ORBIT_PHASE_STAMP();
        test_step();
        ORBIT_PHASE_STAMP();
    } else {
        // The group's candidates in light order = its coarse segments back to back.  A step takes the next kLightTile
        // entries of that sequence; wave w filters the w-th quarter of the step into LDS segment w.  The next step's
        // loads are issued before this step's test phase.
        auto coarse_at = [&](uint32_t k) -> uint32_t { // slot of the k-th candidate of the group, k < ctotal
            uint32_t sg = 0; // last segment whose prefix is <= k (bisection over the 16 prefixes)
#pragma unroll
            for (uint32_t half = kCoarseSegs / 2; half >= 1; half /= 2) sg += s_cpre[sg + half] <= k ? half : 0u;
            return sg * p.coarse_seg + (k - s_cpre[sg]);
        };
        auto load_step = [&](uint32_t step0) {
            const uint32_t s_begin = step0 + wave * kSeg, s_end = min(s_begin + kSeg, ctotal);
#pragma unroll
            for (uint32_t r = 0; r < kLoads; r++) {
                const uint32_t k = s_begin + r * 64u + (uint32_t)lane;
                have[r] = k < s_end;
                const uint32_t slot = have[r] ? coarse_at(k) : 0u; // one address for the lanes without a candidate
                tag[r] = cbase[slot];
                l[r] = clbase[slot];
            }
        };
        if (ctotal > 0) load_step(0);
        for (uint32_t step0 = 0; step0 < ctotal; step0 += kLightTile) { // block-uniform trip count
            filter_step();
            if (step0 + kLightTile < ctotal) load_step(step0 + kLightTile); // in flight during the test phase
            __syncthreads(); // all segments complete
            ORBIT_PHASE_STAMP();
            test_step();
            __syncthreads(); // candidates consumed before the next step's filter overwrites them
            ORBIT_PHASE_STAMP();
        }
    }
    ORBIT_COUNT_TESTS(WRITE ? 4 : 2, n_filter);
    ORBIT_COUNT_TESTS(WRITE ? 5 : 3, n_test);
    {   // lane c publishes cluster c's result
        uint32_t mine = 0, lim = 0, off = 0;
#pragma unroll
        for (int c = 0; c < kClustersPerWave; c++)
            if (lane == c) mine = count[c], lim = limit[c], off = offset[c];
        const uint32_t capped = owns ? min(mine, ORBIT_MAX_LIGHTS_PER_CLUSTER) : 0u; // :135
        if (owns) {
            if (!WRITE) {
                p.counts[u0 + lane] = capped;
            } else if (lim > kHitCache) {
                // imageStore(cluster_offset_image, cluster_id, (offset, count)), :149-150
                p.offset_image[2 * (size_t)my_ci + 0] = off;
                p.offset_image[2 * (size_t)my_ci + 1] = lim;
            }
        }
        if (!WRITE) { // the chunk's sum (what every offset behind it is made of) and the list of heavy blocks
            if (lane < kClustersPerWave) s_wtot[wave * kClustersPerWave + lane] = capped;
        }
    }
    if (!WRITE) {
        __syncthreads(); // (also: s_cpre / s_wbox are rewritten by the next block of clusters)
        if (threadIdx.x == 0) {
            // ONE atomic per block, and every chunk's word on a line of its own: atomics on one 128-B line queue in one
            // L2 channel whatever their words (17 000 of them on the 34 adjacent words of config 4: 85 us)
            uint32_t block_total = 0, most = 0;
#pragma unroll
            for (int c = 0; c < kClustersPerBlock; c++) block_total += s_wtot[c], most = max(most, s_wtot[c]);
            if (block_total != 0u)
                (void)__hip_atomic_fetch_add(p.block_sums + (size_t)(block_first / kScanChunk) * kChunkSumStride, block_total,
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (most > kHitCache)
                p.block_base[__hip_atomic_fetch_add(p.total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = block_first / kClustersPerBlock;
        }
    }
    if (WRITE) __syncthreads(); // s_cpre / s_wbox are rewritten by the next block of clusters
    }
#ifdef ORBIT_TRIAGE_STAMPS
    if (!WRITE && p.debug_tests && threadIdx.x == 0 && blockIdx.x < 4096u) {
        __builtin_amdgcn_s_waitcnt(0);
        p.debug_tests[16 + 2 * blockIdx.x] = t_begin;
        p.debug_tests[16 + 2 * blockIdx.x + 1] = wall_clock64();
        p.debug_tests[16 + 2 * (4096u + blockIdx.x)] = dbg_phase; // (fast path only: four phases; else what was passed)
        // candidates behind the block's filter (wave 0's view: all segments) << 32 | the group's coarse candidates
        p.debug_tests[16 + 2 * (4096u + blockIdx.x) + 1] = dbg_filtered << 32 | dbg_ctotal;
    }
#endif
}

// header for an empty active list: light_count = 0 (cluster.rs:575-581)
__global__ void write_word(uint32_t *p, uint32_t v) { *p = v; }

// orbit_debug_log2_guard (tests): depth_slice against depth_slice_canonical for EVERY float bit pattern in [lo, hi] —
// out[0] = mismatches, out[1] = samples the canonical path decided, out[2] = bits of the largest
// |v_log_f32 - log2c| / (1 + |log2c|) over the positive normal inputs (the guard band's premise: < 2^-22)
__global__ __launch_bounds__(256) void log2_guard_kernel(uint32_t lo, uint32_t hi, float z_scale, float z_bias,
                                                         unsigned long long *out) {
    unsigned long long bad = 0, slow = 0;
    float worst = 0.0f;
    const uint64_t n = (uint64_t)hi - lo + 1u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        const float x = __uint_as_float(lo + (uint32_t)i);
        const uint32_t a = depth_slice(x, z_scale, z_bias), b = depth_slice_canonical(x, z_scale, z_bias);
        bad += a != b ? 1u : 0u;
        const float L = __builtin_amdgcn_logf(x);
        slow += depth_slice_safe(x, L, __builtin_fmaf(L, z_scale, z_bias), z_scale) ? 0u : 1u;
        if (x >= 0x1p-126f && x < __builtin_inff()) {
            const float c = log2c(x);
            worst = fmaxf(worst, __builtin_fabsf(L - c) / (1.0f + __builtin_fabsf(c)));
        }
    }
    atomicAdd(out + 0, bad);
    atomicAdd(out + 1, slow);
    atomicMax(reinterpret_cast<unsigned int *>(out + 2), __float_as_uint(worst));
}

} // namespace

hipError_t launch_log2_guard_check(uint32_t lo, uint32_t hi, float z_scale, float z_bias, unsigned long long *out, hipStream_t s) {
    hipLaunchKernelGGL(log2_guard_kernel, dim3(256 * 16), dim3(256), 0, s, lo, hi, z_scale, z_bias, out);
    return hipGetLastError();
}

hipError_t launch_cluster_mark(const ClusterMarkParams &p, hipStream_t s) {
    const uint32_t tiles = p.pc.cluster_count[0] * p.pc.cluster_count[1];
    if (tiles == 0) return hipSuccess;
    const uint32_t per_block = kMarkWaves * kMarkTiles;
    hipLaunchKernelGGL(cluster_mark_kernel, dim3((tiles + per_block - 1) / per_block), dim3(kMarkWaves * 64), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_cluster_compact(const ClusterCompactParams &p, hipStream_t s) {
    const uint32_t tiles = p.cc[0] * p.cc[1];
    const uint32_t nchunks = (tiles + 1023) / 1024;
    const uint32_t cz = p.cc[2];
    if (nchunks == 0 || cz == 0) {
        hipLaunchKernelGGL(write_word, dim3(1), dim3(1), 0, s, reinterpret_cast<uint32_t *>(p.unique) + 3, 0u);
        return hipGetLastError();
    }
    if (!p.counted) {
        hipLaunchKernelGGL(cluster_compact_kernel<false>, dim3(nchunks, cz), dim3(1024), 0, s, p);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(cluster_compact_kernel<true>, dim3(nchunks, cz), dim3(1024), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_cluster_assign(const ClusterAssignParams &p, uint32_t num_cus, hipStream_t s) {
    const uint32_t nl = p.info.global_light_count;
    const uint32_t prepare_blocks = (nl + 255u) / 256u;
    if (p.max_clusters == 0) {
        hipLaunchKernelGGL(write_word, dim3(1), dim3(1), 0, s, reinterpret_cast<uint32_t *>(p.light_index_buffer), 0u);
        return hipGetLastError();
    }
    // the number of active clusters is only known on the device: grids are bounded by the chip, not by the capacity,
    // and the kernels stride over the work that exists
    const uint32_t groups = (p.max_clusters + kGroupClusters - 1) / kGroupClusters;
    const uint32_t blocks = min((p.max_clusters + kClustersPerBlock - 1) / kClustersPerBlock, num_cus * 32u);
    hipLaunchKernelGGL(cluster_aabb_kernel, dim3(prepare_blocks + 1u + min(groups, num_cus * 4u)), dim3(256), 0, s, p, prepare_blocks);
    hipLaunchKernelGGL(cluster_coarse_kernel, dim3(min(groups * kCoarseSplit, num_cus * 8u)), dim3(256), 0, s, p);
    hipLaunchKernelGGL(cluster_assign_kernel<false>, dim3(blocks), dim3(kAssignWaves * 64), 0, s, p, 0u);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // placement blocks (256 clusters each) + workers for the heavy blocks (they leave at once when there are none)
    const uint32_t place_blocks = min((p.max_clusters + 255u) / 256u, num_cus * 2u);
#ifndef ORBIT_HEAVY_WORKERS
#define ORBIT_HEAVY_WORKERS num_cus
#endif
    hipLaunchKernelGGL(cluster_assign_kernel<true>, dim3(place_blocks + ORBIT_HEAVY_WORKERS), dim3(kAssignWaves * 64), 0, s, p, place_blocks);
    return hipGetLastError();
}

} // namespace orbit
