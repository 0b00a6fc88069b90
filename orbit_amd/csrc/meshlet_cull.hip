// meshlet_cull.hip — per-meshlet frustum / normal-cone / HiZ cull and ordered
// compaction into the indirect-draw list, for gfx950 (wave64).
//
// Restates shaders/meshlet_cull.comp:108-255 (reference repo) under the launch
// protocol of src/passes/draw_gen.rs:382-435.  MI355X-first structure:
//
//   wave tile  = 16 dispatch records = 8 rows x (2 records on 64 lanes), one
//                meshlet per lane per row -> every row is one fully coalesced
//                2 KiB read of 32-B meshlets (two dwordx4 per lane).
//   setup      = the 16 records' (view x model) matrices are built ONCE per tile
//                with all 64 lanes (lane = record*4 + column) and staged in a
//                wave-private LDS slab; rows then read them as broadcast
//                ds_read_b128.  The reference recomputes the 4x4 product in
//                every one of the 32 lanes of a workgroup (meshlet_cull.comp:117).
//   compaction = __ballot + mbcnt lane prefix inside the row, row prefix inside
//                the tile, tile base from a prefix sum over tiles -> the command
//                sequence is the canonical (record, lane) order, no atomics.
//   visibility = the two 32-bit halves of the wave64 ballot are the two records'
//                visibility words (word = visibility_offset + lane/32, S = 32).
//
// Tile bases come from either (mode 0) a count launch + scan + emit launch,
// or (mode 1) a single launch with a decoupled look-back over tile granules.
#include "kernels.h"
#include "scan.h"

namespace orbit {

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWavesPerBlock * 64;

enum { MODE_COUNT = 0, MODE_EMIT = 1 };

struct __attribute__((aligned(16))) WaveTileLds {
    float4 mcol[kTileRecords][4]; // view x model, column-major, per record
    uint4 rec[kTileRecords];      // MeshletDispatch
    float scale[kTileRecords];    // largest_scale_from_matrix(mat3(M))
};

// One meshlet (one lane): meshlet_cull.comp:113-213.  Returns bit0 = visible,
// bit1 = should_draw.
__device__ __forceinline__ uint32_t eval_meshlet(const MeshletCullParams &p, const WaveTileLds &L, uint32_t rid,
                                                 const uint4 &a, const uint4 &b, bool visible_in_buffer,
                                                 bool meshlet_occ) {
    const OrbitGpuCullInfo &ci = p.ci;
    const float4 m0 = L.mcol[rid][0], m1 = L.mcol[rid][1], m2 = L.mcol[rid][2], m3 = L.mcol[rid][3];
    const float cx = __uint_as_float(a.x), cy = __uint_as_float(a.y), cz = __uint_as_float(a.z);
    // transform_sphere (:37-43): matrix * vec4(c, 1), divide by w.
    float px = ((m0.x * cx + m1.x * cy) + m2.x * cz) + m3.x * 1.0f;
    float py = ((m0.y * cx + m1.y * cy) + m2.y * cz) + m3.y * 1.0f;
    float pz = ((m0.z * cx + m1.z * cy) + m2.z * cz) + m3.z * 1.0f;
    const float pw = ((m0.w * cx + m1.w * cy) + m2.w * cz) + m3.w * 1.0f;
    if (pw != 1.0f) { // x / 1.0f == x bit-for-bit, so the affine case skips three IEEE divisions
        px = px / pw;
        py = py / pw;
        pz = pz / pw;
    }
    Sphere s;
    s.x = px;
    s.y = py;
    s.z = pz;
    s.r = __uint_as_float(a.w) * L.scale[rid];

    // cone axis / cutoff (:121-122): int(i8) / 127.0
    const int ax_i = (int)(int8_t)(b.x & 0xFF), ay_i = (int)(int8_t)((b.x >> 8) & 0xFF);
    const int az_i = (int)(int8_t)((b.x >> 16) & 0xFF), co_i = (int)(int8_t)(b.x >> 24);
    const float ax = snorm8_div127(ax_i), ay = snorm8_div127(ay_i), az = snorm8_div127(az_i);
    const float cutoff = snorm8_div127(co_i);
    const float axx = ((m0.x * ax + m1.x * ay) + m2.x * az) + m3.x * 0.0f;
    const float axy = ((m0.y * ax + m1.y * ay) + m2.y * az) + m3.y * 0.0f;
    const float axz = ((m0.z * ax + m1.z * ay) + m2.z * az) + m3.z * 0.0f;

    const uint32_t material_index = b.w & 0xFFFFu;
    const uint32_t alpha_mode = p.materials[material_index].alpha_mode; // :124

    bool visible = true;
    if (ci.occlusion_pass == 1) visible = visible_in_buffer; // :137
    if (visible) visible = plane_test(ci, s);                 // :139-146
    if (visible) {                                            // :148-158
        float camx = 0.0f, camy = 0.0f, camz = 0.0f;
        if (ci.projection_type == 1) {
            camx = s.x - 0.0f;
            camy = s.y - 0.0f;
            camz = s.z - -1.0f;
        }
        const float dx = s.x - camx, dy = s.y - camy, dz = s.z - camz;
        const bool cull = dot3(dx, dy, dz, axx, axy, axz) >= cutoff * sqrtf(dot3(dx, dy, dz, dx, dy, dz)) + s.r;
        visible = !cull;
    }
    if (meshlet_occ && ci.occlusion_pass == 2 && visible) visible = occlusion_test(ci, s, p.pyr); // :161-205

    bool should_draw = visible && ((shl1(alpha_mode) & ci.alpha_mode_flag) != 0); // :207
    if ((ci.occlusion_pass == 2 && meshlet_occ) && !((shl1(alpha_mode) & ci.noskip_alphamode) != 0))
        should_draw = visible && !visible_in_buffer; // :210-213
    return (visible ? 1u : 0u) | (should_draw ? 2u : 0u);
}

// Builds the tile's 16 (view x model) matrices with all 64 lanes.
__device__ __forceinline__ void tile_setup(const MeshletCullParams &p, WaveTileLds &L, uint32_t tile, uint32_t nrec,
                                           int lane) {
    const OrbitMeshletDispatch *recs =
        reinterpret_cast<const OrbitMeshletDispatch *>(p.dispatch_buffer + ORBIT_DISPATCH_HEADER);
    const uint32_t rid = lane >> 2, col = lane & 3;
    const uint32_t rec_index = tile * kTileRecords + rid;
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    float4 mc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rec_index < nrec) {
        rec = *reinterpret_cast<const uint4 *>(recs + rec_index);
        mc = reinterpret_cast<const float4 *>(p.entities[rec.x].model_matrix)[col];
    }
    float o[4];
    mat4_mul_col(p.ci.view_matrix, mc.x, mc.y, mc.z, mc.w, o);
    L.mcol[rid][col] = make_float4(o[0], o[1], o[2], o[3]);
    const float d = dot3(o[0], o[1], o[2], o[0], o[1], o[2]);
    const float d0 = __shfl(d, (int)(rid * 4 + 0), 64);
    const float d1 = __shfl(d, (int)(rid * 4 + 1), 64);
    const float d2 = __shfl(d, (int)(rid * 4 + 2), 64);
    if (col == 0) {
        L.scale[rid] = sqrtf(gmax(d0, gmax(d1, d2)));
        L.rec[rid] = rec;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Evaluates the 8 rows of a tile.  Fills the should-draw masks; WRITE_VIS also
// writes the pass-2 visibility words.
template <bool WRITE_VIS>
__device__ __forceinline__ uint32_t tile_rows(const MeshletCullParams &p, const WaveTileLds &L, int lane,
                                              uint64_t draw_mask[kTileRows]) {
    const OrbitGpuCullInfo &ci = p.ci;
    const bool meshlet_occ = ci.meshlet_visibility_buffer != ORBIT_NONE;
    const bool rw_vis = (ci.occlusion_pass == 1 || ci.occlusion_pass == 2) && meshlet_occ;
    const uint32_t half = lane >> 5, ml = lane & 31;
    uint32_t total = 0;

    // issue all meshlet loads of the tile first: 16 x 16 B per lane in flight
    uint4 a[kTileRows], b[kTileRows];
    uint32_t prev[kTileRows];
#pragma unroll
    for (int r = 0; r < (int)kTileRows; r++) {
        const uint4 rec = L.rec[2 * r + half];
        a[r] = make_uint4(0u, 0u, 0u, 0u);
        b[r] = make_uint4(0u, 0u, 0u, 0u);
        prev[r] = 0u;
        if (ml < rec.z) { // :111
            const uint4 *src = reinterpret_cast<const uint4 *>(p.meshlets + (rec.y + ml));
            a[r] = src[0];
            b[r] = src[1];
            if (rw_vis) prev[r] = p.meshlet_visibility[rec.w]; // word = visibility_offset + lane/32, :129-134
        }
    }
#pragma unroll
    for (int r = 0; r < (int)kTileRows; r++) {
        const uint32_t rid = 2 * r + half;
        const uint4 rec = L.rec[rid];
        const bool active = ml < rec.z;
        uint32_t res = 0;
        if (active) {
            const bool visible_in_buffer = rw_vis ? ((prev[r] >> ml) & 1u) != 0 : true;
            res = eval_meshlet(p, L, rid, a[r], b[r], visible_in_buffer, meshlet_occ);
        }
        const uint64_t vis_mask = __ballot((res & 1u) != 0);
        const uint64_t mask = __ballot((res & 2u) != 0);
        draw_mask[r] = mask;
        total += (uint32_t)__popcll(mask);
        if (WRITE_VIS) {
            // pass-2 visibility words (:233-254); records with no lane write nothing
            if (ci.occlusion_pass == 2 && meshlet_occ && ml == 0 && rec.z > 0)
                p.meshlet_visibility[rec.w] = half ? (uint32_t)(vis_mask >> 32) : (uint32_t)vis_mask;
        }
    }
    return total;
}

// Writes the tile's surviving commands at [base, base + total) (:216-230).
// The survivor's command words (Meshlet bytes 16..31) are re-read here: they
// were streamed a few microseconds ago and are L2/MALL resident, and not
// carrying them across the look-back keeps the kernel at 4 waves per SIMD.
__device__ __forceinline__ void tile_emit(const MeshletCullParams &p, const WaveTileLds &L, int lane, uint32_t base,
                                          const uint64_t draw_mask[kTileRows]) {
    uint32_t *out = reinterpret_cast<uint32_t *>(p.draw_buffer + ORBIT_DRAW_HEADER);
    const uint32_t half = lane >> 5, ml = lane & 31;
    uint32_t row_base = base;
#pragma unroll
    for (int r = 0; r < (int)kTileRows; r++) {
        const uint64_t mask = draw_mask[r];
        if ((mask >> lane) & 1ull) {
            const uint32_t slot = row_base + lane_prefix(mask);
            if (slot < p.draw_capacity) {
                const uint4 rec = L.rec[2 * r + half];
                const uint4 b = reinterpret_cast<const uint4 *>(p.meshlets + (rec.y + ml))[1];
                const uint32_t vertex_count = (b.w >> 16) & 0xFFu;
                const uint32_t triangle_count = b.w >> 24;
                uint32_t *c = out + (size_t)slot * 7u;
                c[0] = triangle_count * 3u;           // cmd_index_count
                c[1] = 1u;                            // cmd_instance_count
                c[2] = (b.z + vertex_count) * 4u;     // cmd_first_index
                c[3] = b.z;                           // cmd_vertex_offset = int(data_offset)
                c[4] = rec.x;                         // cmd_first_instance = entity_index
                c[5] = b.y;                           // meshlet_vertex_offset
                c[6] = rec.y + ml;                    // meshlet_index
            }
        }
        row_base += (uint32_t)__popcll(mask);
    }
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void meshlet_cull_two_pass(const MeshletCullParams p) {
    __shared__ WaveTileLds lds[kWavesPerBlock];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    WaveTileLds &L = lds[wave];
    const uint32_t nrec = min(*reinterpret_cast<const uint32_t *>(p.dispatch_buffer), p.dispatch_capacity);
    const uint32_t ntiles = (nrec + kTileRecords - 1) / kTileRecords;
    for (uint32_t tile = blockIdx.x * kWavesPerBlock + wave; tile < ntiles; tile += gridDim.x * kWavesPerBlock) {
        tile_setup(p, L, tile, nrec, lane);
        uint64_t draw_mask[kTileRows];
        const uint32_t total = tile_rows<MODE == MODE_EMIT>(p, L, lane, draw_mask);
        if (MODE == MODE_COUNT) {
            if (lane == 0) p.tile_counts[tile] = total;
        } else {
            tile_emit(p, L, lane, p.tile_base[tile], draw_mask);
        }
        __builtin_amdgcn_wave_barrier(); // LDS slab is reused by the next tile
    }
    if (MODE == MODE_EMIT && blockIdx.x == 0 && threadIdx.x == 0) {
        // `count` header: subsumes the fill_buffer(count = 0) of draw_gen.rs:411-417
        const uint32_t total = *p.total;
        *reinterpret_cast<uint32_t *>(p.draw_buffer) = min(total, p.draw_capacity);
        if (total > p.draw_capacity) latch_status(p.status, ORBIT_E_CAPACITY);
    }
}

// ---------------------------------------------------------------------------
// Single-pass mode: one launch, ordered compaction by a decoupled look-back
// scan over per-wave-tile granules.
//
//  * tile order comes from a ticket per BLOCK iteration (kSpWaves adjacent wave
//    tiles per ticket): one returning atomic on one address saturates near 88/us
//    on MI355X (MI355X_MICROARCH.md "dequeue"), so a ticket per 512-meshlet wave
//    tile (98 k at 50 M meshlets) would cost more than the whole kernel; a ticket
//    per 8192 meshlets is ~20/us.  A block that waits on tile j only ever waits
//    on tickets drawn before its own, i.e. on running or finished blocks — no
//    residency or dispatch-order assumption.
//  * granule = one naturally aligned 8-byte {epoch:30 | status:2 | value:32}
//    written by ONE relaxed agent-scope store and polled with relaxed
//    agent-scope loads (cdna_hip_programming.md Guideline 16, recipe R2: the
//    data is the flag, no fence).  The epoch (a device word bumped by the last
//    block to finish) makes stale granules of earlier launches read as INVALID,
//    so no per-launch memset and the launch is graph-replay safe.
//  * every spin is bounded; on give-up ORBIT_E_TIMEOUT is latched and the tile
//    proceeds with base 0 (wrong output, flagged) instead of hanging the GPU.
constexpr int kSpWaves = 16;
constexpr int kSpBlock = kSpWaves * 64;
constexpr uint32_t ST_INVALID = 0u, ST_AGG = 1u, ST_PREFIX = 2u;
constexpr uint32_t kSpinLimit = 1u << 22;

__device__ __forceinline__ uint64_t make_granule(uint32_t epoch, uint32_t status, uint32_t value) {
    return ((uint64_t)epoch << 34) | ((uint64_t)status << 32) | (uint64_t)value;
}

// Exclusive prefix of wave tile `w` (> 0): sums predecessor aggregates back to
// the nearest published inclusive prefix, 64 granules per step.
__device__ __forceinline__ uint32_t look_back(const MeshletCullParams &p, uint32_t w, uint32_t epoch, int lane) {
    uint32_t exclusive = 0;
    int64_t pos = (int64_t)w - 1; // nearest predecessor handled by lane 0
    uint32_t spins = 0;
    for (;;) {
        const int64_t idx = pos - lane;
        const bool in_range = idx >= 0;
        uint64_t g = 0;
        if (in_range) g = __hip_atomic_load(p.scan_state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t st = (uint32_t)(g >> 32) & 3u;
        if ((uint32_t)(g >> 34) != epoch) st = ST_INVALID;
        if (!in_range) st = ST_PREFIX; // "before tile 0": an inclusive prefix of 0
        const uint32_t val = in_range ? (uint32_t)g : 0u;
        const uint64_t m_prefix = __ballot(st == ST_PREFIX);
        const uint64_t m_invalid = __ballot(st == ST_INVALID);
        const int first_prefix = m_prefix ? (int)__builtin_ctzll(m_prefix) : 64;
        const int first_invalid = m_invalid ? (int)__builtin_ctzll(m_invalid) : 64;
        if (first_invalid < first_prefix) { // a needed predecessor has not published yet
            if (++spins > kSpinLimit) {
                if (lane == 0) latch_status(p.status, ORBIT_E_TIMEOUT);
                return 0u;
            }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        const uint32_t contrib = (lane <= first_prefix) ? val : 0u;
        exclusive += wave_reduce_add(contrib);
        if (first_prefix < 64) return exclusive;
        pos -= 64;
    }
}

__global__ __launch_bounds__(kSpBlock) void meshlet_cull_single_pass(const MeshletCullParams p) {
    __shared__ WaveTileLds lds[kSpWaves];
    __shared__ uint32_t s_ticket;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    WaveTileLds &L = lds[wave];
    const uint32_t nrec = min(*reinterpret_cast<const uint32_t *>(p.dispatch_buffer), p.dispatch_capacity);
    const uint32_t ntiles = (nrec + kTileRecords - 1) / kTileRecords;
    const uint32_t nrows = (ntiles + kSpWaves - 1) / kSpWaves;
    uint32_t *ctl = p.ticket; // {ticket, done, epoch}
    const uint32_t epoch = ctl[2];
    for (;;) {
        __syncthreads(); // previous iteration has consumed s_ticket
        if (threadIdx.x == 0) s_ticket = atomicAdd(&ctl[0], 1u);
        __syncthreads();
        const uint32_t row = s_ticket;
        if (row >= nrows) {
            if (row == 0 && threadIdx.x == 0) *reinterpret_cast<uint32_t *>(p.draw_buffer) = 0u; // no records
            break;
        }
        const uint32_t w = row * kSpWaves + wave;
        if (w < ntiles) { // wave-uniform
            tile_setup(p, L, w, nrec, lane);
            uint64_t draw_mask[kTileRows];
            const uint32_t total = tile_rows<true>(p, L, lane, draw_mask);
            uint32_t base = 0;
            if (w > 0) {
                if (lane == 0)
                    __hip_atomic_store(p.scan_state + w, make_granule(epoch, ST_AGG, total), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                base = look_back(p, w, epoch, lane);
            }
            const uint32_t inclusive = base + total;
            if (lane == 0)
                __hip_atomic_store(p.scan_state + w, make_granule(epoch, ST_PREFIX, inclusive), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            tile_emit(p, L, lane, base, draw_mask);
            if (w == ntiles - 1 && lane == 0) {
                // `count` header: subsumes the fill_buffer(count = 0) of draw_gen.rs:411-417
                *reinterpret_cast<uint32_t *>(p.draw_buffer) = min(inclusive, p.draw_capacity);
                if (inclusive > p.draw_capacity) latch_status(p.status, ORBIT_E_CAPACITY);
            }
        }
    }
    // self-cleaning control words: the last block to leave resets the ticket and
    // bumps the epoch, so the next launch (stream-ordered) starts clean.
    if (threadIdx.x == 0) {
        const uint32_t done = atomicAdd(&ctl[1], 1u);
        if (done == gridDim.x - 1) {
            uint32_t next = (epoch + 1u) & 0x3FFFFFFFu;
            if (next == 0u) next = 1u; // epoch 0 is what a zero-initialised granule carries
            __hip_atomic_store(&ctl[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl[2], next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

} // namespace

hipError_t launch_meshlet_cull(const MeshletCullParams &p, uint32_t num_cus, int mode, hipStream_t s) {
    const uint32_t max_tiles = (p.dispatch_capacity + kTileRecords - 1) / kTileRecords;
    if (mode == 1) {
        // persistent blocks, one 1024-thread block per CU (<= 128 VGPRs), looping over tickets
        uint32_t blocks = num_cus;
        const uint32_t need = (max_tiles + kSpWaves - 1) / kSpWaves;
        if (blocks > need) blocks = need;
        if (blocks == 0) blocks = 1;
        hipLaunchKernelGGL(meshlet_cull_single_pass, dim3(blocks), dim3(kSpBlock), 0, s, p);
        return hipGetLastError();
    }
    // two-pass fallback (count -> scan -> emit): persistent-style grid, grid-stride over tiles
    uint32_t blocks = num_cus * 8u;
    const uint32_t need = (max_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > need) blocks = need;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(meshlet_cull_two_pass<MODE_COUNT>, dim3(blocks), dim3(kBlock), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = launch_scan_exclusive_indirect(p.tile_counts, p.tile_base,
                                       reinterpret_cast<const uint32_t *>(p.dispatch_buffer), kTileRecords, max_tiles,
                                       p.total, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(meshlet_cull_two_pass<MODE_EMIT>, dim3(blocks), dim3(kBlock), 0, s, p);
    return hipGetLastError();
}

} // namespace orbit
