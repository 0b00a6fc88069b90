// depth_reduce.hip — the whole HiZ depth pyramid in two launches on gfx950.
//
// Restates shaders/depth_reduce.comp:14-20 driven by DepthPyramid::update
// (src/passes/draw_gen.rs:510-566): every level is the ReduceMin-sampled
// (2x2 clamped footprint minimum, src/graphics/device.rs:1404-1420) image of
// the level below, level 0 of the screen depth buffer.  The reference issues
// one dispatch + barrier per level (<= 12); here one kernel produces levels
// 0..5 of each 32x32 level-0 tile through LDS, and one single-block kernel
// finishes the remaining (<= 64x64-texel) tail.  Several pyramids (the frame's
// main view and shadow cascades: update_multiple_depth_pyramids::<C>,
// draw_gen.rs:569-628) share the two launches: blockIdx.z selects the pyramid.
// Every level is addressed through its own base pointer and row pitch, so the
// chain may be one packed buffer or separate per-mip (linear-tiled) images.
//
// For the power-of-two pyramid levels the footprint at u = (X+0.5)/w_dst is
// exactly texels {2X, 2X+1} of the source (clamped when the source dimension
// is already 1); only level 0 (screen -> next_pow2/2, a non-integer ratio) uses
// the fractional footprint arithmetic.
#include "handoff.h"

namespace orbit {

namespace {

constexpr int kTile = 32;
constexpr int kInBlockLevels = 6; // levels 0..5 from one 32x32 tile

// does the tile launch finish this pyramid itself (its last workgroup reduces levels 6.. in LDS)?
__host__ __device__ inline bool tail_in_launch(const DepthReduceParams &p) {
    return p.mips > (uint32_t)kInBlockLevels && (p.w0 >> (kInBlockLevels - 1)) <= 64u && (p.h0 >> (kInBlockLevels - 1)) <= 64u;
}

__device__ __forceinline__ float reduce4(const float *src, uint32_t pitch, int ox, int oy, uint32_t sw, uint32_t sh,
                                         uint32_t X, uint32_t Y) {
    // source texels {2X, 2X+1} x {2Y, 2Y+1}, clamped to the source size, in
    // coordinates local to a tile whose origin at the source level is (ox, oy)
    const int x0 = (int)min(2u * X, sw - 1u) - ox, x1 = (int)min(2u * X + 1u, sw - 1u) - ox;
    const int y0 = (int)min(2u * Y, sh - 1u) - oy, y1 = (int)min(2u * Y + 1u, sh - 1u) - oy;
    const float a = src[y0 * pitch + x0], b = src[y0 * pitch + x1];
    const float c = src[y1 * pitch + x0], d = src[y1 * pitch + x1];
    return gmin(gmin(a, b), gmin(c, d));
}

// Levels kInBlockLevels.. : the workgroup that finishes LAST among a pyramid's tile workgroups does them, inside the
// same launch (rounds 1-3: a second, one-workgroup launch that went to memory and back for every level: 4.2 us + a
// launch boundary at 1080p).  Every workgroup stores its one texel of level kInBlockLevels - 1 write-through, drains it
// and takes a ticket; the last ticket's holder loads that level (at most kTailSide^2 texels) with sc1 loads into LDS and
// reduces the remaining levels there.  Larger pyramids (level 5 beyond 64 x 64: a level 0 beyond 2048^2) keep the second
// launch.  handoff.h has the rules this follows.
constexpr uint32_t kTailSide = 64;

__device__ __forceinline__ void depth_reduce_tail_in_lds(const DepthReduceParams &p, float *a, float *b) {
    const uint32_t k0 = kInBlockLevels - 1;
    const uint32_t w5 = max(p.w0 >> k0, 1u), h5 = max(p.h0 >> k0, 1u);
    const __amdgpu_buffer_rsrc_t L5 = rsrc_of(p.level[k0], (uint32_t)(((size_t)(h5 - 1u) * p.pitch[k0] + w5) * 4u));
    for (uint32_t i = threadIdx.x; i < w5 * h5; i += blockDim.x)
        a[i] = __uint_as_float(ld_sc1(L5, ((i / w5) * p.pitch[k0] + i % w5) * 4u));
    __syncthreads();
    float *src = a, *dst = b;
    uint32_t spitch = w5;
    for (uint32_t k = kInBlockLevels; k < p.mips; k++) {
        const uint32_t sw = max(p.w0 >> (k - 1), 1u), sh = max(p.h0 >> (k - 1), 1u);
        const uint32_t dw = max(p.w0 >> k, 1u), dh = max(p.h0 >> k, 1u);
        for (uint32_t i = threadIdx.x; i < dw * dh; i += blockDim.x) {
            const uint32_t X = i % dw, Y = i / dw;
            const float v = reduce4(src, spitch, 0, 0, sw, sh, X, Y);
            dst[Y * dw + X] = v;
            p.level[k][(size_t)Y * p.pitch[k] + X] = v;
        }
        __syncthreads();
        float *t = src;
        src = dst, dst = t;
        spitch = dw;
    }
}

// min of a value and its partner's (lane ^ mask) in reduce4's operand order: `first` = this lane holds the first operand
// (the lower coordinate); both lanes of the pair end up with the same value
__device__ __forceinline__ float pair_min(float v, int mask, bool first) {
    const float o = __shfl_xor(v, mask, 64);
    return first ? gmin(v, o) : gmin(o, v);
}

// The tail of a pyramid whose level kInBlockLevels - 1 is 32 x 32 and whose top is 1 x 1 five levels on (a 1024^2 level 0:
// the 1080p screen's) by ONE wave without LDS: lane (lx, ly) of an 8 x 8 grid loads its 4 x 4 block of that level (sc1:
// written through by the tile workgroups), reduces two levels in its own registers and three by lane exchanges.  The
// general form above takes five LDS round trips and barriers for the same levels (2.8 of the launch's 10.3 us).
__device__ __forceinline__ void depth_reduce_tail_one_wave(const DepthReduceParams &p, int lane) {
    const uint32_t k0 = kInBlockLevels - 1;
    const uint32_t lx = (uint32_t)lane & 7u, ly = (uint32_t)lane >> 3;
    const __amdgpu_buffer_rsrc_t L5 = rsrc_of(p.level[k0], (uint32_t)((31u * p.pitch[k0] + 32u) * 4u));
    float a[4][4];
#pragma unroll
    for (uint32_t j = 0; j < 4; j++)
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) a[j][i] = __uint_as_float(ld_sc1(L5, ((4u * ly + j) * p.pitch[k0] + 4u * lx + i) * 4u));
    float b[2][2]; // level k0 + 1: 16 x 16
#pragma unroll
    for (uint32_t j = 0; j < 2; j++)
#pragma unroll
        for (uint32_t i = 0; i < 2; i++) {
            b[j][i] = gmin(gmin(a[2 * j][2 * i], a[2 * j][2 * i + 1]), gmin(a[2 * j + 1][2 * i], a[2 * j + 1][2 * i + 1]));
            p.level[k0 + 1][(size_t)(2u * ly + j) * p.pitch[k0 + 1] + 2u * lx + i] = b[j][i];
        }
    const float c = gmin(gmin(b[0][0], b[0][1]), gmin(b[1][0], b[1][1])); // 8 x 8: one per lane
    p.level[k0 + 2][(size_t)ly * p.pitch[k0 + 2] + lx] = c;
    const float d = pair_min(pair_min(c, 1, (lx & 1u) == 0u), 8, (ly & 1u) == 0u); // 4 x 4
    if (((lx | ly) & 1u) == 0u) p.level[k0 + 3][(size_t)(ly >> 1) * p.pitch[k0 + 3] + (lx >> 1)] = d;
    const float e = pair_min(pair_min(d, 2, (lx & 2u) == 0u), 16, (ly & 2u) == 0u); // 2 x 2
    if (((lx | ly) & 3u) == 0u) p.level[k0 + 4][(size_t)(ly >> 2) * p.pitch[k0 + 4] + (lx >> 2)] = e;
    const float f = pair_min(pair_min(e, 4, (lx & 4u) == 0u), 32, (ly & 4u) == 0u); // 1 x 1
    if (lane == 0) p.level[k0 + 5][0] = f;
}

// (Staging a tile's source region — at most 66 x 66 texels of the depth buffer — in LDS first, rows loaded coalesced, was
// measured in round 4 and lost: 14.0 against 12.1 us for the launch at 1080p; the footprints' own loads hit the L1.)
__global__ __launch_bounds__(256) void depth_reduce_tiles(const DepthReduceBatch batch) {
    __shared__ float tail_lds[kTailSide * kTailSide]; // the in-launch tail's (one workgroup of the launch uses it)
    __shared__ float buf0[kTile * kTile];
    __shared__ float buf1[(kTile / 2) * (kTile / 2)];
    const DepthReduceParams &p = batch.p[blockIdx.z];
    const int t = threadIdx.x;
    const uint32_t ox0 = blockIdx.x * kTile, oy0 = blockIdx.y * kTile;
    if (ox0 >= p.w0 || oy0 >= p.h0) return; // the grid covers the largest pyramid of the batch (block-uniform)
#ifdef ORBIT_TRIAGE_STAMPS // tools/depth_stamps.py: the critical path of the launch in wall-clock stamps (100 MHz), written
    // into the unused words of the pyramid's arrival counters: [0] workgroup (0, 0)'s start; of the workgroup that
    // finishes last: [1] its start, [2] level 0 done (a whole tile: levels 0..5), [3] levels 1..5 done, [4] its stores
    // drained, [5] tickets taken, [6] tail done
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(batch.tickets + blockIdx.z * kDepthTicketWords + 2);
    const unsigned long long st_start = wall_clock64();
    unsigned long long st_l0 = 0, st_levels = 0, st_drained = 0;
    if (t == 0 && blockIdx.x == 0 && blockIdx.y == 0) stamps[0] = st_start;
#endif

    // a tile that lies whole inside every level it produces (level-0 sides multiples of 32, six levels or more — every
    // pyramid of a screen of 64 pixels a side or more): levels 1..3 by lane exchanges, 4 and 5 by one wave behind ONE
    // barrier; others go through LDS level by level with the clamps of the general form
    const bool whole = p.w0 % kTile == 0u && p.h0 % kTile == 0u && p.mips >= (uint32_t)kInBlockLevels; // (block-uniform)
    __shared__ float s_l3[16];
    // level 0: each thread produces a 2x2 quad of the tile.  All four footprints are fetched before the first texel is
    // stored: a store into the pyramid may alias the depth buffer for all the compiler knows, so with the store between
    // them every footprint's loads waited for the one before (four round trips in a row: 4.6 of the launch's 11 us).
    {
        float *dst = p.level[0];
        const int qx = t & 15, qy = t >> 4;
        float q[2][2];
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const uint32_t x = ox0 + 2 * qx + i, y = oy0 + 2 * qy + j;
                // (vec2(pos) + 0.5) / vec2(dimensions), depth_reduce.comp:18; a texel outside the level: its clamped
                // neighbour's footprint, fetched and dropped
                const float u = ((float)min(x, p.w0 - 1u) + 0.5f) / (float)p.w0;
                const float vv = ((float)min(y, p.h0 - 1u) + 0.5f) / (float)p.h0;
                q[j][i] = footprint_min(p.depth, p.depth_pitch, p.screen_w, p.screen_h, u, vv);
            }
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int lx = 2 * qx + i, ly = 2 * qy + j;
                const uint32_t x = ox0 + lx, y = oy0 + ly;
                const bool in = x < p.w0 && y < p.h0;
                if (in) dst[(size_t)y * p.pitch[0] + x] = q[j][i];
                if (!whole) buf0[ly * kTile + lx] = in ? q[j][i] : 0.0f;
            }
        if (whole) {
            static_assert(kTile == 32 && kInBlockLevels == 6, "lane = (qy & 3) * 16 + qx; levels 1..5 of a 32 x 32 tile");
            const uint32_t uqx = (uint32_t)qx, uqy = (uint32_t)qy;
            // reduce4's order: (x0, y0), (x1, y0) | (x0, y1), (x1, y1)
            const float v1 = gmin(gmin(q[0][0], q[0][1]), gmin(q[1][0], q[1][1]));
            p.level[1][(size_t)((oy0 >> 1) + uqy) * p.pitch[1] + (ox0 >> 1) + uqx] = v1;
            const float v2 = pair_min(pair_min(v1, 1, (uqx & 1u) == 0u), 16, (uqy & 1u) == 0u);
            if (((uqx | uqy) & 1u) == 0u) p.level[2][(size_t)((oy0 >> 2) + (uqy >> 1)) * p.pitch[2] + (ox0 >> 2) + (uqx >> 1)] = v2;
            const float v3 = pair_min(pair_min(v2, 2, (uqx & 2u) == 0u), 32, (uqy & 2u) == 0u);
            if (((uqx | uqy) & 3u) == 0u) {
                p.level[3][(size_t)((oy0 >> 3) + (uqy >> 2)) * p.pitch[3] + (ox0 >> 3) + (uqx >> 2)] = v3;
                s_l3[(uqy >> 2) * 4u + (uqx >> 2)] = v3; // (rows 4 w .. 4 w + 3 of quads are wave w's: the next level crosses waves)
            }
            __syncthreads();
            if (t < 64) { // (wave 0; lanes 16.. repeat lanes 0..15)
                const uint32_t x = (uint32_t)t & 3u, y = ((uint32_t)t >> 2) & 3u;
                const float w3 = s_l3[y * 4u + x];
                const float v4 = pair_min(pair_min(w3, 1, (x & 1u) == 0u), 4, (y & 1u) == 0u);
                if (t < 16 && ((x | y) & 1u) == 0u) p.level[4][(size_t)((oy0 >> 4) + (y >> 1)) * p.pitch[4] + (ox0 >> 4) + (x >> 1)] = v4;
                const float v5 = pair_min(pair_min(v4, 2, (x & 2u) == 0u), 8, (y & 2u) == 0u);
                // the level the in-launch tail reads from other workgroups' hands: write-through
                if (t == 0 && p.mips > 5u)
                    __hip_atomic_store(reinterpret_cast<uint32_t *>(p.level[5] + ((size_t)(oy0 >> 5) * p.pitch[5] + (ox0 >> 5))),
                                       __float_as_uint(v5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (!whole) __syncthreads();
#ifdef ORBIT_TRIAGE_STAMPS
    st_l0 = wall_clock64();
#endif

    // levels 1..5 inside the block, ping-ponging between the two LDS buffers
    const int last = whole ? 0 : min((int)p.mips, kInBlockLevels);
    for (int k = 1; k < last; k++) {
        const float *src = (k & 1) ? buf0 : buf1;
        float *dstl = (k & 1) ? buf1 : buf0;
        const int tw = kTile >> k;                       // tile width at level k
        const uint32_t spitch = (uint32_t)(kTile >> (k - 1));
        const uint32_t sw = max(p.w0 >> (k - 1), 1u), sh = max(p.h0 >> (k - 1), 1u);
        const uint32_t dw = max(p.w0 >> k, 1u), dh = max(p.h0 >> k, 1u);
        const uint32_t oxs = ox0 >> (k - 1), oys = oy0 >> (k - 1);
        const uint32_t oxd = ox0 >> k, oyd = oy0 >> k;
        if (t < tw * tw) {
            const int lx = t % tw, ly = t / tw;
            const uint32_t X = oxd + lx, Y = oyd + ly;
            float v = 0.0f;
            if (X < dw && Y < dh) {
                v = reduce4(src, spitch, (int)oxs, (int)oys, sw, sh, X, Y);
                float *out = p.level[k] + ((size_t)Y * p.pitch[k] + X);
                // the level the in-launch tail reads from other workgroups' hands: write-through
                if (k == kInBlockLevels - 1) __hip_atomic_store(reinterpret_cast<uint32_t *>(out), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *out = v;
            }
            dstl[ly * tw + lx] = v;
        }
        __syncthreads();
    }
    if (!tail_in_launch(p)) return; // (block-uniform)
    // the pyramid's last workgroup out does levels 6.. (the buffers above are free again: reused as its LDS)
    __shared__ uint32_t s_last;
#ifdef ORBIT_TRIAGE_STAMPS
    st_levels = wall_clock64();
#endif
    drain_stores();
    __syncthreads();
#ifdef ORBIT_TRIAGE_STAMPS
    st_drained = wall_clock64();
#endif
    if (t == 0) {
        // two levels of arrival counters — a row of tile workgroups, then the rows: a thousand returning atomics on ONE
        // word queue for 12 us on this chip (~88 per us and word); 32 + 32 on 33 words do not
        const uint32_t tx = (p.w0 + kTile - 1) / kTile, ty = (p.h0 + kTile - 1) / kTile;
        uint32_t *base = batch.tickets + blockIdx.z * kDepthTicketWords;
        uint32_t *row = base + (1u + blockIdx.y) * 16u;
        uint32_t last = 0u;
        if (ticket(row) == tx - 1u) {
            __hip_atomic_store(row, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // next launch
            if (ticket(base) == ty - 1u) {
                __hip_atomic_store(base, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1u;
            }
        }
        s_last = last;
    }
    __syncthreads();
    if (uniform(s_last) == 0u) return;
#ifdef ORBIT_TRIAGE_STAMPS
    const unsigned long long st_tickets = wall_clock64();
#endif
    static_assert((kTailSide / 2) * (kTailSide / 2) <= kTile * kTile, "tail buffers");
    if ((p.w0 >> (kInBlockLevels - 1)) == 32u && (p.h0 >> (kInBlockLevels - 1)) == 32u && p.mips == (uint32_t)kInBlockLevels + 5u) {
        if (t < 64) depth_reduce_tail_one_wave(p, t);
    } else {
        depth_reduce_tail_in_lds(p, tail_lds, buf0);
    }
#ifdef ORBIT_TRIAGE_STAMPS
    if (t == 0) {
        drain_stores();
        stamps[1] = st_start, stamps[2] = st_l0, stamps[3] = st_levels, stamps[4] = st_drained, stamps[5] = st_tickets;
        stamps[6] = wall_clock64();
    }
#endif
}

// Levels kInBlockLevels.. from global memory, one block (pyramids too large for the in-launch tail).
__global__ __launch_bounds__(1024) void depth_reduce_tail(const DepthReduceBatch batch) {
    const DepthReduceParams &p = batch.p[blockIdx.x];
    if (tail_in_launch(p)) return;
    for (uint32_t k = kInBlockLevels; k < p.mips; k++) {
        const float *src = p.level[k - 1];
        float *dst = p.level[k];
        const uint32_t sw = max(p.w0 >> (k - 1), 1u), sh = max(p.h0 >> (k - 1), 1u);
        const uint32_t dw = max(p.w0 >> k, 1u), dh = max(p.h0 >> k, 1u);
        for (uint32_t i = threadIdx.x; i < dw * dh; i += blockDim.x) {
            const uint32_t X = i % dw, Y = i / dw;
            dst[(size_t)Y * p.pitch[k] + X] = reduce4(src, p.pitch[k - 1], 0, 0, sw, sh, X, Y);
        }
        __syncthreads(); // workgroup-scope release/acquire: next level reads these stores
    }
}

} // namespace

hipError_t launch_depth_reduce(const DepthReduceBatch &b, hipStream_t s) {
    if (b.count == 0 || b.count > kMaxPyramidBatch) return hipErrorInvalidValue;
    uint32_t gx = 0, gy = 0;
    bool second = false;
    for (uint32_t i = 0; i < b.count; i++) {
        gx = max(gx, (b.p[i].w0 + kTile - 1) / kTile);
        gy = max(gy, (b.p[i].h0 + kTile - 1) / kTile);
        second = second || (b.p[i].mips > (uint32_t)kInBlockLevels && !tail_in_launch(b.p[i]));
    }
    hipLaunchKernelGGL(depth_reduce_tiles, dim3(gx, gy, b.count), dim3(256), 0, s, b);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (second) {
        hipLaunchKernelGGL(depth_reduce_tail, dim3(b.count), dim3(1024), 0, s, b);
        e = hipGetLastError();
    }
    return e;
}

} // namespace orbit
