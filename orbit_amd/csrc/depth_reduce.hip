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
#include "kernels.h"

namespace orbit {

namespace {

constexpr int kTile = 32;
constexpr int kInBlockLevels = 6; // levels 0..5 from one 32x32 tile

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

__global__ __launch_bounds__(256) void depth_reduce_tiles(const DepthReduceBatch batch) {
    __shared__ float buf0[kTile * kTile];
    __shared__ float buf1[(kTile / 2) * (kTile / 2)];
    const DepthReduceParams &p = batch.p[blockIdx.z];
    const int t = threadIdx.x;
    const uint32_t ox0 = blockIdx.x * kTile, oy0 = blockIdx.y * kTile;
    if (ox0 >= p.w0 || oy0 >= p.h0) return; // the grid covers the largest pyramid of the batch (block-uniform)

    // level 0: each thread produces a 2x2 quad of the tile
    {
        float *dst = p.level[0];
        const int qx = t & 15, qy = t >> 4;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int lx = 2 * qx + i, ly = 2 * qy + j;
                const uint32_t x = ox0 + lx, y = oy0 + ly;
                float v = 0.0f;
                if (x < p.w0 && y < p.h0) {
                    // (vec2(pos) + 0.5) / vec2(dimensions), depth_reduce.comp:18
                    const float u = ((float)x + 0.5f) / (float)p.w0;
                    const float vv = ((float)y + 0.5f) / (float)p.h0;
                    v = footprint_min(p.depth, p.depth_pitch, p.screen_w, p.screen_h, u, vv);
                    dst[(size_t)y * p.pitch[0] + x] = v;
                }
                buf0[ly * kTile + lx] = v;
            }
    }
    __syncthreads();

    // levels 1..5 inside the block, ping-ponging between the two LDS buffers
    const int last = min((int)p.mips, kInBlockLevels);
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
                p.level[k][(size_t)Y * p.pitch[k] + X] = v;
            }
            dstl[ly * tw + lx] = v;
        }
        __syncthreads();
    }
}

// Levels kInBlockLevels.. from global memory, one block.
__global__ __launch_bounds__(1024) void depth_reduce_tail(const DepthReduceBatch batch) {
    const DepthReduceParams &p = batch.p[blockIdx.x];
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
    uint32_t gx = 0, gy = 0, max_mips = 0;
    for (uint32_t i = 0; i < b.count; i++) {
        gx = max(gx, (b.p[i].w0 + kTile - 1) / kTile);
        gy = max(gy, (b.p[i].h0 + kTile - 1) / kTile);
        max_mips = max(max_mips, b.p[i].mips);
    }
    hipLaunchKernelGGL(depth_reduce_tiles, dim3(gx, gy, b.count), dim3(256), 0, s, b);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (max_mips > (uint32_t)kInBlockLevels) { // pyramids without a tail fall through their (empty) level loop
        hipLaunchKernelGGL(depth_reduce_tail, dim3(b.count), dim3(1024), 0, s, b);
        e = hipGetLastError();
    }
    return e;
}

} // namespace orbit
