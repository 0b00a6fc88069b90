// orbit_device.h — device-side scalar semantics shared by all gfx950 kernels.
//
// Everything that decides a visibility bit is written here once, in the
// canonical arithmetic of DESIGN.md §"Canonical arithmetic": IEEE binary32,
// round-to-nearest, NO contraction (the translation units are built with
// -ffp-contract=off), fixed left-to-right association, GLSL min/max/clamp
// semantics, correctly rounded '/' and sqrtf (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt), software log2.
//
// Reference arithmetic restated: shaders/entity_cull.comp:28-102,
// shaders/meshlet_cull.comp:28-106 (paths relative to the reference repo).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orbit_abi.h"

#define ORBIT_WAVE 64

namespace orbit {

// GLSL: max(x,y) = y if x < y else x ; min(x,y) = y if y < x else x.
__device__ __forceinline__ float gmax(float x, float y) { return (x < y) ? y : x; }
__device__ __forceinline__ float gmin(float x, float y) { return (y < x) ? y : x; }
__device__ __forceinline__ float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }

// uint(float): saturating, NaN -> 0.
__device__ __forceinline__ uint32_t f2u_sat(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
// float texel index -> int in [0, hi]; NaN -> 0.
__device__ __forceinline__ int f2i_clamp(float f, int hi) {
    if (!(f >= 0.0f)) return 0;
    if (f > (float)hi) return hi;
    return (int)f;
}
// 1u << s, 0 for s >= 32 (gfx950 would wrap the shift count: mask explicitly).
__device__ __forceinline__ uint32_t shl1(uint32_t s) { return s < 32u ? (1u << s) : 0u; }

// The one driver-defined point of the cull shaders that OrbitCaps.arith_profile selects: what becomes of
// OpDot / OpMatrixTimesVector / OpMatrixTimesMatrix / Length / Distance.  madd(a, b, acc) is one term of such a sum:
// canonical (ORBIT_CONTRACT 0, the product's default build) a rounded product added to the accumulator — the expression
// trees below are then exactly the left-to-right sums of rounds 1-5 —, contracted (ORBIT_CONTRACT 1: the translation units
// that evaluate geometry are compiled a second time, *_contracted.hip) one fma, which is how oracle/spirv_vm.py's
// `fused_dot` runs the reference's binaries for tests/golden/spirv_cull_contracted.npz.  Everything the binaries spell out
// themselves (FMul, FAdd, Fma, FDiv) is written out below and is the same in both builds.
#ifndef ORBIT_CONTRACT
#define ORBIT_CONTRACT 0
#endif
__device__ __forceinline__ float madd(float a, float b, float acc) {
#if ORBIT_CONTRACT
    return __builtin_fmaf(a, b, acc);
#else
    return acc + a * b;
#endif
}
__device__ __forceinline__ float dot2(float ax, float ay, float bx, float by) { return madd(ay, by, ax * bx); }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return madd(az, bz, madd(ay, by, ax * bx));
}

// Canonical software log2: same specification as oracle/orbit_oracle.c
// orbit_log2f (exponent + degree-9 Horner polynomial, mul/add only).
__device__ __forceinline__ float log2c(float x) {
    uint32_t b = __float_as_uint(x);
    if ((b & 0x7FFFFFFFu) > 0x7F800000u) return x;
    if ((b & 0x7FFFFFFFu) == 0u) return -__builtin_inff();
    if (b & 0x80000000u) return __uint_as_float(0x7FC00000u);
    if (b == 0x7F800000u) return x;
    int e = 0;
    if (b < 0x00800000u) {
        x = x * 16777216.0f;
        b = __float_as_uint(x);
        e = -24;
    }
    e += (int)(b >> 23) - 127;
    float m = __uint_as_float((b & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421354f) {
        m = m * 0.5f;
        e += 1;
    }
    const float f = m - 1.0f;
    float p = -0x1.cebep-4f;
    p = p * f; p = p + 0x1.80ab18p-3f;
    p = p * f; p = p + -0x1.865ffcp-3f;
    p = p * f; p = p + 0x1.a265fcp-3f;
    p = p * f; p = p + -0x1.eac694p-3f;
    p = p * f; p = p + 0x1.2782e6p-2f;
    p = p * f; p = p + -0x1.715a68p-2f;
    p = p * f; p = p + 0x1.ec708p-2f;
    p = p * f; p = p + -0x1.71547p-1f;
    p = p * f; p = p + 0x1.715476p+0f;
    const float r = p * f;
    return r + (float)e;
}

// cluster_common.glsl:18-20 as compiled (mark_active.comp.spv: one fused operation): the canonical z slice of a depth
// sample, uint(fma(log2c(linear_z), z_scale, z_bias)).
__device__ __forceinline__ uint32_t depth_slice_canonical(float linear_z, float z_scale, float z_bias) {
    return f2u_sat(__builtin_fmaf(log2c(linear_z), z_scale, z_bias));
}
// The same integer from the HARDWARE log2 (v_log_f32, one instruction instead of log2c's thirty) wherever that cannot
// differ.  v_log_f32 and log2c both approximate log2: over every positive normal float they lie within 2^-22 (1 + |log2 x|)
// of each other (measured, not assumed: orbit_debug_log2_guard runs all of them; the test holds the library to it), so the
// two fused results t = fma(L, z_scale, z_bias) differ by at most |z_scale| 2^-22 (1 + |L|) plus one rounding of t — and a
// t further than `guard` (twice that) from every integer truncates to the same slice either way.  Inside the band (about one
// sample in ten thousand at the renderer's grids), for denormal / non-finite / negative inputs and for huge t the canonical
// form decides: the result is the canonical one BY CONSTRUCTION, the band only has to be wide enough.
__device__ __forceinline__ bool depth_slice_safe(float linear_z, float L, float t, float z_scale) {
    const float guard = __builtin_fmaf(__builtin_fabsf(z_scale) * (1.0f + __builtin_fabsf(L)), 0x1p-21f,
                                       __builtin_fmaf(__builtin_fabsf(t), 0x1p-22f, 0x1p-21f));
    const float off = __builtin_fabsf(t - __builtin_rintf(t));
    // (every comparison is false for a NaN: NaN takes the canonical path)
    return off > guard && __builtin_fabsf(t) < 0x1p20f && linear_z >= 0x1p-126f && linear_z < __builtin_inff();
}
__device__ __forceinline__ uint32_t depth_slice(float linear_z, float z_scale, float z_bias) {
    const float L = __builtin_amdgcn_logf(linear_z);
    const float t = __builtin_fmaf(L, z_scale, z_bias);
    const bool safe = depth_slice_safe(linear_z, L, t, z_scale);
    uint32_t slice = f2u_sat(t);
    if (__ballot(!safe) != 0ull) { // wave-uniform: a wave without a lane in the band never runs the polynomial
        const uint32_t canonical = depth_slice_canonical(linear_z, z_scale, z_bias);
        slice = safe ? slice : canonical;
    }
    return slice;
}

// int(i8) / 127.0 (meshlet_cull.comp:121-122) AS COMPILED: the reference's shader binaries multiply by the rounded
// reciprocal (meshlet_cull.comp.spv, OpFMul by 0x3C010204 — spirv-opt's reciprocal folding); 16 of the 256 inputs
// differ from the IEEE quotient by one ulp, and the oracle follows the binary (oracle/orbit_oracle.c snorm8).
__device__ __forceinline__ float snorm8(int i) { return (float)i * __uint_as_float(0x3C010204u); }

// Pyramid geometry handed to the kernels by value (host-computed).  `texels` is the packed R32F mip chain; when
// `levels` is non-null the chain is NOT packed: level k lives at levels[k].texels with levels[k].row_pitch texels per
// row (a DEVICE table of `mips` entries: separate per-mip images, orbit_abi.h OrbitDepthPyramidLevel).
struct PyramidView {
    const float *texels;
    uint32_t w0, h0, mips;
    const OrbitDepthPyramidLevel *levels;
};

// 2x2 clamped footprint minimum of one w x h level (ReduceMin sampler model,
// src/graphics/device.rs:1404-1420).
__device__ __forceinline__ float footprint_min(const float *img, uint32_t pitch, uint32_t w, uint32_t h, float u,
                                               float v) {
    const float x = u * (float)w - 0.5f;
    const float y = v * (float)h - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const int x0 = f2i_clamp(fx, (int)w - 1), x1 = f2i_clamp(fx + 1.0f, (int)w - 1);
    const int y0 = f2i_clamp(fy, (int)h - 1), y1 = f2i_clamp(fy + 1.0f, (int)h - 1);
    const float a = img[(size_t)y0 * pitch + x0], b = img[(size_t)y0 * pitch + x1];
    const float c = img[(size_t)y1 * pitch + x0], d = img[(size_t)y1 * pitch + x1];
    return gmin(gmin(a, b), gmin(c, d));
}

// textureLod(pyramid, ReduceMin, uv, lod), NEAREST mip mode:
// level = clamp(ceil(lod + 0.5) - 1, 0, mips - 1).
__device__ __forceinline__ float hiz_sample(const PyramidView &p, float u, float v, float lod) {
    const float lf = ceilf(lod + 0.5f) - 1.0f;
    const int level = f2i_clamp(lf, (int)p.mips - 1);
    // texel offset of `level` in the packed mip chain (level k is
    // max(dim >> k, 1), src/graphics/image.rs:533); summed here instead of
    // indexed from a kernarg table, which would need a per-lane waterfall.
    const uint32_t w = max(p.w0 >> level, 1u), h = max(p.h0 >> level, 1u);
    if (p.levels) { // separate per-mip images: one 16-B table entry per lane (cached), then the pitched footprint
        const OrbitDepthPyramidLevel lv = p.levels[level];
        return footprint_min(lv.texels, lv.row_pitch, w, h, u, v);
    }
    uint32_t off = 0;
    for (int k = 0; k < level; k++) off += max(p.w0 >> k, 1u) * max(p.h0 >> k, 1u);
    return footprint_min(p.texels + off, w, w, h, u, v);
}

struct Sphere {
    float x, y, z, r;
};

// project_sphere, entity_cull.comp:83-102 / meshlet_cull.comp:83-102, as the compiled shaders hold it (the fused
// operations are spirv-opt's: oracle/orbit_oracle.c project_sphere).
__device__ __forceinline__ void project_sphere(const Sphere &s, float p00, float p11, float aabb[4]) {
    const float r = s.r;
    const float cxx = -s.x, cxy = -s.z;
    const float vxx = sqrtf(__builtin_fmaf(-r, r, dot2(cxx, cxy, cxx, cxy))), vxy = r;
    // (mat2(vx.x, vx.y, -vx.y, vx.x) * cx and its mirror: OpMatrixTimesVector in the binaries)
    const float minx_x = madd(-vxy, cxy, vxx * cxx), minx_y = madd(vxx, cxy, vxy * cxx);
    const float maxx_x = madd(vxy, cxy, vxx * cxx), maxx_y = madd(vxx, cxy, (-vxy) * cxx);
    const float cyx = -s.y, cyy = -s.z;
    const float vyx = sqrtf(__builtin_fmaf(-r, r, dot2(cyx, cyy, cyx, cyy))), vyy = r;
    const float miny_x = madd(-vyy, cyy, vyx * cyx), miny_y = madd(vyx, cyy, vyy * cyx);
    const float maxy_x = madd(vyy, cyy, vyx * cyx), maxy_y = madd(vyx, cyy, (-vyy) * cyx);
    const float a0 = minx_x / minx_y * p00, a1 = miny_x / miny_y * p11;
    const float a2 = maxx_x / maxx_y * p00, a3 = maxy_x / maxy_y * p11;
    aabb[0] = __builtin_fmaf(a0, 0.5f, 0.5f);
    aabb[1] = __builtin_fmaf(a3, -0.5f, 0.5f);
    aabb[2] = __builtin_fmaf(a2, 0.5f, 0.5f);
    aabb[3] = __builtin_fmaf(a1, -0.5f, 0.5f);
}

// Pass-2 HiZ test, entity_cull.comp:146-191 == meshlet_cull.comp:160-205, operation by operation as compiled
// (oracle/orbit_oracle.c occlusion_test): `radius` is the model-space radius and `scale` the matrix's largest scale —
// s.r = radius * scale — because three sums of the test are fused with that product in the reference's binaries.
// Flips s.z in the perspective branch; the flip persists for the caller.
__device__ __forceinline__ bool occlusion_test(const OrbitGpuCullInfo &ci, Sphere &s, float radius, float scale,
                                               const PyramidView &pyr) {
    bool cullable = true;
    float aabb[4], closest;
    if (ci.projection_type == 0) {
        s.z = -s.z;
        cullable = s.z >= __builtin_fmaf(radius, scale, ci.z_near);
        project_sphere(s, ci.p00_or_width_recipx2, ci.p11_or_height_recipx2, aabb);
        closest = ci.z_near / __builtin_fmaf(-radius, scale, s.z);
    } else {
        const float sr = ci.p00_or_width_recipx2;
        const float cx = s.x * sr, cy = s.y * sr;
        const float bs = sr * s.r;
        const float b0 = __builtin_fmaf(bs, -1.0f, cx), b1 = __builtin_fmaf(bs, -1.0f, cy);
        const float b2 = __builtin_fmaf(bs, 1.0f, cx), b3 = __builtin_fmaf(bs, 1.0f, cy);
        aabb[0] = __builtin_fmaf(gclamp(b0, -1.0f, 1.0f), 0.5f, 0.5f);
        aabb[1] = __builtin_fmaf(gclamp(b1, -1.0f, 1.0f), -0.5f, 0.5f);
        aabb[2] = __builtin_fmaf(gclamp(b2, -1.0f, 1.0f), 0.5f, 0.5f);
        aabb[3] = __builtin_fmaf(gclamp(b3, -1.0f, 1.0f), -0.5f, 0.5f);
        const float closest_z = __builtin_fmaf(radius, scale, s.z);
        const float r = 1.0f / (ci.z_far - ci.z_near);
        closest = r * (closest_z + ci.z_far);
    }
    if (!cullable) return true;
    const float width = (aabb[2] - aabb[0]) * (float)pyr.w0;
    const float height = (aabb[3] - aabb[1]) * (float)pyr.h0;
    const float u = (aabb[0] + aabb[2]) * 0.5f, v = (aabb[1] + aabb[3]) * 0.5f;
    const float sampled = hiz_sample(pyr, u, v, log2c(gmax(width, height)));
    return closest >= sampled;
}

// Plane loop, entity_cull.comp:137-144 == meshlet_cull.comp:139-146.  The
// CullInfo lives in the kernarg segment, so plane loads are scalar loads.
__device__ __forceinline__ bool plane_test(const OrbitGpuCullInfo &ci, const Sphere &s) {
    bool visible = true;
    const uint32_t n = ci.cull_plane_count;
    for (uint32_t i = 0; i < n; i++) {
        const float d = dot3(ci.cull_planes[i][0], ci.cull_planes[i][1], ci.cull_planes[i][2], s.x, s.y, s.z) +
                        ci.cull_planes[i][3];
        visible = visible && (d > -s.r);
    }
    return visible;
}

// One column of (A * B): A column-major in registers/SGPRs, b = column of B.
__device__ __forceinline__ void mat4_mul_col(const float *a, float b0, float b1, float b2, float b3, float out[4]) {
#pragma unroll
    for (int r = 0; r < 4; r++) out[r] = madd(a[12 + r], b3, madd(a[8 + r], b2, madd(a[4 + r], b1, a[0 + r] * b0)));
}

// Latched device status word (orbit_ctx_status): first error wins.
__device__ __forceinline__ void latch_status(int32_t *status, int32_t code) {
    atomicCAS((int *)status, 0, code);
}

} // namespace orbit
