/*
 * orbit_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A sequential, plain-C restatement of the reference's cull / HiZ / light
 * cluster path (Thefefe/orbit), used exclusively by tests/, by
 * __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg as the checker
 * of the HIP engine.  Nothing under orbit_amd/ may import, link or call it.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference holds no golden vector,
 * known-answer test or fixture for this path and cannot be
 * built or run here (Rust + Vulkan; no cargo, no ICD).  The oracle is pinned
 * instead by (1) an independent numpy restatement of the predicates
 * (tests/np_restatement.py), (2) property tests, (3) golden fixtures generated
 * by this file (tests/golden/, script tests/golden/make_golden.py), (4) the
 * reference's own CPU twins of two pieces of the path (math.rs:170-199,
 * cluster.rs:150-184) restated in orbit_amd/host and compared with this file
 * (tests/test_host_cpu.py).
 *
 * Canonical execution model: sequential schedule in
 * ascending invocation order, subgroup = dispatch size S = 32, IEEE-754 binary32
 * round-to-nearest, no FMA contraction (-ffp-contract=off), the non-subgroup
 * `#else` visibility addressing (word = base + lane/32, bit = lane%32), and
 * log2 replaced by the software orbit_log2f below at all three decision points.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/orbit_abi.h"

#define S_DISPATCH 32u

/* ---- GLSL scalar semantics ------------------------------------------------ */

/* GLSL max(x,y) = y if x < y else x; min(x,y) = y if y < x else x. */
static inline float gmax(float x, float y) { return (x < y) ? y : x; }
static inline float gmin(float x, float y) { return (y < x) ? y : x; }
static inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
static inline uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* uint(float): saturating, NaN -> 0. */
static inline uint32_t f2u_sat(float f) {
    if (!(f > 0.0f)) return 0u; /* negative, -0, NaN */
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
/* float index -> int clamped to [0, hi]; NaN -> 0. */
static inline int32_t f2i_clamp(float f, int32_t hi) {
    if (!(f >= 0.0f)) return 0;
    if (f > (float)hi) return hi;
    return (int32_t)f;
}
/* 1u << s with s >= 32 yielding 0 (quirk 4). */
static inline uint32_t shl1(uint32_t s) { return s < 32u ? (1u << s) : 0u; }

/* dot products: fixed left-to-right association. */
static inline float dot2(float ax, float ay, float bx, float by) { return ax * bx + ay * by; }
static inline float dot3(const float *a, const float *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

/* GLSL mat4 * vec4, column-major m[c*4+r]: col0*v.x + col1*v.y + col2*v.z + col3*v.w */
static inline void mat4_mul_vec4(const float *m, const float *v, float *out) {
    for (int r = 0; r < 4; r++)
        out[r] = ((m[0 + r] * v[0] + m[4 + r] * v[1]) + m[8 + r] * v[2]) + m[12 + r] * v[3];
}
/* GLSL mat4 * mat4: column j of the product = A * column j of B. */
static inline void mat4_mul(const float *a, const float *b, float *out) {
    for (int c = 0; c < 4; c++) mat4_mul_vec4(a, b + 4 * c, out + 4 * c);
}

/* Canonical software log2 (shared specification with the HIP kernels; see
 * DESIGN.md "log2").  Exponent extraction + degree-9 polynomial in f = m - 1,
 * m in [sqrt(1/2), sqrt(2)); mul and add only, Horner order, exact at powers
 * of two.  Replaces GLSL log2 at entity_cull.comp:187,206, meshlet_cull.comp:201
 * and cluster_common.glsl:19, whose precision is implementation-defined. */
float orbit_log2f(float x) {
    static const float C[10] = {
        0x1.715476p+0f,  -0x1.71547p-1f,  0x1.ec708p-2f,  -0x1.715a68p-2f, 0x1.2782e6p-2f,
        -0x1.eac694p-3f, 0x1.a265fcp-3f, -0x1.865ffcp-3f, 0x1.80ab18p-3f,  -0x1.cebep-4f};
    uint32_t b = f2bits(x);
    if ((b & 0x7FFFFFFFu) > 0x7F800000u) return x;            /* NaN */
    if ((b & 0x7FFFFFFFu) == 0u) return -INFINITY;            /* +-0 */
    if (b & 0x80000000u) return bits2f(0x7FC00000u);          /* negative -> NaN */
    if (b == 0x7F800000u) return x;                           /* +inf */
    int32_t e = 0;
    if (b < 0x00800000u) { /* subnormal: scale by 2^24 (exact) */
        x = x * 16777216.0f;
        b = f2bits(x);
        e = -24;
    }
    e += (int32_t)(b >> 23) - 127;
    float m = bits2f((b & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421354f) {
        m = m * 0.5f;
        e += 1;
    }
    float f = m - 1.0f;
    float p = C[9];
    for (int i = 8; i >= 0; i--) {
        p = p * f;
        p = p + C[i];
    }
    float r = p * f;
    return r + (float)e;
}

/* ---- host-side parameter derivation ---------------------------------------- */

/* math::mip_levels_from_size, src/math.rs:18-20 (inputs are powers of two). */
uint32_t oracle_mip_levels_from_size(uint32_t max_size) {
    uint32_t l = 0;
    while ((max_size >> (l + 1)) != 0) l++;
    return l + 1 > 1 ? l + 1 : 1;
}

static uint32_t next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

/* DepthPyramid::new, src/passes/draw_gen.rs:457-459; level sizes image.rs:533. */
void oracle_depth_pyramid_desc(uint32_t sw, uint32_t sh, OrbitDepthPyramidDesc *d) {
    memset(d, 0, sizeof(*d));
    d->width = next_pow2(sw) / 2;
    d->height = next_pow2(sh) / 2;
    if (d->width == 0) d->width = 1; /* vulkan image extents are >= 1 */
    if (d->height == 0) d->height = 1;
    d->mip_levels = oracle_mip_levels_from_size(umax(d->width, d->height));
    uint32_t off = 0;
    for (uint32_t k = 0; k < d->mip_levels; k++) {
        d->mip_offset[k] = off;
        d->mip_width[k] = umax(d->width >> k, 1);
        d->mip_height[k] = umax(d->height >> k, 1);
        off += d->mip_width[k] * d->mip_height[k];
    }
    d->total_texels = off;
}
void oracle_depth_pyramid_desc_from_mip0(uint32_t w0, uint32_t h0, OrbitDepthPyramidDesc *d) {
    memset(d, 0, sizeof(*d));
    d->width = w0;
    d->height = h0;
    d->mip_levels = oracle_mip_levels_from_size(umax(w0, h0));
    uint32_t off = 0;
    for (uint32_t k = 0; k < d->mip_levels; k++) {
        d->mip_offset[k] = off;
        d->mip_width[k] = umax(w0 >> k, 1);
        d->mip_height[k] = umax(h0 >> k, 1);
        off += d->mip_width[k] * d->mip_height[k];
    }
    d->total_texels = off;
}

/* math::frustum_planes_from_matrix + normalize_plane, src/math.rs:71-89.
 * `m` column-major; planes[i] = row3 +- row_{i/2}, divided by |xyz|. */
void oracle_frustum_planes_from_matrix(const float *m, float planes[6][4]) {
    for (int i = 0; i < 6; i++) {
        int row = i / 2;
        for (int c = 0; c < 4; c++) {
            float r3 = m[c * 4 + 3], rr = m[c * 4 + row];
            planes[i][c] = (i & 1) ? (r3 - rr) : (r3 + rr);
        }
        float len = sqrtf(dot3(planes[i], planes[i]));
        for (int c = 0; c < 4; c++) planes[i][c] = planes[i][c] / len;
    }
}

/* ---- sampler model ---------------------------------------------------------- */

/* 2x2 clamped bilinear footprint minimum: VK_SAMPLER_REDUCTION_MODE_MIN,
 * LINEAR min/mag, CLAMP_TO_EDGE (src/graphics/device.rs:1404-1420); SURVEY.md
 * appendix A6.  `img` is w x h row-major. */
static float footprint_min(const float *img, uint32_t w, uint32_t h, float u, float v) {
    float x = u * (float)w - 0.5f;
    float y = v * (float)h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    int32_t x0 = f2i_clamp(fx, (int32_t)w - 1), x1 = f2i_clamp(fx + 1.0f, (int32_t)w - 1);
    int32_t y0 = f2i_clamp(fy, (int32_t)h - 1), y1 = f2i_clamp(fy + 1.0f, (int32_t)h - 1);
    float a = img[(size_t)y0 * w + x0], b = img[(size_t)y0 * w + x1];
    float c = img[(size_t)y1 * w + x0], d = img[(size_t)y1 * w + x1];
    return gmin(gmin(a, b), gmin(c, d));
}

/* textureLod(pyramid, ReduceMin, uv, lod) with NEAREST mip mode: level =
 * clamp(ceil(lod + 0.5) - 1, 0, mips-1) (Vulkan's preferred rounding). */
float oracle_hiz_sample(const float *pyr, uint32_t w0, uint32_t h0, float u, float v, float lod) {
    OrbitDepthPyramidDesc d;
    oracle_depth_pyramid_desc_from_mip0(w0, h0, &d);
    float lf = ceilf(lod + 0.5f) - 1.0f;
    int32_t level = f2i_clamp(lf, (int32_t)d.mip_levels - 1);
    return footprint_min(pyr + d.mip_offset[level], d.mip_width[level], d.mip_height[level], u, v);
}

/* shaders/depth_reduce.comp:14-20 driven by DepthPyramid::update,
 * src/passes/draw_gen.rs:510-566: one min-reduction per mip, mip 0 from the
 * screen depth buffer, mip k from mip k-1. */
void oracle_depth_reduce(const float *depth, uint32_t sw, uint32_t sh, float *pyr) {
    OrbitDepthPyramidDesc d;
    oracle_depth_pyramid_desc(sw, sh, &d);
    for (uint32_t k = 0; k < d.mip_levels; k++) {
        const float *src = k == 0 ? depth : pyr + d.mip_offset[k - 1];
        uint32_t srcw = k == 0 ? sw : d.mip_width[k - 1], srch = k == 0 ? sh : d.mip_height[k - 1];
        uint32_t dw = d.mip_width[k], dh = d.mip_height[k];
        float *dst = pyr + d.mip_offset[k];
        for (uint32_t y = 0; y < dh; y++)
            for (uint32_t x = 0; x < dw; x++) {
                /* (vec2(pos) + vec2(0.5)) / vec2(dimensions), depth_reduce.comp:18 */
                float u = ((float)x + 0.5f) / (float)dw;
                float v = ((float)y + 0.5f) / (float)dh;
                dst[(size_t)y * dw + x] = footprint_min(src, srcw, srch, u, v);
            }
    }
}

/* ---- shared cull helpers ---------------------------------------------------- */

/* largest_scale_from_matrix, entity_cull.comp:28-35 / meshlet_cull.comp:28-35 */
static float largest_scale_from_matrix(const float *m) {
    float dx = dot3(m + 0, m + 0), dy = dot3(m + 4, m + 4), dz = dot3(m + 8, m + 8);
    return sqrtf(gmax(dx, gmax(dy, dz)));
}

/* transform_sphere, entity_cull.comp:37-43.  `scale` (optional) receives largest_scale_from_matrix: the compiled
 * shaders contract sphere.w * scale into three fused operations of the occlusion test (occlusion_test below). */
static void transform_sphere(const float *m, const float *sphere, float *out, float *scale) {
    float v[4] = {sphere[0], sphere[1], sphere[2], 1.0f}, p[4];
    mat4_mul_vec4(m, v, p);
    out[0] = p[0] / p[3];
    out[1] = p[1] / p[3];
    out[2] = p[2] / p[3];
    const float sc = largest_scale_from_matrix(m);
    out[3] = sphere[3] * sc;
    if (scale) *scale = sc;
}

/* project_sphere, entity_cull.comp:83-102 (Mara & McGuire 2013), as compiled (entity_cull.comp.spv /
 * meshlet_cull.comp.spv, identical in both): dot(c, c) - r * r is one fused operation fma(-r, r, dot(c, c)), and the
 * remap aabb.xwzy * (0.5, -0.5, 0.5, -0.5) + 0.5 is one fused operation per component. */
static void project_sphere(const float *s, float p00, float p11, float *aabb) {
    float r = s[3];
    float cxx = -s[0], cxy = -s[2];
    float vxx = sqrtf(fmaf(-r, r, dot2(cxx, cxy, cxx, cxy))), vxy = r;
    /* mat2(vx.x, vx.y, -vx.y, vx.x) * cx : columns (vx.x,vx.y), (-vx.y,vx.x) */
    float minx_x = vxx * cxx + (-vxy) * cxy, minx_y = vxy * cxx + vxx * cxy;
    float maxx_x = vxx * cxx + vxy * cxy, maxx_y = (-vxy) * cxx + vxx * cxy;
    float cyx = -s[1], cyy = -s[2];
    float vyx = sqrtf(fmaf(-r, r, dot2(cyx, cyy, cyx, cyy))), vyy = r;
    float miny_x = vyx * cyx + (-vyy) * cyy, miny_y = vyy * cyx + vyx * cyy;
    float maxy_x = vyx * cyx + vyy * cyy, maxy_y = (-vyy) * cyx + vyx * cyy;
    float a0 = minx_x / minx_y * p00, a1 = miny_x / miny_y * p11;
    float a2 = maxx_x / maxx_y * p00, a3 = maxy_x / maxy_y * p11;
    /* aabb.xwzy * (0.5,-0.5,0.5,-0.5) + 0.5 */
    aabb[0] = fmaf(a0, 0.5f, 0.5f);
    aabb[1] = fmaf(a3, -0.5f, 0.5f);
    aabb[2] = fmaf(a2, 0.5f, 0.5f);
    aabb[3] = fmaf(a1, -0.5f, 0.5f);
}

/* Test hook: project_sphere on its own, so that it can be compared with the host mirror of the reference's CPU twin
 * math::project_sphere_clip_space (src/math.rs:170-199), which is the same construction before the uv remap. */
void oracle_project_sphere(const float *sphere, float p00, float p11, float *aabb_uv) {
    project_sphere(sphere, p00, p11, aabb_uv);
}

/* Pass-2 occlusion test shared by entity_cull.comp:146-191 and meshlet_cull.comp:160-205, operation by operation as
 * the compiled shaders hold it (the two binaries agree).  spirv-opt's floating-point folding left its marks here:
 * `sphere.w + z_near`, `sphere.z - sphere.w` and `sphere.z + sphere.w` are fused with the product sphere.w =
 * radius * scale that feeds them (radius = the model-space radius, scale = largest_scale_from_matrix), the
 * orthographic box is fma(size, (-1,-1,1,1), center), and closest_z * r + r * z_far became r * (closest_z + z_far).
 * Mutates sphere[2] in the perspective branch (the mutation persists into the LOD distance, entity_cull.comp:155,204). */
static int occlusion_test(const OrbitGpuCullInfo *ci, float *sphere, float radius, float scale, const float *pyr,
                          uint32_t pw, uint32_t ph) {
    int cullable = 1;
    float aabb[4], closest;
    if (ci->projection_type == 0) {
        sphere[2] = -sphere[2];
        cullable = sphere[2] >= fmaf(radius, scale, ci->z_near);
        project_sphere(sphere, ci->p00_or_width_recipx2, ci->p11_or_height_recipx2, aabb);
        closest = ci->z_near / fmaf(-radius, scale, sphere[2]);
    } else {
        float sr = ci->p00_or_width_recipx2; /* vec2(p00): both axes, quirk 9 */
        float cx = sphere[0] * sr, cy = sphere[1] * sr;
        float bs = sr * sphere[3];
        float b0 = fmaf(bs, -1.0f, cx), b1 = fmaf(bs, -1.0f, cy), b2 = fmaf(bs, 1.0f, cx), b3 = fmaf(bs, 1.0f, cy);
        aabb[0] = fmaf(gclamp(b0, -1.0f, 1.0f), 0.5f, 0.5f);
        aabb[1] = fmaf(gclamp(b1, -1.0f, 1.0f), -0.5f, 0.5f);
        aabb[2] = fmaf(gclamp(b2, -1.0f, 1.0f), 0.5f, 0.5f);
        aabb[3] = fmaf(gclamp(b3, -1.0f, 1.0f), -0.5f, 0.5f);
        float closest_z = fmaf(radius, scale, sphere[2]);
        float r = 1.0f / (ci->z_far - ci->z_near);
        closest = r * (closest_z + ci->z_far);
    }
    if (!cullable) return 1;
    float width = (aabb[2] - aabb[0]) * (float)pw;
    float height = (aabb[3] - aabb[1]) * (float)ph;
    float u = (aabb[0] + aabb[2]) * 0.5f, v = (aabb[1] + aabb[3]) * 0.5f;
    float sampled = oracle_hiz_sample(pyr, pw, ph, u, v, orbit_log2f(gmax(width, height)));
    return closest >= sampled;
}

/* int(i8) / 127.0 (meshlet_cull.comp:121-122) as the compiled shader evaluates it: spirv-opt folded the division
 * by a constant into a multiplication by its reciprocal (meshlet_cull.comp.spv: OpFMul by 0x3C010204).  16 of the 256
 * inputs differ from the IEEE quotient by one ulp. */
static inline float snorm8(int8_t v) { return (float)(int)v * bits2f(0x3C010204u); }

/* coneCull (meshlet_cull.comp:104-106) as compiled: dot(d, axis) >= fma(cutoff, length(d), radius) */
static inline int cone_cull(const float *d, const float *axis, float cutoff, float radius) {
    return dot3(d, axis) >= fmaf(cutoff, sqrtf(dot3(d, d)), radius);
}

static int plane_test(const OrbitGpuCullInfo *ci, const float *sphere) {
    int visible = 1;
    for (uint32_t i = 0; i < ci->cull_plane_count; i++) {
        const float *pl = ci->cull_planes[i];
        float d = dot3(pl, sphere) + pl[3];
        visible = visible && (d > -sphere[3]);
    }
    return visible;
}

static uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static void wr32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }

/* ---- entity_cull ------------------------------------------------------------- */

/* shaders/entity_cull.comp:104-245 under the clear of draw_gen.rs:356-363 and
 * the grid of draw_gen.rs:377.  Returns the number of records that did not fit
 * `dispatch_capacity` (0 = ok); the header x still counts only stored records. */
uint32_t oracle_entity_cull(const OrbitGpuCullInfo *ci, const uint8_t *entity_draw_buffer,
                            uint32_t draw_first, uint32_t entity_draw_count, uint32_t exact_range,
                            const OrbitMeshInfo *mesh_infos, const OrbitEntityData *entities,
                            uint8_t *dispatch_buffer, uint32_t dispatch_capacity,
                            uint32_t *visibility, const float *pyr, uint32_t pw, uint32_t ph) {
    uint32_t count = rd32(entity_draw_buffer);
    const OrbitEntityDraw *draws = (const OrbitEntityDraw *)(entity_draw_buffer + ORBIT_ENTITY_DRAW_HEADER);
    OrbitMeshletDispatch *out = (OrbitMeshletDispatch *)(dispatch_buffer + ORBIT_DISPATCH_HEADER);
    uint32_t total = 0, dropped = 0;
    int meshlet_occ = ci->meshlet_visibility_buffer != ORBIT_NONE;
    /* !exact_range: the reference's grid, ceil(n/256) workgroups of 256 (draw_gen.rs:377), bounded by
     * the in-buffer count (entity_cull.comp:106).  exact_range: the shard [first, first + n) of the
     * multi-GPU path (orbit_entity_cull_range). */
    uint32_t groups = (entity_draw_count + 255u) / 256u;
    uint64_t end64 = exact_range ? (uint64_t)draw_first + entity_draw_count : (uint64_t)groups * 256u;
    uint32_t end = end64 > count ? count : (uint32_t)end64;
    /* pass 2 rewrites whole 32-bit words from ballots; inactive lanes give 0 */
    uint32_t word = 0;
    for (uint32_t g = draw_first; g < end; g++) {
        OrbitEntityDraw ed = draws[g];
        const OrbitMeshInfo *mi = &mesh_infos[ed.mesh_index];
        const OrbitEntityData *en = &entities[ed.entity_index];
        int visible = 1, visible_in_buffer = 1;
        if (ci->occlusion_pass == 1 || ci->occlusion_pass == 2)
            visible_in_buffer = (visibility[g / 32] & (1u << (g % 32))) != 0;
        if (ci->occlusion_pass == 1) visible = visible_in_buffer;
        float mv[16], sphere[4];
        mat4_mul(ci->view_matrix, en->model_matrix, mv);
        float scale;
        transform_sphere(mv, mi->bounding_sphere, sphere, &scale);
        if (visible) visible = plane_test(ci, sphere);
        if (ci->occlusion_pass == 2 && visible) visible = occlusion_test(ci, sphere, mi->bounding_sphere[3], scale, pyr, pw, ph);
        int should_draw = visible;
        if (ci->occlusion_pass == 2) should_draw = visible && (!visible_in_buffer || meshlet_occ);
        if (should_draw) {
            float d[3] = {ci->lod_target_pos_view_space[0] - sphere[0],
                          ci->lod_target_pos_view_space[1] - sphere[1],
                          ci->lod_target_pos_view_space[2] - sphere[2]};
            float lod_distance = sqrtf(dot3(d, d)) - sphere[3];
            float lod_f = orbit_log2f(gmax(lod_distance, 0.0f) / ci->lod_base) / orbit_log2f(ci->lod_step);
            uint32_t lod = f2u_sat(gmax(lod_f + 1.0f, 0.0f));
            lod = umin(umax(lod, ci->min_mesh_lod), ci->max_mesh_lod); /* clamp() */
            lod = umin(lod, mi->lod_count - 1u);
            lod = umin(lod, ORBIT_MAX_MESH_LODS - 1u); /* memory safety; no-op for valid assets */
            OrbitMeshLod ml = mi->mesh_lods[lod];
            uint32_t n = (ml.meshlet_count + S_DISPATCH - 1u) / S_DISPATCH;
            uint32_t vis_off = ed.visibility_offset;
            for (uint32_t j = 0; j < n; j++) {
                OrbitMeshletDispatch r;
                r.entity_index = ed.entity_index;
                r.meshlet_offset = ml.meshlet_offset + S_DISPATCH * j;
                r.meshlet_count = umin(ml.meshlet_count - S_DISPATCH * j, S_DISPATCH);
                r.visibility_offset = vis_off;
                if (total < dispatch_capacity) out[total++] = r;
                else dropped++;
                vis_off += r.meshlet_count / S_DISPATCH;
            }
        }
        if (ci->occlusion_pass == 2) {
            if (g % 32 == 0) word = 0;
            word |= (uint32_t)visible << (g % 32);
            if (g % 32 == 31 || g + 1 == end) visibility[g / 32] = word;
        }
    }
    wr32(dispatch_buffer + 0, total);
    wr32(dispatch_buffer + 4, 1);
    wr32(dispatch_buffer + 8, 1);
    return dropped;
}

/* ---- meshlet_cull ------------------------------------------------------------ */

/* shaders/meshlet_cull.comp:108-255 under the clear of draw_gen.rs:411-417 and
 * the indirect grid of draw_gen.rs:432 (one workgroup of S lanes per record). */
uint32_t oracle_meshlet_cull(const OrbitGpuCullInfo *ci, const uint8_t *dispatch_buffer,
                             const OrbitMeshlet *meshlets, uint8_t *draw_buffer, uint32_t draw_capacity,
                             const OrbitEntityData *entities, const OrbitMaterialData *materials,
                             uint32_t *meshlet_visibility, const float *pyr, uint32_t pw, uint32_t ph) {
    uint32_t nrec = rd32(dispatch_buffer);
    const OrbitMeshletDispatch *recs = (const OrbitMeshletDispatch *)(dispatch_buffer + ORBIT_DISPATCH_HEADER);
    OrbitMeshletDrawCommand *out = (OrbitMeshletDrawCommand *)(draw_buffer + ORBIT_DRAW_HEADER);
    uint32_t total = 0, dropped = 0;
    int meshlet_occ = ci->meshlet_visibility_buffer != ORBIT_NONE;
    for (uint32_t ri = 0; ri < nrec; ri++) {
        OrbitMeshletDispatch rec = recs[ri];
        float mv[16];
        mat4_mul(ci->view_matrix, entities[rec.entity_index].model_matrix, mv);
        uint32_t ballot = 0, prev_word = 0;
        int rw = (ci->occlusion_pass == 1 || ci->occlusion_pass == 2) && meshlet_occ;
        if (rw) prev_word = meshlet_visibility[rec.visibility_offset]; /* lane/32 == 0 for S=32 */
        for (uint32_t lane = 0; lane < S_DISPATCH; lane++) {
            if (lane >= rec.meshlet_count) continue; /* meshlet_cull.comp:111 */
            uint32_t meshlet_index = rec.meshlet_offset + lane;
            const OrbitMeshlet *m = &meshlets[meshlet_index];
            float sphere[4];
            float scale;
            transform_sphere(mv, m->bounding_sphere, sphere, &scale);
            float ax[4] = {snorm8(m->cone_axis[0]), snorm8(m->cone_axis[1]),
                           snorm8(m->cone_axis[2]), 0.0f};
            float axis[4];
            mat4_mul_vec4(mv, ax, axis);
            float cutoff = snorm8(m->cone_cutoff);
            uint32_t alpha_mode = materials[m->material_index].alpha_mode;
            int visible = 1, visible_in_buffer = 1;
            if (rw) visible_in_buffer = (prev_word & (1u << (lane % 32))) != 0;
            if (ci->occlusion_pass == 1) visible = visible_in_buffer;
            if (visible) visible = plane_test(ci, sphere);
            if (visible) {
                /* coneCull, meshlet_cull.comp:104-106,148-158 */
                float cam[3] = {0.0f, 0.0f, 0.0f};
                if (ci->projection_type == 1) {
                    cam[0] = sphere[0] - 0.0f;
                    cam[1] = sphere[1] - 0.0f;
                    cam[2] = sphere[2] - -1.0f;
                }
                if (ci->projection_type <= 1) {
                    float d[3] = {sphere[0] - cam[0], sphere[1] - cam[1], sphere[2] - cam[2]};
                    int cull = cone_cull(d, axis, cutoff, sphere[3]);
                    visible = !cull;
                }
            }
            if (meshlet_occ && ci->occlusion_pass == 2 && visible)
                visible = occlusion_test(ci, sphere, m->bounding_sphere[3], scale, pyr, pw, ph);
            int should_draw = visible && ((shl1(alpha_mode) & ci->alpha_mode_flag) != 0);
            if ((ci->occlusion_pass == 2 && meshlet_occ) && !((shl1(alpha_mode) & ci->noskip_alphamode) != 0))
                should_draw = visible && !visible_in_buffer;
            if (should_draw) {
                OrbitMeshletDrawCommand c;
                c.cmd_first_instance = rec.entity_index;
                c.cmd_instance_count = 1;
                c.cmd_vertex_offset = (int32_t)m->data_offset;
                c.cmd_first_index = (m->data_offset + m->vertex_count) * 4u;
                c.cmd_index_count = (uint32_t)m->triangle_count * 3u;
                c.meshlet_vertex_offset = m->vertex_offset;
                c.meshlet_index = meshlet_index;
                if (total < draw_capacity) out[total++] = c;
                else dropped++;
            }
            ballot |= (uint32_t)visible << lane;
        }
        if (ci->occlusion_pass == 2 && meshlet_occ && rec.meshlet_count > 0)
            meshlet_visibility[rec.visibility_offset] = ballot;
    }
    wr32(draw_buffer, total);
    return dropped;
}

/* ---- task-shader cull (mesh-shading path) ----------------------------------------- */

/* shaders/forward/forward_depth_prepass.task:115-257 (== forward/forward.task:106-248): the same per-meshlet
 * decision as meshlet_cull.comp, one workgroup of S lanes per MeshletDispatch record, but the survivors stay with
 * their record — MeshTaskPayload{entity_index, meshlet_offset, meshlet_indices[]} (types.glsl:196-200) and
 * EmitMeshTasksEXT(task_mesh_count, 1, 1) — instead of being appended to a global list.  One OrbitMeshTaskRecord per
 * record, in dispatch order.  Differences from meshlet_cull.comp that are reproduced literally:
 *   - lanes past meshlet_count do not return: their `visible` stays true (:124) and is written to the visibility
 *     word in pass 2 (:232-254, canonical #else addressing word = offset + lane/32, bit = lane%32);
 *   - payload slots are claimed with atomicAdd(task_mesh_count) (:226): canonical order = ascending lane;
 *     slots past the count are never written by the shader: canonical value 0. */
void oracle_meshlet_task_cull(const OrbitGpuCullInfo *ci, const uint8_t *dispatch_buffer, const OrbitMeshlet *meshlets,
                              const OrbitEntityData *entities, const OrbitMaterialData *materials,
                              uint32_t *meshlet_visibility, const float *pyr, uint32_t pw, uint32_t ph,
                              OrbitMeshTaskRecord *records) {
    uint32_t nrec = rd32(dispatch_buffer);
    const OrbitMeshletDispatch *recs = (const OrbitMeshletDispatch *)(dispatch_buffer + ORBIT_DISPATCH_HEADER);
    int meshlet_occ = ci->meshlet_visibility_buffer != ORBIT_NONE;
    for (uint32_t ri = 0; ri < nrec; ri++) {
        OrbitMeshletDispatch rec = recs[ri];
        OrbitMeshTaskRecord out;
        memset(&out, 0, sizeof(out));
        float mv[16];
        mat4_mul(ci->view_matrix, entities[rec.entity_index].model_matrix, mv);
        uint32_t ballot = 0, prev_word = 0;
        int rw = meshlet_occ && (ci->occlusion_pass == 1 || ci->occlusion_pass == 2);
        if (rw) prev_word = meshlet_visibility[rec.visibility_offset]; /* lane/32 == 0 for S=32 */
        for (uint32_t lane = 0; lane < S_DISPATCH; lane++) {
            int visible = 1; /* :124, also for lanes past the count */
            if (lane < rec.meshlet_count) {
                uint32_t meshlet_index = rec.meshlet_offset + lane;
                const OrbitMeshlet *m = &meshlets[meshlet_index];
                float sphere[4];
                float scale;
                transform_sphere(mv, m->bounding_sphere, sphere, &scale);
                float ax[4] = {snorm8(m->cone_axis[0]), snorm8(m->cone_axis[1]),
                               snorm8(m->cone_axis[2]), 0.0f};
                float axis[4];
                mat4_mul_vec4(mv, ax, axis);
                float cutoff = snorm8(m->cone_cutoff);
                uint32_t alpha_mode = materials[m->material_index].alpha_mode;
                int visible_in_buffer = 1;
                if (rw) visible_in_buffer = (prev_word & (1u << (lane % 32))) != 0;
                if (ci->occlusion_pass == 1) visible = visible_in_buffer;
                if (visible) visible = plane_test(ci, sphere);
                if (visible) {
                    float cam[3] = {0.0f, 0.0f, 0.0f};
                    if (ci->projection_type == 1) {
                        cam[0] = sphere[0] - 0.0f;
                        cam[1] = sphere[1] - 0.0f;
                        cam[2] = sphere[2] - -1.0f;
                    }
                    if (ci->projection_type <= 1) {
                        float d[3] = {sphere[0] - cam[0], sphere[1] - cam[1], sphere[2] - cam[2]};
                        int cull = cone_cull(d, axis, cutoff, sphere[3]);
                        visible = !cull;
                    }
                }
                if (meshlet_occ && ci->occlusion_pass == 2 && visible)
                    visible = occlusion_test(ci, sphere, m->bounding_sphere[3], scale, pyr, pw, ph);
                int should_draw = visible && ((shl1(alpha_mode) & ci->alpha_mode_flag) != 0);
                if ((ci->occlusion_pass == 2 && meshlet_occ) && !((shl1(alpha_mode) & ci->noskip_alphamode) != 0))
                    should_draw = visible && !visible_in_buffer;
                if (should_draw) out.payload.meshlet_indices[out.task_mesh_count++] = (uint8_t)lane; /* :225-228 */
            }
            ballot |= (uint32_t)visible << lane;
        }
        if (meshlet_occ && ci->occlusion_pass == 2) meshlet_visibility[rec.visibility_offset] = ballot;
        out.payload.meshlet_offset = rec.meshlet_offset; /* :255-256 */
        out.payload.entity_index = rec.entity_index;
        records[ri] = out;
    }
}

/* ---- light clusters ------------------------------------------------------------ */

/* cluster_grid_info, src/passes/cluster.rs:63-72 with the canonical log2. */
void oracle_cluster_grid_info(float near, float far, uint32_t z_slices, float *z_scale, float *z_bias) {
    float n = (float)z_slices;
    float log_f_n = orbit_log2f(far / near);
    *z_scale = n / log_f_n;
    *z_bias = -((n * orbit_log2f(near)) / log_f_n);
}

/* shaders/light_cluster/mark_active.comp:27-57 under the zero-fill of
 * cluster.rs:439-456.  depth: [h][w][samples]. */
void oracle_cluster_mark(const OrbitMarkActivePush *pc, const float *depth, uint32_t *masks,
                         OrbitClusterDepthBounds *bounds) {
    uint32_t cx = pc->cluster_count[0], cy = pc->cluster_count[1], cz = pc->cluster_count[2];
    memset(masks, 0, (size_t)cx * cy * 4);
    memset(bounds, 0, (size_t)cx * cy * cz * sizeof(*bounds));
    uint32_t sc = pc->depth_buffer_sample_count;
    for (uint32_t py = 0; py < pc->screen_size[1]; py++)
        for (uint32_t px = 0; px < pc->screen_size[0]; px++) {
            uint32_t tx = px / pc->tile_size_px, ty = py / pc->tile_size_px;
            uint32_t bitmask = 0;
            for (uint32_t s = 0; s < sc; s++) {
                float d = depth[((size_t)py * pc->screen_size[0] + px) * sc + s];
                float linear_z = pc->z_near / d;
                /* cluster_common.glsl:18-20 as compiled (mark_active.comp.spv): uint(fma(log2(z), z_scale, z_bias)) */
                uint32_t slice = f2u_sat(fmaf(orbit_log2f(linear_z), pc->z_scale, pc->z_bias));
                bitmask |= shl1(slice);
                if (slice < cz) {
                    size_t ci = tx + (size_t)ty * cx + (size_t)slice * cx * cy;
                    float inv = 1.0f - d;
                    /* sign / payload of an arithmetic NaN is implementation-defined (x86 keeps the operand's,
                     * gfx950 returns the negated one): canonical quiet NaN 0x7fc00000 */
                    bounds[ci].min_depth = umax(bounds[ci].min_depth, inv != inv ? 0x7fc00000u : f2bits(inv));
                    bounds[ci].max_depth = umax(bounds[ci].max_depth, f2bits(d));
                }
            }
            if (bitmask > 0) masks[tx + ty * cx] |= bitmask;
        }
}

/* shaders/light_cluster/active_cluster_compaction.comp:17-44 in canonical
 * (ascending linear index) order; header per :36-41. */
uint32_t oracle_cluster_compact(const uint32_t cc[3], const uint32_t *masks, uint8_t *unique,
                                uint32_t index_capacity) {
    uint32_t n = 0, dropped = 0;
    uint32_t *idx = (uint32_t *)(unique + ORBIT_COMPACT_HEADER);
    for (uint32_t z = 0; z < cc[2]; z++)
        for (uint32_t y = 0; y < cc[1]; y++)
            for (uint32_t x = 0; x < cc[0]; x++)
                if (masks[x + y * cc[0]] & shl1(z)) {
                    if (n < index_capacity) idx[n++] = x + y * cc[0] + z * cc[0] * cc[1];
                    else dropped++;
                }
    wr32(unique + 0, (n + 255u) / 256u);
    wr32(unique + 4, 1);
    wr32(unique + 8, 1);
    wr32(unique + 12, n);
    return dropped;
}

/* screen_to_view, light_culling.comp:34-48 */
static void screen_to_view(const OrbitClusterCullInfo *in, float sx, float sy, float *out) {
    float tx = sx / (float)in->screen_size[0], ty = sy / (float)in->screen_size[1];
    float clip[4] = {tx * 2.0f - 1.0f, (1.0f - ty) * 2.0f - 1.0f, 1.0f, 1.0f};
    float v[4];
    mat4_mul_vec4(in->screen_to_view_matrix, clip, v);
    out[0] = v[0] / v[3];
    out[1] = v[1] / v[3];
    out[2] = v[2] / v[3];
}

/* line_intersection_to_z_plane with a = eye = 0, light_culling.comp:50-60 */
static void line_z(const float *b, float zd, float *out) {
    float ab[3] = {b[0] - 0.0f, b[1] - 0.0f, b[2] - 0.0f};
    /* dot(normal, a) with normal (0,0,-1), a = 0; dot(normal, ab) */
    float dna = (0.0f * 0.0f + 0.0f * 0.0f) + -1.0f * 0.0f;
    float dnab = (0.0f * ab[0] + 0.0f * ab[1]) + -1.0f * ab[2];
    float t = (zd - dna) / dnab;
    out[0] = 0.0f + t * ab[0];
    out[1] = 0.0f + t * ab[1];
    out[2] = 0.0f + t * ab[2];
}

/* compute_cluster_volume, light_culling.comp:62-90 */
void oracle_cluster_aabb(const OrbitClusterCullInfo *in, const OrbitClusterDepthBounds *bounds,
                         uint32_t cluster_index, float *mn, float *mx) {
    uint32_t cx = in->cluster_count[0], cy = in->cluster_count[1];
    uint32_t idx = cluster_index;
    uint32_t z = idx / (cx * cy);
    idx -= z * cx * cy;
    uint32_t y = idx / cx;
    idx -= y * cx;
    uint32_t x = idx;
    float minx = (float)(x * in->tile_size_px), miny = (float)(y * in->tile_size_px);
    float maxx = gmin(minx + (float)in->tile_size_px, (float)in->screen_size[0]);
    float maxy = gmin(miny + (float)in->tile_size_px, (float)in->screen_size[1]);
    float minv[3], maxv[3];
    screen_to_view(in, minx, miny, minv);
    screen_to_view(in, maxx, maxy, maxv);
    float min_depth = 1.0f - bits2f(bounds[cluster_index].min_depth);
    float max_depth = bits2f(bounds[cluster_index].max_depth);
    float cnear = in->z_near / max_depth, cfar = in->z_near / min_depth;
    float p[4][3];
    line_z(minv, cnear, p[0]);
    line_z(minv, cfar, p[1]);
    line_z(maxv, cnear, p[2]);
    line_z(maxv, cfar, p[3]);
    for (int i = 0; i < 3; i++) {
        mn[i] = gmin(gmin(p[0][i], p[1][i]), gmin(p[2][i], p[3][i]));
        mx[i] = gmax(gmax(p[0][i], p[1][i]), gmax(p[2][i], p[3][i]));
    }
}

/* is_light_in_cluster, light_culling.comp:92-119 */
static int light_in_cluster(const OrbitClusterCullInfo *in, const OrbitLightData *l, const float *mn,
                            const float *mx) {
    if (l->light_type != ORBIT_LIGHT_TYPE_POINT) return 1;
    float pos[4] = {l->position[0], l->position[1], l->position[2], 1.0f}, c[4];
    mat4_mul_vec4(in->world_to_view_matrix, pos, c);
    float sq = 0.0f;
    for (int i = 0; i < 3; i++) {
        float v = c[i];
        /* as compiled (light_culling.comp.spv): sqr_dist = fma(d, d, sqr_dist) */
        if (v < mn[i]) sq = fmaf(mn[i] - v, mn[i] - v, sq);
        if (v > mx[i]) sq = fmaf(v - mx[i], v - mx[i], sq);
    }
    return sq <= l->outer_radius * l->outer_radius;
}

/* light_culling.comp:121-151 under the clear of cluster.rs:575-581; ranges are
 * allocated in compacted-list order. */
uint32_t oracle_cluster_assign(const OrbitClusterCullInfo *in, const uint8_t *unique,
                               const OrbitClusterDepthBounds *bounds, const OrbitLightData *lights,
                               uint8_t *light_index_buffer, uint32_t light_index_capacity,
                               uint32_t *offset_image) {
    uint32_t n = rd32(unique + 12);
    const uint32_t *idx = (const uint32_t *)(unique + ORBIT_COMPACT_HEADER);
    uint32_t *out = (uint32_t *)(light_index_buffer + ORBIT_LIGHT_INDEX_HEADER);
    uint32_t total = 0, dropped = 0;
    for (uint32_t u = 0; u < n; u++) {
        uint32_t ci = idx[u];
        float mn[3], mx[3];
        oracle_cluster_aabb(in, bounds, ci, mn, mx);
        uint32_t count = 0;
        for (uint32_t l = 0; l < in->global_light_count; l++)
            if (light_in_cluster(in, &lights[l], mn, mx)) count++;
        count = umin(count, ORBIT_MAX_LIGHTS_PER_CLUSTER);
        uint32_t offset = total;
        total += count;
        uint32_t written = 0;
        for (uint32_t l = 0; written < count; l++)
            if (light_in_cluster(in, &lights[l], mn, mx)) {
                if (offset + written < light_index_capacity) out[offset + written] = l;
                else dropped++;
                written++;
            }
        offset_image[2 * (size_t)ci + 0] = offset;
        offset_image[2 * (size_t)ci + 1] = count;
    }
    wr32(light_index_buffer, total);
    return dropped;
}

/* ---- config 1: frustum-only sphere cull (BASELINE.json configs[0]) -------------- */

/* The plumbing case: N world-space spheres against the 5 view-space planes of
 * forward.rs:264,268 with the test of entity_cull.comp:138-143; model =
 * translation so the view-space centre is view * (c,1).  Returns the count. */
uint32_t oracle_sphere_frustum_cull(const float *view, const float planes[][4], uint32_t plane_count,
                                    const float *spheres, uint32_t n, uint32_t *bitset) {
    uint32_t visible_count = 0;
    memset(bitset, 0, ((size_t)n + 31) / 32 * 4);
    for (uint32_t i = 0; i < n; i++) {
        float v[4] = {spheres[4 * i], spheres[4 * i + 1], spheres[4 * i + 2], 1.0f}, p[4];
        mat4_mul_vec4(view, v, p);
        float c[3] = {p[0] / p[3], p[1] / p[3], p[2] / p[3]};
        int vis = 1;
        for (uint32_t k = 0; k < plane_count; k++) {
            float d = dot3(planes[k], c) + planes[k][3];
            vis = vis && (d > -spheres[4 * i + 3]);
        }
        if (vis) {
            bitset[i / 32] |= 1u << (i % 32);
            visible_count++;
        }
    }
    return visible_count;
}
