// meshlet_common.h — what the meshlet-cull translation units share on the device: the wave tile's LDS slab, the
// per-meshlet evaluation (shaders/meshlet_cull.comp:113-213 of the reference repo), the row loads, the pass-2
// candidate ring and the tile setup.  Header-only (every function is forceinline); the tuning macros of the path live
// here, so an A/B build varies one macro and recompiles only the unit that reads it (tools/mkvariant_any.sh).
//
//   wave tile  = 16 dispatch records = 8 rows x (2 records on 64 lanes), one
//                meshlet per lane per row -> every row is one fully coalesced
//                2 KiB read of 32-B meshlets (two dwordx4 per lane).
//   setup      = the 16 records' (view x model) matrices are built ONCE per tile
//                with all 64 lanes (lane = record*4 + column) and staged in a
//                wave-private LDS slab; rows then read them as broadcast
//                ds_read_b128.  The reference recomputes the 4x4 product in
//                every one of the 32 lanes of a workgroup (meshlet_cull.comp:117).
#pragma once
#include <type_traits>

#include "kernels.h"
#include "scan.h"

namespace orbit {

namespace {

// survivors per tile whose command words are cached by meshlet_eval for the emit launch
constexpr uint32_t kPayloadCap = 128;

// Range-checked buffer accesses (no branch, no pointer select around a load: a lane that must not read is given an
// offset outside the resource and gets zeros).
constexpr uint32_t kBufFlags = 0x00020000u; // raw buffer, 32-bit data format
constexpr uint32_t kNoOffset = 0xFFFFFF00u; // beyond every resource of this launch, also after a lane's +60 bytes
                                            // (orbit_meshlet_stream_create bounds the stream's arrays: 2^27 meshlets)

// The streams a row reads, as buffer resources based at the stream's first meshlet: a row's address arithmetic is
// one subtraction, one select and a shift per array instead of a 64-bit pointer select per load.
// Cache policy of the derived streams' one-pass reads: nt (non-temporal, aux bit 1).  The evaluation's rows are read
// once per frame and are four times the L2 + MALL: loaded with the default policy they evict each other and everything
// else on their way through; as nt loads the same kernel streams 9 % faster (in-process A/B of both builds, both
// orders: 0.2054 / 0.2146 -> 0.1901 / 0.1929 ms; profiles/r03_notes.md).  The chain emit's count stream likewise (-1 us);
// its command STORES not (nt stores: +4 ... +17 us), nor the 32-B rows of the Meshlet-buffer evaluation (nothing for
// the evaluation, +9 us for the payload emit behind it).
#ifndef ORBIT_EVAL_LOAD_AUX
#define ORBIT_EVAL_LOAD_AUX 2
#endif
#ifndef ORBIT_EMIT_CNT_AUX
#define ORBIT_EMIT_CNT_AUX 2
#endif
#ifndef ORBIT_EMIT_STORE_AUX
#define ORBIT_EMIT_STORE_AUX 0
#endif
struct StreamRsrc {
    __amdgpu_buffer_rsrc_t sphere, cone, mat;
    uint32_t first, count;
};

// Everything a row's lanes read of their record sits behind ONE address (record-major, 112 B): a row's record is
// 2 r + lane / 32, so every read is "the wave's (slab + half) base + a constant" and a row spends no vector
// instruction on LDS addressing (field-major arrays cost one address computation per field and row).
struct __attribute__((aligned(16))) RecordLds {
    float4 mcol[4]; // view x model, column-major
    uint4 rec;      // MeshletDispatch
    float4 tzero;   // column 3 x 0.0f: the w = 0 term of the cone-axis transform (:121)
    uint2 amask;    // CLS: {(:207) bits, (:210) bits} of the record's 32 lanes, from their alpha classes
    float scale;    // largest_scale_from_matrix(mat3(M))
    uint32_t prev;  // pass 1: last frame's visibility word of the record
};
static_assert(sizeof(RecordLds) == 112, "RecordLds layout");

struct __attribute__((aligned(16))) WaveTileLds {
    RecordLds r[kTileRecords];
    uint64_t draw_mask[kTileRows]; // should-draw ballots of the tile's rows (kept here, not in SGPRs)
    uint64_t vis_mask[kTileRows];  // pass 2 with meshlet occlusion: visible ballots, assembled by the candidate flushes
    // bit 8 r: both records of row r have an affine view x model matrix (last row 0 0 0 1) with a finite translation:
    // for finite centres their w is exactly 1 and the w = 0 term of the cone-axis transform exactly +-0 (setup_write)
    uint64_t affine_rows;
    uint64_t pad_;
};

// The cull planes live in LDS (one copy per block, written once): a row reads
// them as back-to-back broadcast ds_read_b128 instead of one scalar load + full
// wait per plane per row.
struct __attribute__((aligned(16))) PlaneLds {
    float4 plane[ORBIT_MAX_CULL_PLANES];
    float4 std0; // the symmetric five-plane frustum (MeshletCullParams::std_planes): a, c, b, c'
    float4 std1; //                                                                     e, w, -, -
};

__device__ __forceinline__ void planes_to_lds(const MeshletCullParams &p, PlaneLds &P) {
    if (threadIdx.x < ORBIT_MAX_CULL_PLANES) {
        const float *pl = p.ci.cull_planes[threadIdx.x];
        P.plane[threadIdx.x] = make_float4(pl[0], pl[1], pl[2], pl[3]);
    }
    if (threadIdx.x == ORBIT_MAX_CULL_PLANES) {
        P.std0 = make_float4(p.stdp[0], p.stdp[1], p.stdp[2], p.stdp[3]);
        P.std1 = make_float4(p.stdp[4], p.stdp[5], 0.0f, 0.0f);
    }
}

// Verdicts are wave masks, not lane booleans: every comparison is balloted as it is made and the masks are combined
// with scalar instructions (a lane boolean that is an AND / OR of several comparisons costs two vector instructions
// to turn back into a ballot).  lane_of() is the way back where a lane predicate is needed: the mask IS the
// predicate register, no instruction.
__device__ __forceinline__ uint64_t ballot(bool c) { return __builtin_amdgcn_ballot_w64(c); }
__device__ __forceinline__ bool lane_of(uint64_t wave_mask) { return __builtin_amdgcn_inverse_ballot_w64(wave_mask); }

// The plane loop (:139-146) for the symmetric five-plane frustum and a FINITE view-space centre: left / right share
// their products (((a x + 0 y) + c z) + 0 and ((-a x + 0 y) + c z) + 0 are t1 + t2 and t2 - t1 up to the sign of a
// zero, which no comparison sees), so do bottom / top; the near plane is e z + w.
__device__ __forceinline__ uint64_t planes_std(const PlaneLds &P, const Sphere &s) {
    const float4 k = P.std0;
    const float4 n = P.std1;
    const float nr = -s.r;
#if ORBIT_CONTRACT
    // the same planes as fma chains: (a, 0, c, 0) . s = fma(c, z, fma(0, y, a x)) = fma(c, z, a x) for a finite y, its
    // mirror fma(c, z, -(a x)); the near plane's fma(e, z, fma(0, y, 0 x)) is the rounded product e z
    const float t1 = k.x * s.x, t3 = k.z * s.y, t5 = n.x * s.z;
    uint64_t in = ballot(__builtin_fmaf(k.y, s.z, t1) > nr);
    in &= ballot(__builtin_fmaf(k.y, s.z, -t1) > nr);
    in &= ballot(__builtin_fmaf(k.w, s.z, t3) > nr);
    in &= ballot(__builtin_fmaf(k.w, s.z, -t3) > nr);
    in &= ballot((t5 + n.y) > nr);
#else
    const float t1 = k.x * s.x, t2 = k.y * s.z, t3 = k.z * s.y, t4 = k.w * s.z, t5 = n.x * s.z;
    uint64_t in = ballot((t1 + t2) > nr);
    in &= ballot((t2 - t1) > nr);
    in &= ballot((t3 + t4) > nr);
    in &= ballot((t4 - t3) > nr);
    in &= ballot((t5 + n.y) > nr);
#endif
    return in;
}

// Plane loop, meshlet_cull.comp:139-146, planes from LDS in batches of up to
// three back-to-back broadcast reads (one wait per batch).  Straight-line: the
// distances of a batch are independent chains.  The batch shape follows the
// (wave-uniform) plane count, so no plane past the count is ever evaluated.
template <uint32_t N>
__device__ __forceinline__ uint64_t plane_batch(const PlaneLds &P, uint32_t base, const Sphere &s) {
    float4 pl[N];
#pragma unroll
    for (uint32_t i = 0; i < N; i++) pl[i] = P.plane[base + i];
    uint64_t in = ~0ull;
#pragma unroll
    for (uint32_t i = 0; i < N; i++) {
        const float d = dot3(pl[i].x, pl[i].y, pl[i].z, s.x, s.y, s.z) + pl[i].w;
        in &= ballot(d > -s.r);
    }
    return in;
}

__device__ __forceinline__ uint64_t plane_test_lds(const PlaneLds &P, uint32_t n, const Sphere &s) {
    uint64_t visible = ~0ull;
#pragma unroll
    for (uint32_t base = 0; base < ORBIT_MAX_CULL_PLANES; base += 3) {
        if (base + 3 <= n) visible &= plane_batch<3>(P, base, s);
        else if (base + 2 == n) visible &= plane_batch<2>(P, base, s);
        else if (base + 1 == n) visible &= plane_batch<1>(P, base, s);
    }
    return visible;
}

// sqrtf, correctly rounded (what -fhip-fp32-correctly-rounded-divide-sqrt makes of it), without the range scaling
// and the zero / infinity fix-up where no lane of the wave needs them: v_sqrt_f32 is within one ulp, the two
// neighbours are tested with exact fma residuals.  x in [2^-96, inf) is the range the compiler's own sequence runs
// unscaled; anything else (zero, tiny, infinite, NaN) takes sqrtf itself.
__device__ __forceinline__ float sqrt_rn(float x) {
    const uint64_t plain = ballot(x >= 0x1p-96f) & ballot(x < __builtin_inff());
    if (plain != ballot(true)) return sqrtf(x);
    const float y = __builtin_amdgcn_sqrtf(x);
    const float ym = __uint_as_float(__float_as_uint(y) - 1u), yp = __uint_as_float(__float_as_uint(y) + 1u);
    const float rm = __builtin_fmaf(-ym, y, x), rp = __builtin_fmaf(-yp, y, x);
    float r = rm <= 0.0f ? ym : y;
    r = rp > 0.0f ? yp : r;
    return r;
}

// NR rows of a tile in flight: 2 x 16 B per lane per row (+ previous-frame words).
template <int NR>
struct RowRegs {
    uint4 a[NR], b[NR];
    uint32_t prev[NR];
};

// materials[i].alpha_mode (:124) is only ever used as (1 << alpha_mode) & flag
// (:207, :210), so the per-block LDS table (filled when the caller states a
// material count <= kAlphaTable) holds the two predicate bits per material;
// without the table they come from a global gather per meshlet.
constexpr uint32_t kAlphaTable = 4096;
struct AlphaLds {
    uint8_t bits[kAlphaTable];
};

// bit0: (1 << mode) & alpha_mode_flag != 0 (:207), bit1: (1 << mode) & noskip_alphamode != 0 (:210)
__device__ __forceinline__ uint32_t alpha_bits(const OrbitGpuCullInfo &ci, uint32_t alpha_mode) {
    const uint32_t m = shl1(alpha_mode);
    return ((m & ci.alpha_mode_flag) != 0u ? 1u : 0u) | ((m & ci.noskip_alphamode) != 0u ? 2u : 0u);
}

__device__ __forceinline__ bool alpha_table_fill(const MeshletCullParams &p, AlphaLds &A) {
    const bool use = p.material_count != 0u && p.material_count <= kAlphaTable;
    if (use)
        for (uint32_t i = threadIdx.x; i < p.material_count; i += blockDim.x)
            A.bits[i] = (uint8_t)alpha_bits(p.ci, p.materials[i].alpha_mode);
    return use;
}

// The two predicate bits of lane `ml` of record `rid`: from the material table / gather, or — CLS — from the record's
// masks in the slab (setup_write_cls).
template <bool CLS>
__device__ __forceinline__ uint32_t lane_alpha_bits(const MeshletCullParams &p, const WaveTileLds &L, const AlphaLds *A,
                                                    uint32_t rid, uint32_t ml, uint32_t word7) {
    if constexpr (CLS) {
        const uint2 am = L.r[rid].amask;
        return ((am.x >> ml) & 1u) | (((am.y >> ml) & 1u) << 1);
    } else {
        const uint32_t mi = word7 & 0xFFFFu; // inactive lanes carry material 0 from the zero page (in bounds; unused)
        return A ? (uint32_t)A->bits[mi & (kAlphaTable - 1u)] : alpha_bits(p.ci, p.materials[mi].alpha_mode);
    }
}

// wave-uniform, and a scalar for the compiler: row r's two records are affine with finite translations (setup_write)
__device__ __forceinline__ bool row_is_affine(const WaveTileLds &L, int r) {
    const uint32_t w = reinterpret_cast<const uint32_t *>(&L.affine_rows)[r >> 2];
    return (((uint32_t)__builtin_amdgcn_readfirstlane((int)w) >> (8 * (r & 3))) & 1u) != 0u;
}

// Survivor payload (12 B): Meshlet::vertex_offset, Meshlet::data_offset, and
// {triangle_count:8 | vertex_count:8 | lane within the record:5 << 4 | tile-local record id:4} — the material index is
// not part of a command, its bits carry where the survivor sat in the tile, from which the emit launch gets the
// meshlet index (record's meshlet_offset + lane) and the entity index (record's entity_index).
__device__ __forceinline__ Payload make_payload(uint32_t ml, uint32_t rid, const uint4 &b) {
    return Payload{b.y, b.z, (b.w & 0xFFFF0000u) | (ml << 4) | rid};
}
// Cache policy of the payload stores (A/B builds: tools/ab_strict.py; 0 = plain stores, the product; 2 = nt, 16 = sc1)
#ifndef ORBIT_PAYLOAD_STORE_AUX
#define ORBIT_PAYLOAD_STORE_AUX 0
#endif
__device__ __forceinline__ void store_payload(Payload *tile_payload, uint32_t rank, const Payload &v) {
#if ORBIT_PAYLOAD_STORE_AUX == 0
    tile_payload[rank] = v;
#else
    typedef unsigned int v3u __attribute__((ext_vector_type(3)));
    const __amdgpu_buffer_rsrc_t R = __builtin_amdgcn_make_buffer_rsrc(tile_payload, 0, kPayloadCap * 12u, kBufFlags);
    const v3u w = {v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b96(w, R, rank * 12u, 0, ORBIT_PAYLOAD_STORE_AUX);
#endif
}

// One meshlet (one lane): meshlet_cull.comp:113-213.  PASS = cull_info.occlusion_pass
// and PROJ = cull_info.projection_type are compile-time (PROJ < 0: read at run time).
// Geometry part (:117-158): view-space sphere `s`, pass-1 gate, plane loop, cone test.
// `affine_row` (wave-uniform): both records of the row have an affine matrix with a finite translation (WaveTileLds::
// affine_rows).  Then a FINITE centre has w = ((0 cx + 0 cy) + 0 cz) + 1 = 1 exactly — no w, no division — and the
// w = 0 term of the cone-axis transform is +-0, which no comparison sees; and a finite view-space centre is what the
// shortened plane tests of the symmetric frustum need (planes_std).  One non-finite lane (its products with 0 are NaN
// in the reference's arithmetic) sends the whole row down the literal path.
template <int PROJ, bool FENCE = true>
__device__ __forceinline__ uint64_t eval_geometry_mask(const MeshletCullParams &p, const WaveTileLds &L,
                                                       const PlaneLds &P, uint32_t rid, const uint4 &a, const uint4 &b,
                                                       Sphere &s, bool affine_row = false) {
    const OrbitGpuCullInfo &ci = p.ci;
    const RecordLds &R = L.r[rid];
    const float4 m0 = R.mcol[0], m1 = R.mcol[1], m2 = R.mcol[2], m3 = R.mcol[3];
    const float cx = __uint_as_float(a.x), cy = __uint_as_float(a.y), cz = __uint_as_float(a.z);
    // transform_sphere (:37-43): matrix * vec4(c, 1), divide by w.
    float px = madd(m3.x, 1.0f, madd(m2.x, cz, madd(m1.x, cy, m0.x * cx)));
    float py = madd(m3.y, 1.0f, madd(m2.y, cz, madd(m1.y, cy, m0.y * cx)));
    float pz = madd(m3.z, 1.0f, madd(m2.z, cz, madd(m1.z, cy, m0.z * cx)));
    // (a non-finite centre makes all three non-finite: each holds a product with it)
    const uint64_t nonfinite = ballot(!__builtin_isfinite(px)) | ballot(!__builtin_isfinite(py)) | ballot(!__builtin_isfinite(pz));
    // wave-uniform and held as a scalar integer (not a lane boolean)
    const uint32_t literal = (affine_row ? 0u : 1u) | (uint32_t)nonfinite | (uint32_t)(nonfinite >> 32);
    if (literal != 0u) {
        const float pw = madd(m3.w, 1.0f, madd(m2.w, cz, madd(m1.w, cy, m0.w * cx)));
        if (pw != 1.0f) { // x / 1.0f == x bit-for-bit, so the affine case skips three IEEE divisions
            px = px / pw;
            py = py / pw;
            pz = pz / pw;
        }
    }
    s.x = px;
    s.y = py;
    s.z = pz;
    s.r = __uint_as_float(a.w) * R.scale;

    // keeps the scheduler from hoisting the cone test's and the plane loop's LDS reads above the sphere transform:
    // without it the class variant of the evaluation holds 128 registers and spills (profiles/r03_notes.md).
    // (FENCE = false: the one-launch cull — a lone wave per SIMD with registers to spare wants exactly that hoisting)
    if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
    // The cone test (:148-158) does not depend on the plane loop: both are evaluated
    // and combined, which gives the scheduler two independent chains per meshlet.
    uint64_t cull;
    {
        // cone axis / cutoff (:121-122): int(i8) / 127.0, axis = (M * vec4(a, 0)).xyz
        const int ax_i = (int)(int8_t)(b.x & 0xFF), ay_i = (int)(int8_t)((b.x >> 8) & 0xFF);
        const int az_i = (int)(int8_t)((b.x >> 16) & 0xFF), co_i = (int)(int8_t)(b.x >> 24);
        const float ax = snorm8(ax_i), ay = snorm8(ay_i), az = snorm8(az_i);
        const float cutoff = snorm8(co_i);
        float axx = madd(m2.x, az, madd(m1.x, ay, m0.x * ax));
        float axy = madd(m2.y, az, madd(m1.y, ay, m0.y * ax));
        float axz = madd(m2.z, az, madd(m1.z, ay, m0.z * ax));
        if (literal != 0u) {
            const float4 tz = R.tzero; // m3 * 0.0f
            axx += tz.x;
            axy += tz.y;
            axz += tz.z;
        }
        const bool ortho = PROJ < 0 ? ci.projection_type == 1 : PROJ == 1;
        float dx, dy, dz;
        if (ortho) { // camera_position = center - (0, 0, -1)
            const float camx = s.x - 0.0f, camy = s.y - 0.0f, camz = s.z - -1.0f;
            dx = s.x - camx;
            dy = s.y - camy;
            dz = s.z - camz;
        } else { // camera_position = 0
            dx = s.x - 0.0f;
            dy = s.y - 0.0f;
            dz = s.z - 0.0f;
        }
        // as compiled: dot(d, axis) >= fma(cutoff, length(d), radius)  (oracle/orbit_oracle.c cone_cull)
        cull = ballot(dot3(dx, dy, dz, axx, axy, axz) >= __builtin_fmaf(cutoff, sqrt_rn(dot3(dx, dy, dz, dx, dy, dz)), s.r));
    }
    // :139-146
    const uint64_t inside = (p.std_planes != 0u && literal == 0u) ? planes_std(P, s) : plane_test_lds(P, ci.cull_plane_count, s);
    return inside & ~cull;
}

// The lane's own verdict (pass 1 gates on last frame's bit first, :137).
template <int PASS, int PROJ>
__device__ __forceinline__ bool eval_geometry(const MeshletCullParams &p, const WaveTileLds &L, const PlaneLds &P,
                                              uint32_t rid, const uint4 &a, const uint4 &b, bool visible_in_buffer,
                                              Sphere &s, bool affine_row = false) {
    const bool visible = lane_of(eval_geometry_mask<PROJ>(p, L, P, rid, a, b, s, affine_row));
    return PASS == 1 ? (visible & visible_in_buffer) : visible;
}

// :207-213 from the final `visible`, the material's predicate bits and last frame's bit.
__device__ __forceinline__ bool should_draw_of(bool visible, uint32_t abits, bool visible_in_buffer, bool occ2) {
    bool should_draw = visible && (abits & 1u) != 0;                                 // :207
    if (occ2 && (abits & 2u) == 0) should_draw = visible && !visible_in_buffer;      // :210-213
    return should_draw;
}

// Issues the meshlet loads of rows R0 .. R0+NR-1 of a tile (2 x 16 B per lane per row; STREAM: 16 + 4 + 2 B from the
// derived streams — words y/z of `b` and the counts in w, which only a command needs, stay zero).
template <int PASS, int R0, int NR, bool STREAM = false, bool CLS = false>
__device__ __forceinline__ void rows_load(const MeshletCullParams &p, const WaveTileLds &L, int lane,
                                          RowRegs<NR> &t, const StreamRsrc *SR = nullptr) {
    const bool meshlet_occ = p.ci.meshlet_visibility_buffer != ORBIT_NONE;
    const bool rw_vis = PASS != 0 && meshlet_occ;
    const uint32_t half = lane >> 5, ml = lane & 31;
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const uint4 rec = L.r[2 * (R0 + k) + half].rec;
        // straight-line loads (exact vmcnt accounting): lanes past meshlet_count (:111) read the zero page
        const bool active = ml < rec.z;
        if constexpr (STREAM) {
            // a meshlet the stream does not cover is not read from it: zeros, and the call's status says so
            const uint32_t rel = rec.y + ml - SR->first;
            const bool in = active & (rel < SR->count);
            if (active & !in) latch_status(p.status, ORBIT_E_RANGE);
            const uint32_t i = in ? rel : kNoOffset >> 4; // x 16, x 4, x 2: outside each of the three resources
            const auto sp = __builtin_amdgcn_raw_buffer_load_b128(SR->sphere, i << 4, 0, ORBIT_EVAL_LOAD_AUX);
            const uint32_t c = __builtin_amdgcn_raw_buffer_load_b32(SR->cone, i << 2, 0, ORBIT_EVAL_LOAD_AUX);
            t.a[k] = make_uint4(sp[0], sp[1], sp[2], sp[3]);
            uint32_t m = 0; // CLS: the predicate bits come from the alpha classes in the slab (setup_write_cls)
            if constexpr (!CLS) m = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(SR->mat, i << 1, 0, ORBIT_EVAL_LOAD_AUX);
            t.b[k] = make_uint4(c, 0u, 0u, m);
        } else {
            const uint4 *src = active ? reinterpret_cast<const uint4 *>(p.meshlets + (rec.y + ml))
                                      : reinterpret_cast<const uint4 *>(p.zero_page);
            t.a[k] = src[0]; // (default cache policy: nt does nothing for this evaluation and costs the payload emit 9 us)
            t.b[k] = src[1];
        }
        t.prev[k] = 0u;
        if (PASS != 0) { // word = visibility_offset + lane/32, :129-134
            const uint32_t *vp = (rw_vis && active) ? p.meshlet_visibility + rec.w
                                                    : reinterpret_cast<const uint32_t *>(p.zero_page);
            t.prev[k] = *vp;
        }
    }
}

// Evaluates rows R0 .. R0+NR-1 of a tile: should-draw ballots into the slab, survivor payload (optional).  `total`
// is the tile's running survivor count; the new count is returned.  This is pass 0, or pass 2 WITHOUT meshlet
// occlusion (row_eval_occ has the other case): no visibility word is read or written, and a meshlet is drawn iff it
// exists (:111), passes the geometry tests and its material the :207 test.  Every lane evaluates — lanes past
// meshlet_count hold zeros, which are finite and harmless — and the verdict is masked: no divergent region around a
// row.
template <int PASS, int PROJ, int R0, int NR, bool CLS = false, uint32_t CAP = kPayloadCap, bool FENCE = true>
__device__ __forceinline__ uint32_t rows_eval(const MeshletCullParams &p, WaveTileLds &L, const PlaneLds &P,
                                              const AlphaLds *A, int lane, const RowRegs<NR> &t, uint32_t total,
                                              Payload *payload) {
    // (PASS == 1: the streamed form of pass 1, taken by the one-launch cull of cull_fused.hip — a lane is gated on last
    // frame's bit first, :137; the large-scene launch is list-driven instead: meshlet_pass1_body)
    static_assert(PASS != 1 || !CLS, "pass 1 reads the Meshlet buffer");
    const uint32_t half = lane >> 5, ml = lane & 31;
#pragma unroll
    for (int k = 0; k < NR; k++) {
        const int r = R0 + k;
        const uint32_t rid = 2 * r + half;
        uint64_t allow; // lanes that hold a meshlet whose material passes (:207)
        if constexpr (CLS) {
            // from the alpha classes, already cut to meshlet_count (setup_write_cls): two uniform words
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.r[2 * r].amask.x);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.r[2 * r + 1].amask.x);
            allow = (uint64_t)hi << 32 | lo;
        } else {
            const uint32_t abits = lane_alpha_bits<false>(p, L, A, rid, ml, t.b[k].w);
            allow = ballot(ml < L.r[rid].rec.z) & ballot((abits & 1u) != 0u);
        }
        if constexpr (PASS == 1) { // visible = visible_in_buffer (:123,:137); without the buffer the bit stays true (:128)
            if (p.ci.meshlet_visibility_buffer != ORBIT_NONE) allow &= ballot(((t.prev[k] >> ml) & 1u) != 0u);
        }
#ifdef ORBIT_TRIAGE_NO_EVAL // scratch builds only (tools/mkvariant_any.sh): a free predicate with the same survival ratio
        const uint64_t geo = ballot(((t.a[k].x + (t.b[k].x >> 31)) % 1000u) < 95u);
#else
        Sphere s;
        const uint64_t geo = eval_geometry_mask<PROJ, FENCE>(p, L, P, rid, t.a[k], t.b[k], s, row_is_affine(L, r));
#endif
        const uint64_t mask = geo & allow;
        if (lane == 0) L.draw_mask[r] = mask;
        if (payload && lane_of(mask)) {
            // survivor payload at the tile-local rank: everything the command needs except its
            // slot, so the emit launch reads one contiguous run per tile and rebuilds nothing
            const uint32_t rank = total + lane_prefix(mask);
            if (rank < CAP) {
                if constexpr (CAP == kPayloadCap) store_payload(payload, rank, make_payload(ml, rid, t.b[k]));
                else payload[rank] = make_payload(ml, rid, t.b[k]); // (the one-launch cull's LDS payload)
            }
        }
        total += (uint32_t)__popcll(mask);
    }
    return total;
}

// ---------------------------------------------------------------------------
// Pass 2 with meshlet occlusion: the HiZ test (:161-205) is by far the longest
// part of the evaluation (projected bounds with four IEEE divisions, log2, four
// texel gathers), and only the ~10 % of the lanes that survive planes + cone run
// it — inline, every 64-meshlet row would issue it for a handful of lanes.
// Instead the survivors of the geometry tests ("candidates") are appended, in
// canonical (row, lane) order, to a wave-private LDS ring with what the rest of
// the decision needs (view-space sphere, Meshlet words 5..7, predicate bits);
// whenever the ring holds >= kCandFlush of them, up to 64 are tested at once
// with full lanes, the results OR-ed into the tile's per-row masks and the
// drawn ones appended to the payload.  Ring order == canonical order, so the
// payload ranks are the same as with the inline test.
constexpr uint32_t kCandRing = 96, kCandFlush = 32; // < kCandFlush left after a flush, + one row of 64 <= kCandRing

template <uint32_t N>
struct __attribute__((aligned(16))) CandLds {
    float4 sphere[N]; // view-space centre, MODEL-space radius (x the record's scale = view-space radius)
    uint4 info[N];    // {code, Meshlet words 5, 6, 7}; code = row*64+lane | a1<<9 | a2<<10 | visible_in_buffer<<11
};

struct CandState {
    uint32_t head = 0, count = 0; // ring window (wave-uniform)
    uint32_t total = 0;           // survivors of the tile so far
};

__device__ __forceinline__ uint32_t ring_wrap(uint32_t i) { return i >= kCandRing ? i - kCandRing : i; }

// Tests the first n (<= 64) candidates of the ring; lane j takes candidate j.
template <uint32_t N, uint32_t CAP = kPayloadCap>
__device__ __forceinline__ void cand_flush(const MeshletCullParams &p, WaveTileLds &L, const CandLds<N> &C, int lane,
                                           CandState &cs, uint32_t n, Payload *payload) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool valid = (uint32_t)lane < n;
    const uint32_t slot = ring_wrap(cs.head + (uint32_t)lane);
    const float4 sp = C.sphere[slot];
    const uint4 info = C.info[slot];
    bool visible = false;
    if (valid) {
        // the ring holds the MODEL-space radius: the test fuses radius * scale into three of its sums
        const float scale = L.r[(info.x >> 5) & 15u].scale;
        Sphere s;
        s.x = sp.x, s.y = sp.y, s.z = sp.z, s.r = sp.w * scale;
        visible = occlusion_test(p.ci, s, sp.w, scale, p.pyr); // :161-205
    }
    const uint32_t code = info.x;
    const uint32_t abits = (code >> 9) & 3u;
    const bool should_draw = should_draw_of(visible, abits, ((code >> 11) & 1u) != 0, true);
    const uint32_t word = (code >> 5) & 15u, bit = 1u << (code & 31u); // record 2*row + half, lane within the record
    if (visible) atomicOr(reinterpret_cast<uint32_t *>(L.vis_mask) + word, bit);
    if (should_draw) atomicOr(reinterpret_cast<uint32_t *>(L.draw_mask) + word, bit);
    const uint64_t dm = __ballot(should_draw);
    if (payload && should_draw) {
        const uint32_t rank = cs.total + lane_prefix(dm);
        if (rank < CAP)
            payload[rank] = make_payload(code & 31u, word, info);
    }
    cs.total += (uint32_t)__popcll(dm);
    cs.head = ring_wrap(cs.head + n);
    cs.count -= n;
}

// One row of a tile in pass 2 with meshlet occlusion: geometry tests, candidates into the ring.
template <int PROJ, int R, uint32_t N, bool CLS = false, uint32_t CAP = kPayloadCap>
__device__ __forceinline__ void row_eval_occ(const MeshletCullParams &p, WaveTileLds &L, const PlaneLds &P,
                                             const AlphaLds *A, CandLds<N> &C, int lane, const RowRegs<1> &t,
                                             CandState &cs, Payload *payload) {
    const uint32_t half = lane >> 5, ml = lane & 31;
    const uint32_t rid = 2 * R + half;
    const uint4 rec = L.r[rid].rec;
    const uint32_t abits = lane_alpha_bits<CLS>(p, L, A, rid, ml, t.b[0].w);
    const bool active = ml < rec.z;
    const bool visible_in_buffer = ((t.prev[0] >> ml) & 1u) != 0;
    Sphere s;
    s.x = s.y = s.z = s.r = 0.0f;
    bool cand = false;
    if (active) cand = eval_geometry<2, PROJ>(p, L, P, rid, t.a[0], t.b[0], visible_in_buffer, s, row_is_affine(L, R));
    const uint64_t mask = __ballot(cand);
    if (cand) {
        const uint32_t slot = ring_wrap(ring_wrap(cs.head + cs.count) + lane_prefix(mask));
        C.sphere[slot] = make_float4(s.x, s.y, s.z, __uint_as_float(t.a[0].w)); // model-space radius (cand_flush)
        C.info[slot] = make_uint4((uint32_t)(R * 64 + lane) | (abits << 9) | ((visible_in_buffer ? 1u : 0u) << 11),
                                  t.b[0].y, t.b[0].z, t.b[0].w);
    }
    cs.count += (uint32_t)__popcll(mask);
    if (cs.count >= kCandFlush) cand_flush<N, CAP>(p, L, C, lane, cs, min(cs.count, 64u), payload);
}

// End of a tile: remaining candidates, then the pass-2 visibility words (:233-254) of its 16 records.
template <uint32_t N, uint32_t CAP = kPayloadCap>
__device__ __forceinline__ void tile_finish_occ(const MeshletCullParams &p, WaveTileLds &L, const CandLds<N> &C,
                                                int lane, CandState &cs, Payload *payload) {
    if (cs.count > 0) cand_flush<N, CAP>(p, L, C, lane, cs, cs.count, payload);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < (int)kTileRecords) {
        const uint4 rec = L.r[lane].rec;
        uint32_t word = reinterpret_cast<const uint32_t *>(L.vis_mask)[lane];
        // task shaders keep `visible = true` in lanes past the count and write it (forward_depth_prepass.task:124,232)
        if (p.task_records && rec.z < 32u) word |= ~0u << rec.z;
        if (rec.z > 0) p.meshlet_visibility[rec.w] = word;
    }
}

// Tile setup, split so that each step's loads can be issued well before they are consumed (see the
// pipelines of the kernels): record load, model-column load, (view x model) product + LDS slab write.
__device__ __forceinline__ uint4 setup_load_rec(const MeshletCullParams &p, uint32_t tile, bool tile_valid,
                                                uint32_t nrec, int lane) {
    const OrbitMeshletDispatch *recs =
        reinterpret_cast<const OrbitMeshletDispatch *>(p.dispatch_buffer + ORBIT_DISPATCH_HEADER);
    const uint32_t rec_index = tile * kTileRecords + (uint32_t)(lane >> 2);
    // unconditional load (exact vmcnt accounting): lanes without a record read the zero page
    const bool has = tile_valid && rec_index < nrec;
    const uint4 *src = has ? reinterpret_cast<const uint4 *>(recs + rec_index)
                           : reinterpret_cast<const uint4 *>(p.zero_page);
    return *src;
}

__device__ __forceinline__ float4 setup_load_mat(const MeshletCullParams &p, const uint4 &rec, int lane) {
    // meshlet_count == 0 marks "no record" (zero page above): read zeros instead of entity 0
#ifdef ORBIT_TRIAGE_NO_MAT
    return *reinterpret_cast<const float4 *>(p.zero_page);
#endif
    const float4 *src = rec.z != 0u ? reinterpret_cast<const float4 *>(p.entities[rec.x].model_matrix) + (lane & 3)
                                    : reinterpret_cast<const float4 *>(p.zero_page);
    return *src;
}

// Alpha classes of a record's lanes: the four lanes of the record load the two words of each bit plane that cover
// meshlets [meshlet_offset, meshlet_offset + 32) (kernels.h MeshletStreamView).
__device__ __forceinline__ uint32_t setup_load_cls(const MeshletCullParams &p, const uint4 &rec, int lane) {
    const uint32_t col = lane & 3;
    const uint32_t *plane = (col & 2u) ? p.ms.cls1 : p.ms.cls0;
    const uint32_t *src = (rec.z != 0u && p.ms.covers(rec.y)) ? plane + (rec.y >> 5) + (col & 1u)
                                                              : reinterpret_cast<const uint32_t *>(p.zero_page);
    return *src;
}

// -> the record's predicate masks in the slab: bit l of amask.x = (1 << alpha_mode) & alpha_mode_flag != 0 (:207) for
// lane l, of amask.y the same for noskip_alphamode (:210).  The library takes this path only for streams WITHOUT
// meshlets of class 3 (abi.hip): nothing here ever looks a material up.
__device__ __forceinline__ void setup_write_cls(WaveTileLds &L, const uint32_t *sel, const uint4 &rec, uint32_t cls_word,
                                                int lane) {
    const uint32_t w0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cls_word, 0x00, 0xF, 0xF, false); // quad_perm broadcasts
    const uint32_t w1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cls_word, 0x55, 0xF, 0xF, false);
    const uint32_t w2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cls_word, 0xAA, 0xF, 0xF, false);
    const uint32_t w3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cls_word, 0xFF, 0xF, 0xF, false);
    const uint32_t sh = rec.y & 31u;
    const uint32_t lo = __funnelshift_r(w0, w1, sh), hi = __funnelshift_r(w2, w3, sh);
    const uint32_t c0 = ~(hi | lo), c1 = lo & ~hi, c2 = hi & ~lo;
    // sel[2 c + k]: all ones if class c has predicate bit k (block-wide table: keeps the cull info's words out of the
    // scalar registers the rows need)
    const uint32_t exist = rec.z >= 32u ? ~0u : ~(~0u << (rec.z & 31u)); // lanes below meshlet_count (:111)
    if ((lane & 3) == 0)
        L.r[lane >> 2].amask = make_uint2(((c0 & sel[0]) | (c1 & sel[2]) | (c2 & sel[4])) & exist,
                                          ((c0 & sel[1]) | (c1 & sel[3]) | (c2 & sel[5])) & exist);
}

__device__ __forceinline__ void setup_write(const MeshletCullParams &p, WaveTileLds &L, const uint4 &rec,
                                            const float4 &mc, int lane, uint32_t prev_word = 0u) {
    const uint32_t rid = lane >> 2, col = lane & 3;
    float o[4];
    mat4_mul_col(p.ci.view_matrix, mc.x, mc.y, mc.z, mc.w, o);
    L.r[rid].mcol[col] = make_float4(o[0], o[1], o[2], o[3]);
    if (col == 3) L.r[rid].tzero = make_float4(o[0] * 0.0f, o[1] * 0.0f, o[2] * 0.0f, 0.0f);
    const float d = dot3(o[0], o[1], o[2], o[0], o[1], o[2]);
    const float d0 = __shfl(d, (int)(rid * 4 + 0), 64);
    const float d1 = __shfl(d, (int)(rid * 4 + 1), 64);
    const float d2 = __shfl(d, (int)(rid * 4 + 2), 64);
    if (col == 0) {
        L.r[rid].scale = sqrtf(gmax(d0, gmax(d1, d2)));
        L.r[rid].rec = rec;
        L.r[rid].prev = prev_word;
    }
    {   // affine_rows (WaveTileLds): a column's w is 0 (1 for the translation, whose xyz must be finite: x * 0 == 0)
        const bool okc = col == 3 ? (o[3] == 1.0f && o[0] * 0.0f == 0.0f && o[1] * 0.0f == 0.0f && o[2] * 0.0f == 0.0f)
                                  : o[3] == 0.0f;
        uint64_t m = __ballot(okc); // four bits per record, eight per row
        m &= m >> 1;
        m &= m >> 2;
        m &= m >> 4; // bit 8 r: all eight
        if (lane == 0) L.affine_rows = m;
    }
    if (lane < (int)kTileRows) L.draw_mask[lane] = 0ull, L.vis_mask[lane] = 0ull; // the candidate flushes OR into them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------
// The record list of the sharded engine (orbit_meshlet_cull_visible_records, visible_list == 2): one 12-B
// {entity_index, meshlet_offset, should-draw ballot} entry per DISPATCH RECORD, in record order, behind an 8-B
// {records, survivors} header — the same ordered list of (meshlet, entity) pairs as the 8-B items (bit b of record r =
// item {meshlet_offset + b, entity_index}; a record without a survivor is an entry with mask 0), at most 3/8 of a
// byte per dispatched meshlet: what travels over xGMI.  An entry's position is its record's index, so the evaluation
// writes the list itself, tile by tile — no count, no scan, no launch behind it (rounds 3-4 compacted the entries of
// records with survivors into a dense list: two more launches per cull, 12 us of an 80-us shard step, to save the 4 % of
// entries that are empty on BASELINE config 5).
// The 16 entries of wave tile `tile` from its slab (records and final ballots): lane = 3 x record + word, one
// contiguous 192-B store; the tile's survivors are added to the wave's sum in LDS (`wave_sum`).
__device__ __forceinline__ void tile_records_out(const MeshletCullParams &p, const WaveTileLds &L, uint32_t nrec, uint32_t tile,
                                                 uint32_t survivors, uint32_t *wave_sum, int lane) {
    // entries past the dispatched records or the list's capacity fall outside the resource and are dropped
    const __amdgpu_buffer_rsrc_t list =
        __builtin_amdgcn_make_buffer_rsrc(p.draw_buffer + ORBIT_VISIBLE_HEADER, 0, min(nrec, p.draw_capacity) * 12u, kBufFlags);
    if (lane == 0) *wave_sum += survivors;
    const uint32_t r = (uint32_t)lane / 3u, k = (uint32_t)lane - 3u * r;
    const uint32_t rr = r < kTileRecords ? r : 0u;
    const uint4 rec = L.r[rr].rec;
    const uint32_t m = reinterpret_cast<const uint32_t *>(L.draw_mask)[rr];
    const uint32_t v = k == 0u ? rec.x : (k == 1u ? rec.y : m);
    const uint32_t off = lane < (int)(3u * kTileRecords) ? tile * (kTileRecords * 12u) + (uint32_t)lane * 4u : kNoOffset;
    __builtin_amdgcn_raw_buffer_store_b32(v, list, off, 0, 0);
}

// End of an evaluation that wrote the record list; every thread of the workgroup calls it (a barrier inside).
// s_red[w] = survivors of the tiles wave w evaluated (tile_records_out).  A workgroup arrives with ONE returning
// 64-bit atomic — arrivals << 40 | survivors — on the word of its group (workgroup b: group b % kListGroups), the last
// one of a group carries the group's sum on to the top word the same way, and the last one there holds the total: it
// writes the header {min(records, capacity), survivors}, latches an overflow, and leaves the words and the tile tickets
// zero for the next launch (a replayed graph finds them as the first launch did).  One atomic on a workgroup's way
// out (two for kListGroups of them), each word reached by 1 / kListGroups of the workgroups.
// (ent_flags / ne_chunks / sync: the shard launch's entity-stage hand-over words, zeroed by the same last workgroup)
template <int WAVES>
__device__ __forceinline__ void records_finish(const MeshletCullParams &p, uint32_t nrec, const uint32_t *s_red,
                                               uint32_t *ent_flags = nullptr, uint32_t ne_chunks = 0u, uint32_t *sync = nullptr,
                                               uint32_t *done_flags = nullptr) {
    // this wave's ticket claims have returned and its list stores have left: nothing of it is in flight behind the arrival
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x != 0) return;
    unsigned long long sum = 0;
#pragma unroll
    for (int w = 0; w < WAVES; w++) sum += s_red[w];
    constexpr unsigned long long kOne = 1ull << 40, kSum = kOne - 1ull;
    unsigned long long *sy = reinterpret_cast<unsigned long long *>(p.list_sync);
    const uint32_t g = blockIdx.x % kListGroups;
    const uint32_t in_group = (gridDim.x - g + kListGroups - 1u) / kListGroups;
    const unsigned long long a = __hip_atomic_fetch_add(sy + g * (kListSyncStride / 2u), kOne | sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(a >> 40) != in_group - 1u) return;
    const unsigned long long group_sum = (a & kSum) + sum;
    const uint32_t groups = min(gridDim.x, kListGroups);
    const unsigned long long t = __hip_atomic_fetch_add(sy + kListGroups * (kListSyncStride / 2u), kOne | group_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(t >> 40) != groups - 1u) return;
    // every workgroup has arrived
    const unsigned long long total = (t & kSum) + group_sum;
    for (uint32_t k = 0; k <= kListGroups; k++) sy[k * (kListSyncStride / 2u)] = 0ull;
    uint32_t *hdr = reinterpret_cast<uint32_t *>(p.draw_buffer);
    hdr[0] = min(nrec, p.draw_capacity);
    hdr[1] = (uint32_t)total;
    *p.total = nrec;
    if (nrec > p.draw_capacity) latch_status(p.status, ORBIT_E_CAPACITY);
    for (uint32_t q = 0; q < kTicketPools; q++) p.tickets[q * kTicketStride] = 0u; // the next evaluation's tile tickets
    if (ent_flags != nullptr) {
        for (uint32_t i = 0; i < ne_chunks; i++) ent_flags[i] = 0u;
        for (uint32_t i = 0; i < kFusedSyncWords; i++) sync[i] = 0u;
        for (uint32_t k = 0; k < kShardDoneWords; k++) done_flags[k * kShardDoneStride] = 0u, done_flags[k * kShardDoneStride + 1u] = 0u;
    }
}

// Triage switches (tools/: wave_tail.py, ab_inproc.py, phase.py) exist only in -DORBIT_TRIAGE builds
// (tools/mkvariant_any.sh); the product library's kernels take no debug input of any kind.
#ifdef ORBIT_TRIAGE
#define ORBIT_DBG(p, bits) (((p).debug_flags & (bits)) != 0u)
#else
#define ORBIT_DBG(p, bits) false
#endif

} // namespace

} // namespace orbit
