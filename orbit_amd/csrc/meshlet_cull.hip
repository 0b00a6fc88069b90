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
// Tile bases come from a prefix sum over per-tile survivor counts: eval launch
// -> scan -> emit launch.  (A single-launch variant with a decoupled look-back
// was measured slower on MI355X and removed: profiles/r01_notes.md.)
//
//   pass 0 / 2 : meshlet_eval_kernel streams every meshlet of every record; in
//                pass 2 the HiZ test runs on candidates compacted in LDS.
//   pass 1     : meshlet_pass1_kernel is list-driven — only meshlets whose
//                last-frame bit is set are read at all.
//   streams    : with derived meshlet streams bound (orbit_meshlet_stream_*; kernels.h
//                MeshletStreamView) passes 0 / 2 read 22 B per meshlet instead of
//                the 32-B Meshlet and keep no survivor payload; the emit launch
//                gathers the survivors' 12 command bytes (template flag STREAM).
//   outputs    : MeshletDrawCommandBuffer (meshlet_emit_kernel<false>), the
//                sharded engine's 8-B visible list (<true>) with
//                visible_expand_kernel on the receiving side, or per-record
//                task payloads for the mesh-shading path (task_records_kernel).
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
    const float t1 = k.x * s.x, t2 = k.y * s.z, t3 = k.z * s.y, t4 = k.w * s.z, t5 = n.x * s.z;
    const float nr = -s.r;
    uint64_t in = ballot((t1 + t2) > nr);
    in &= ballot((t2 - t1) > nr);
    in &= ballot((t3 + t4) > nr);
    in &= ballot((t4 - t3) > nr);
    in &= ballot((t5 + n.y) > nr);
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

// One meshlet (one lane): meshlet_cull.comp:113-213.  PASS = cull_info.occlusion_pass
// and PROJ = cull_info.projection_type are compile-time (PROJ < 0: read at run time).
// Geometry part (:117-158): view-space sphere `s`, pass-1 gate, plane loop, cone test.
// `affine_row` (wave-uniform): both records of the row have an affine matrix with a finite translation (WaveTileLds::
// affine_rows).  Then a FINITE centre has w = ((0 cx + 0 cy) + 0 cz) + 1 = 1 exactly — no w, no division — and the
// w = 0 term of the cone-axis transform is +-0, which no comparison sees; and a finite view-space centre is what the
// shortened plane tests of the symmetric frustum need (planes_std).  One non-finite lane (its products with 0 are NaN
// in the reference's arithmetic) sends the whole row down the literal path.
template <int PROJ>
__device__ __forceinline__ uint64_t eval_geometry_mask(const MeshletCullParams &p, const WaveTileLds &L,
                                                       const PlaneLds &P, uint32_t rid, const uint4 &a, const uint4 &b,
                                                       Sphere &s, bool affine_row = false) {
    const OrbitGpuCullInfo &ci = p.ci;
    const RecordLds &R = L.r[rid];
    const float4 m0 = R.mcol[0], m1 = R.mcol[1], m2 = R.mcol[2], m3 = R.mcol[3];
    const float cx = __uint_as_float(a.x), cy = __uint_as_float(a.y), cz = __uint_as_float(a.z);
    // transform_sphere (:37-43): matrix * vec4(c, 1), divide by w.
    float px = ((m0.x * cx + m1.x * cy) + m2.x * cz) + m3.x * 1.0f;
    float py = ((m0.y * cx + m1.y * cy) + m2.y * cz) + m3.y * 1.0f;
    float pz = ((m0.z * cx + m1.z * cy) + m2.z * cz) + m3.z * 1.0f;
    // (a non-finite centre makes all three non-finite: each holds a product with it)
    const uint64_t nonfinite = ballot(!__builtin_isfinite(px)) | ballot(!__builtin_isfinite(py)) | ballot(!__builtin_isfinite(pz));
    // wave-uniform and held as a scalar integer (not a lane boolean)
    const uint32_t literal = (affine_row ? 0u : 1u) | (uint32_t)nonfinite | (uint32_t)(nonfinite >> 32);
    if (literal != 0u) {
        const float pw = ((m0.w * cx + m1.w * cy) + m2.w * cz) + m3.w * 1.0f;
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
    // without it the class variant of the evaluation holds 128 registers and spills (profiles/r03_notes.md)
    __builtin_amdgcn_sched_barrier(0);
    // The cone test (:148-158) does not depend on the plane loop: both are evaluated
    // and combined, which gives the scheduler two independent chains per meshlet.
    uint64_t cull;
    {
        // cone axis / cutoff (:121-122): int(i8) / 127.0, axis = (M * vec4(a, 0)).xyz
        const int ax_i = (int)(int8_t)(b.x & 0xFF), ay_i = (int)(int8_t)((b.x >> 8) & 0xFF);
        const int az_i = (int)(int8_t)((b.x >> 16) & 0xFF), co_i = (int)(int8_t)(b.x >> 24);
        const float ax = snorm8(ax_i), ay = snorm8(ay_i), az = snorm8(az_i);
        const float cutoff = snorm8(co_i);
        float axx = (m0.x * ax + m1.x * ay) + m2.x * az;
        float axy = (m0.y * ax + m1.y * ay) + m2.y * az;
        float axz = (m0.z * ax + m1.z * ay) + m2.z * az;
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
template <int PASS, int PROJ, int R0, int NR, bool CLS = false>
__device__ __forceinline__ uint32_t rows_eval(const MeshletCullParams &p, WaveTileLds &L, const PlaneLds &P,
                                              const AlphaLds *A, int lane, const RowRegs<NR> &t, uint32_t total,
                                              Payload *payload) {
    static_assert(PASS == 0 || PASS == 2, "pass 1 is list-driven (meshlet_pass1_body)");
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
#ifdef ORBIT_TRIAGE_NO_EVAL // scratch builds only (tools/mkvariant_any.sh): a free predicate with the same survival ratio
        const uint64_t geo = ballot(((t.a[k].x + (t.b[k].x >> 31)) % 1000u) < 95u);
#else
        Sphere s;
        const uint64_t geo = eval_geometry_mask<PROJ>(p, L, P, rid, t.a[k], t.b[k], s, row_is_affine(L, r));
#endif
        const uint64_t mask = geo & allow;
        if (lane == 0) L.draw_mask[r] = mask;
        if (payload && lane_of(mask)) {
            // survivor payload at the tile-local rank: everything the command needs except its
            // slot, so the emit launch reads one contiguous run per tile and rebuilds nothing
            const uint32_t rank = total + lane_prefix(mask);
            if (rank < kPayloadCap) payload[rank] = make_payload(ml, rid, t.b[k]);
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
template <uint32_t N>
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
        if (rank < kPayloadCap)
            payload[rank] = make_payload(code & 31u, word, info);
    }
    cs.total += (uint32_t)__popcll(dm);
    cs.head = ring_wrap(cs.head + n);
    cs.count -= n;
}

// One row of a tile in pass 2 with meshlet occlusion: geometry tests, candidates into the ring.
template <int PROJ, int R, uint32_t N, bool CLS = false>
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
    if (cs.count >= kCandFlush) cand_flush(p, L, C, lane, cs, min(cs.count, 64u), payload);
}

// End of a tile: remaining candidates, then the pass-2 visibility words (:233-254) of its 16 records.
template <uint32_t N>
__device__ __forceinline__ void tile_finish_occ(const MeshletCullParams &p, WaveTileLds &L, const CandLds<N> &C,
                                                int lane, CandState &cs, Payload *payload) {
    if (cs.count > 0) cand_flush(p, L, C, lane, cs, cs.count, payload);
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
// eval launch -> scan -> emit launch.
//
//   meshlet_eval : streams records, matrices and meshlets through the software
//                  pipeline below and writes, per wave tile, the 8 should-draw
//                  ballots (64 B) and their popcount.  No inter-wave dependency
//                  of any kind: tiles are assigned statically (grid stride), so
//                  waves drift out of phase and one wave's evaluation overlaps
//                  another's loads.  Pass-2 visibility words are written here.
//   scan         : exclusive prefix over the tile counts (scan.h / entity_cull.hip).
//   meshlet_emit : per tile, re-reads only the survivors' command words and
//                  writes the 28-B commands at the scanned offsets.
//
// Software pipeline of one wave in meshlet_eval.  A tile's 8 rows go through two
// single-row register sets A/B, so that one row's meshlet loads are in flight
// during every evaluation (16 waves per CU x 2 KiB is more than the bytes in
// flight that HBM latency x bandwidth asks of a CU):
//     eval row 0 (A) ; load A <- row 2      eval row 1 (B) ; load B <- row 3
//     eval row 2 (A) ; load A <- row 4      eval row 3 (B) ; load B <- row 5
//     issue matrix loads (i+2), record load (i+3)      | small, one iteration ahead
//     eval row 4 (A) ; load A <- row 6      eval row 5 (B) ; load B <- row 7
//     eval row 6 (A) ; load A <- row 0 of i+1   eval row 7 (B) ; load B <- row 1 of i+1
//     write slab(i+2) ; store masks(i), count(i)
// The dependent chain record -> matrix -> meshlets of a tile is spread over three
// iterations; all loads are unconditional (inactive lanes read a zero page) so
// that the compiler's vmcnt accounting stays exact.
constexpr int kEvWaves = 4;
constexpr int kEvWavesPerSimd = 4; // resident waves per SIMD: compiled for (128 VGPRs) and launched at
// Pass 0 from the derived streams needs fewer registers than the passes with the HiZ test (84 against 107-119) and is
// compiled for / launched at five waves per SIMD: with the rows as nt loads the fifth wave is worth 3.5 % (0.1948 /
// 0.1964 -> 0.1879 / 0.1897 ms; six: 0.1897 / 0.1909, three: 0.2175 / 0.2184) — with the default cache policy it was
// worth nothing, and from the Meshlet buffer it costs 1.5 % (0.3470 -> 0.3525; r03_notes.md).
#ifndef ORBIT_EV_WPS0
#define ORBIT_EV_WPS0 5
#endif
constexpr int ev_waves_per_simd(int pass, bool stream) { return (pass == 0 && stream) ? ORBIT_EV_WPS0 : kEvWavesPerSimd; }

// Views of one launch of the *_views kernels (orbit_cull_views): blockIdx.y = slot, idx[slot] = entry of the table.
struct ViewGroup {
    uint8_t idx[ORBIT_MAX_CULL_VIEWS];
};

// Triage switches (tools/: wave_tail.py, ab_inproc.py, phase.py) exist only in -DORBIT_TRIAGE builds of this file
// (tools/mkvariant_any.sh); the product library's kernels take no debug input of any kind.
#ifdef ORBIT_TRIAGE
#define ORBIT_DBG(p, bits) (((p).debug_flags & (bits)) != 0u)
#else
#define ORBIT_DBG(p, bits) false
#endif

// STREAM: meshlets from the derived streams; CLS (with STREAM): predicate bits from the streams' alpha classes
template <int PASS, int PROJ, bool PROF, bool STREAM = false, bool CLS = false>
__device__ __forceinline__ void meshlet_eval_body(const MeshletCullParams &p) {
    __shared__ WaveTileLds lds[kEvWaves][3];
    __shared__ PlaneLds planes;
    __shared__ AlphaLds alpha_tab;
    __shared__ CandLds<(PASS == 2 ? kCandRing : 1u)> cand_ring[kEvWaves]; // pass 2 only
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // pass 2 with meshlet occlusion: HiZ tests run on compacted candidates (cand_flush), not per row
    const bool occ2 = PASS == 2 && p.ci.meshlet_visibility_buffer != ORBIT_NONE;
    __shared__ uint32_t cls_sel[CLS ? 8 : 1]; // [2 c + k] = all ones if alpha class c has predicate bit k (alpha_bits)
    planes_to_lds(p, planes);
    if (CLS && threadIdx.x < 6) cls_sel[threadIdx.x] = ((alpha_bits(p.ci, threadIdx.x >> 1) >> (threadIdx.x & 1u)) & 1u) ? ~0u : 0u;
    const AlphaLds *alpha = alpha_table_fill(p, alpha_tab) ? &alpha_tab : nullptr;
    __syncthreads();
    const uint32_t nrec = min(*reinterpret_cast<const uint32_t *>(p.dispatch_buffer), p.dispatch_capacity);
    const uint32_t ntiles = (nrec + kTileRecords - 1) / kTileRecords;
    // Tile assignment: a static grid-stride share first, tickets for the rest.  With 23 or 24 tiles per wave handed out
    // statically the waves finish over a span of ~45 us (the SIMD arbiter favours older waves, CUs differ) and the last
    // 6 % of the kernel run at falling occupancy (tools/wave_tail.py); handing every tile out through an atomic ticket
    // removes the tail but costs more than it saves (+5 %: the returning atomic sits in the in-order vector-memory
    // queue in front of the row loads, once per tile).  So every wave takes its first S tiles statically (tile
    // k * stride + g, as before) and only the last ~3 tiles' worth per wave go through tickets — 3-4 atomics per wave:
    // fast waves draw more of them.  One counter cannot serve the ~290 claims/us at the very end (a returning atomic
    // on one address saturates near 88/us on MI355X), so there are kTicketPools: wave g draws from pool g % npools,
    // ticket n of pool q is tile S * stride + n * npools + q; every pool is drained by 1/npools of the waves of all
    // XCDs, so the pools run dry within a tile's time of each other and nobody steals.  A claim is made three tiles
    // before its rows are read (records -> matrices -> slab -> rows); a wave whose ticket is past the end drains its
    // pipeline and leaves.
    const uint32_t stride = gridDim.x * kEvWaves, wave_g = blockIdx.x * kEvWaves + wave;
    const bool static_share = ORBIT_DBG(p, 64u); // triage: everything static, for A/B runs
    const uint32_t full_rounds = ntiles / stride;
    // ticketed rounds: three on a long launch, fewer on a short one (a shard of the sharded engine has 2-3 rounds in all)
    const uint32_t dyn_rounds = ORBIT_DBG(p, 256u) ? 2u : ORBIT_DBG(p, 512u) ? 4u : ORBIT_DBG(p, 1024u) ? 1u : min(max(full_rounds / 4u, 1u), 3u);
    // a scene of at most one tile per wave has nothing to balance: static, no atomic in front of the first loads
    // ... and the three tiles a wave claims before its first row never go through a ticket either: their records,
    // matrices and rows are the launch's ramp, and an atomic in front of them is one more round trip before anything
    // streams (a 1/8 shard of config 5 is three tiles per wave)
    const uint32_t n_static = (static_share || ntiles <= stride) ? 0xFFFFFFFFu : (ORBIT_DBG(p, 128u) ? 0u : max(full_rounds >= dyn_rounds ? full_rounds - dyn_rounds : 0u, 3u));
    const uint32_t npools = min((uint32_t)kTicketPools, stride);
    const uint32_t pool = wave_g % npools;
    uint32_t *ticket_ctr = p.tickets + pool * kTicketStride;
    uint32_t claims = 0; // claims made so far by this wave
    auto claim = [&]() -> uint32_t { // lane 0's VGPR carries the claim; not read before ticket_tile()
        uint32_t raw = claims;
        if (claims >= n_static) {
            raw = 0;
            if (lane == 0) raw = atomicAdd(ticket_ctr, 1u);
        }
        claims++;
        return raw;
    };
    auto ticket_tile = [&](uint32_t raw, uint32_t k) -> uint32_t { // k = index of the claim that returned `raw`
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)raw);
        if (k < n_static) return t * stride + wave_g;
        return t >= 0x08000000u ? 0xFFFFFFFFu : n_static * stride + t * npools + pool;
    };
    StreamRsrc SR = {};
    if constexpr (STREAM) {
        SR.first = p.ms.first;
        SR.count = p.ms.count;
        SR.sphere = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(p.ms.sphere + p.ms.first), 0, p.ms.count * 16u, kBufFlags);
        SR.cone = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.ms.cone + p.ms.first), 0, p.ms.count * 4u, kBufFlags);
        if constexpr (!CLS)
            SR.mat = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.ms.mat + p.ms.first), 0, p.ms.count * 2u, kBufFlags);
    }
    uint32_t w0, w1, w2, tk_raw;
    {
        const uint32_t t0 = claim(), t1 = claim(), t2 = claim();
        tk_raw = claim();
        w0 = ticket_tile(t0, 0u);
        w1 = ticket_tile(t1, 1u);
        w2 = ticket_tile(t2, 2u);
    }
    // two single-row register sets: while one row is evaluated the next row's loads are in flight
    // (STREAM: a row is 1.25 KiB instead of 2 KiB; sets of two rows each were measured in rounds 2 and 3 — neutral or
    // slower, as are 5 and 6 resident waves per SIMD: profiles/r02_notes.md, r03_notes.md)
    RowRegs<1> qa, qb;
    uint4 rec2;
    {
        const uint4 r0 = setup_load_rec(p, w0, w0 < ntiles, nrec, lane);
        const uint4 r1 = setup_load_rec(p, w1, w1 < ntiles, nrec, lane);
        rec2 = setup_load_rec(p, w2, w2 < ntiles, nrec, lane);
        const float4 m0 = setup_load_mat(p, r0, lane);
        const float4 m1 = setup_load_mat(p, r1, lane);
        if constexpr (CLS) {
            const uint32_t c0 = setup_load_cls(p, r0, lane), c1 = setup_load_cls(p, r1, lane);
            setup_write_cls(lds[wave][0], cls_sel, r0, c0, lane);
            setup_write_cls(lds[wave][1], cls_sel, r1, c1, lane);
        }
        setup_write(p, lds[wave][0], r0, m0, lane);
        setup_write(p, lds[wave][1], r1, m1, lane);
        rows_load<PASS, 0, 1, STREAM, CLS>(p, lds[wave][0], lane, qa, &SR);
        rows_load<PASS, 1, 1, STREAM, CLS>(p, lds[wave][0], lane, qb, &SR);
    }
    uint32_t it = 0;
#ifdef ORBIT_TRIAGE
    unsigned long long tc[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, t1;
#define EV_STAMP(k)                            \
    if (PROF && !(p.debug_flags & 16u)) {      \
        t1 = __builtin_amdgcn_s_memtime();     \
        __builtin_amdgcn_s_waitcnt(0xC07F);    \
        tc[k] += t1 - t0;                      \
        t0 = t1;                               \
    }
    unsigned long long wall_begin = 0;
    if (p.debug_flags & 16u) wall_begin = wall_clock64(); // per-wave life stamps (tools/wave_tail.py)
    if (PROF) {
        t0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
#else
#define EV_STAMP(k)
    static_assert(!PROF, "phase stamps: -DORBIT_TRIAGE builds only");
#endif
    // The SIMD's arbiter serves the oldest wave first, and the four waves of a SIMD belong to the four blocks a CU
    // holds, dispatched one after the other: left alone, the first block's waves finish at half the kernel's
    // duration and the rest of the kernel runs at 3, 2, 1 waves per SIMD (tools/wave_tail.py).  Every wave
    // therefore rotates its priority once per tile, offset by its block's dispatch rank, so that over any four
    // tiles each wave of a SIMD has held each priority once.
    const uint32_t prio_rank = blockIdx.x / max(gridDim.x / (uint32_t)ev_waves_per_simd(PASS, STREAM), 1u);
    while (w0 < ntiles) {
        if (!ORBIT_DBG(p, 32u)) {
            switch ((it + prio_rank) & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
            }
        }
        const uint32_t w3 = ticket_tile(tk_raw, claims - 1u); // claimed one tile ago
        tk_raw = claim();                                     // the tile after it
        WaveTileLds &L = lds[wave][it % 3];
        WaveTileLds &Ln = lds[wave][(it + 1) % 3]; // slab of an absent tile holds zero records
        // no payload with the derived streams: the emit launch derives the survivors' command words itself
        Payload *payload = (STREAM || p.task_records || p.visible_list == 2u) ? nullptr : p.tile_payload + (size_t)w0 * kPayloadCap;
        CandState cs; // cs.total = the tile's running survivor count
#define EV_ROW(R, Q)                                                                                                  \
    if constexpr (PASS == 2) {                                                                                        \
        if (occ2) row_eval_occ<PROJ, R, kCandRing, CLS>(p, L, planes, alpha, cand_ring[wave], lane, Q, cs, payload);  \
        else cs.total = rows_eval<PASS, PROJ, R, 1, CLS>(p, L, planes, alpha, lane, Q, cs.total, payload);      \
    } else {                                                                                                          \
        cs.total = rows_eval<PASS, PROJ, R, 1, CLS>(p, L, planes, alpha, lane, Q, cs.total, payload);           \
    }
        EV_ROW(0, qa)
        rows_load<PASS, 2, 1, STREAM, CLS>(p, L, lane, qa, &SR);
        EV_ROW(1, qb)
        rows_load<PASS, 3, 1, STREAM, CLS>(p, L, lane, qb, &SR);
        EV_STAMP(0)
        EV_ROW(2, qa)
        rows_load<PASS, 4, 1, STREAM, CLS>(p, L, lane, qa, &SR);
        EV_ROW(3, qb)
        rows_load<PASS, 5, 1, STREAM, CLS>(p, L, lane, qb, &SR);
        const float4 mat2 = setup_load_mat(p, rec2, lane);
        uint32_t cls2 = 0u;
        if constexpr (CLS) cls2 = setup_load_cls(p, rec2, lane);
        const uint4 rec3 = setup_load_rec(p, w3, w3 < ntiles, nrec, lane);
        EV_STAMP(1)
        EV_ROW(4, qa)
        rows_load<PASS, 6, 1, STREAM, CLS>(p, L, lane, qa, &SR);
        EV_ROW(5, qb)
        rows_load<PASS, 7, 1, STREAM, CLS>(p, L, lane, qb, &SR);
        EV_STAMP(2)
        EV_ROW(6, qa)
        rows_load<PASS, 0, 1, STREAM, CLS>(p, Ln, lane, qa, &SR);
        EV_ROW(7, qb)
        rows_load<PASS, 1, 1, STREAM, CLS>(p, Ln, lane, qb, &SR);
        EV_STAMP(3)
        if constexpr (PASS == 2) {
            if (occ2) tile_finish_occ(p, L, cand_ring[wave], lane, cs, payload);
        }
#undef EV_ROW
        if constexpr (CLS) setup_write_cls(lds[wave][(it + 2) % 3], cls_sel, rec2, cls2, lane);
        setup_write(p, lds[wave][(it + 2) % 3], rec2, mat2, lane); // also fences the mask writes of this tile
        // the ballots are read back only by the task-record launch and, in the emit launch, for tiles with more
        // survivors than the payload holds: the common tile writes its count and nothing else
        if ((STREAM || p.task_records != nullptr || p.visible_list == 2u || cs.total > kPayloadCap) && lane < 2 * (int)kTileRows)
            p.tile_masks[(size_t)w0 * (2 * kTileRows) + lane] = reinterpret_cast<const uint32_t *>(L.draw_mask)[lane];
        if (lane == 0) p.tile_counts[w0] = cs.total;
        EV_STAMP(4)
        w0 = w1;
        w1 = w2;
        w2 = w3;
        rec2 = rec3;
        it++;
    }
#undef EV_STAMP
#ifdef ORBIT_TRIAGE
    if (PROF && lane == 0) {
        for (int k = 0; k < 5; k++) atomicAdd(p.debug_cycles + k, tc[k]);
        atomicAdd(p.debug_cycles + 7, 1ull);
    }
    if ((p.debug_flags & 16u) && lane == 0) {
        const uint32_t wg = blockIdx.x * kEvWaves + wave;
        if (wg < 8192u) {
            p.debug_cycles[32 + 2 * wg] = wall_begin;
            p.debug_cycles[32 + 2 * wg + 1] = wall_clock64();
        }
    }
#endif
}

template <int PASS, int PROJ, bool PROF, bool STREAM = false, bool CLS = false>
__global__ __launch_bounds__(kEvWaves * 64, ev_waves_per_simd(PASS, STREAM)) void meshlet_eval_kernel(const MeshletCullParams p) {
    meshlet_eval_body<PASS, PROJ, PROF, STREAM, CLS>(p);
}
// several views in one launch (orbit_cull_views): the views' parameter blocks by value, blockIdx.y picks one
template <int PASS, int PROJ, bool STREAM, bool CLS = false>
__global__ __launch_bounds__(kEvWaves * 64, ev_waves_per_simd(PASS, STREAM)) void meshlet_eval_views_kernel(const MeshletCullViews all,
                                                                                           const ViewGroup g) {
    meshlet_eval_body<PASS, PROJ, false, STREAM, CLS>(all.v[g.idx[blockIdx.y]]);
}

// ---------------------------------------------------------------------------
// Pass 1 (occlusion_pass == 1, :137): only meshlets that were visible last frame
// are processed at all, and which those are is known from 4 bytes per record
// before any meshlet is read.  So pass 1 is list-driven instead of streaming:
// per wave tile the set bits of the 16 visibility words become a candidate list
// in canonical (row, lane) order (LDS), and the tile is evaluated in chunks of
// 64 candidates — gathered 32-B meshlet reads for the candidates only (dense
// runs coalesce exactly like the streamed rows), full lanes in the evaluation.
// At 10 % visibility that is ~40 % of the streamed bytes and one evaluation per
// tile instead of eight.  Outputs (ballots, count, payload) and the scan + emit
// that follow are those of the other passes.
//
// Pipeline of one wave: records 3 tiles ahead, matrices + visibility words 2
// ahead, slab 1 ahead (as in meshlet_eval); the work items are (tile, chunk)
// pairs through two register sets — while item k is evaluated the gathers of
// item k+1 (next chunk, or chunk 0 of the next tile, whose list is built first)
// are in flight.
struct Pass1Item {
    uint4 a, b;    // the candidate's Meshlet
    uint32_t code; // row*64 + lane of the candidate, ~0u = none
};

__device__ __forceinline__ uint32_t pass1_load_prev(const MeshletCullParams &p, const uint4 &rec, int lane) {
    const bool rw_vis = p.ci.meshlet_visibility_buffer != ORBIT_NONE;
    const uint32_t *src = (rw_vis && rec.z != 0u && (lane & 3) == 0) ? p.meshlet_visibility + rec.w
                                                                    : reinterpret_cast<const uint32_t *>(p.zero_page);
    const uint32_t w = *src;
    return rw_vis ? w : 0xFFFFFFFFu; // no meshlet visibility buffer: visible_in_buffer stays true (:128)
}

// Candidate list of a tile: lanes whose last-frame bit is set (:129-137), canonical order.
__device__ __forceinline__ uint32_t pass1_build_list(const WaveTileLds &L, uint16_t *list, int lane) {
    const uint32_t half = lane >> 5, ml = lane & 31;
    uint32_t n = 0;
#pragma unroll
    for (int r = 0; r < (int)kTileRows; r++) {
        const uint32_t rid = 2 * r + half;
        const bool c = ml < L.r[rid].rec.z && ((L.r[rid].prev >> ml) & 1u) != 0;
        const uint64_t mask = __ballot(c);
        if (c) list[n + lane_prefix(mask)] = (uint16_t)(r * 64 + lane);
        n += (uint32_t)__popcll(mask);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return n;
}

__device__ __forceinline__ Pass1Item pass1_load(const MeshletCullParams &p, const WaveTileLds &L,
                                                const uint16_t *list, uint32_t n, uint32_t chunk, int lane) {
    Pass1Item it;
    const uint32_t j = chunk * 64u + (uint32_t)lane;
    const bool valid = j < n;
    it.code = valid ? (uint32_t)list[j] : ~0u;
    const uint32_t rid = (it.code >> 5) & 15u, ml = it.code & 31u;
    // unconditional loads (exact vmcnt accounting): lanes without a candidate read the zero page
    const uint4 *src = valid ? reinterpret_cast<const uint4 *>(p.meshlets + (L.r[rid].rec.y + ml))
                             : reinterpret_cast<const uint4 *>(p.zero_page);
    it.a = src[0];
    it.b = src[1];
    return it;
}

template <int PROJ>
__device__ __forceinline__ uint32_t pass1_eval(const MeshletCullParams &p, WaveTileLds &L, const PlaneLds &P,
                                               const AlphaLds *A, int lane, const Pass1Item &it, uint32_t total,
                                               Payload *payload) {
    const bool valid = it.code != ~0u;
    const uint32_t rid = (it.code >> 5) & 15u, ml = it.code & 31u;
    const uint32_t mi = it.b.w & 0xFFFFu; // zero page for lanes without a candidate: material 0, unused
    const uint32_t abits = A ? (uint32_t)A->bits[mi & (kAlphaTable - 1u)] : alpha_bits(p.ci, p.materials[mi].alpha_mode);
    bool should_draw = false;
    if (valid) {
        Sphere s;
        const bool visible = eval_geometry<1, PROJ>(p, L, P, rid, it.a, it.b, true, s); // candidates have the bit set
        should_draw = should_draw_of(visible, abits, true, false);
    }
    if (should_draw) atomicOr(reinterpret_cast<uint32_t *>(L.draw_mask) + rid, 1u << ml);
    const uint64_t dm = __ballot(should_draw);
    if (payload && should_draw) {
        const uint32_t rank = total + lane_prefix(dm);
        if (rank < kPayloadCap) payload[rank] = make_payload(ml, rid, it.b);
    }
    return total + (uint32_t)__popcll(dm);
}

template <int PROJ>
__device__ __forceinline__ void meshlet_pass1_body(const MeshletCullParams &p) {
    __shared__ WaveTileLds lds[kEvWaves][3];
    __shared__ PlaneLds planes;
    __shared__ AlphaLds alpha_tab;
    __shared__ uint16_t s_list[kEvWaves][2][kTileRows * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    planes_to_lds(p, planes);
    const AlphaLds *alpha = alpha_table_fill(p, alpha_tab) ? &alpha_tab : nullptr;
    __syncthreads();
    const uint32_t nrec = min(*reinterpret_cast<const uint32_t *>(p.dispatch_buffer), p.dispatch_capacity);
    const uint32_t ntiles = (nrec + kTileRecords - 1) / kTileRecords;
    const uint32_t stride = gridDim.x * kEvWaves;
    uint32_t w0 = blockIdx.x * kEvWaves + wave, w1 = w0 + stride, w2 = w1 + stride;
    uint16_t *list_cur = s_list[wave][0], *list_next = s_list[wave][1];
    uint4 rec2;
    uint32_t n_cur;
    Pass1Item cur;
    {
        const uint4 r0 = setup_load_rec(p, w0, w0 < ntiles, nrec, lane);
        const uint4 r1 = setup_load_rec(p, w1, w1 < ntiles, nrec, lane);
        rec2 = setup_load_rec(p, w2, w2 < ntiles, nrec, lane);
        const float4 m0 = setup_load_mat(p, r0, lane);
        const uint32_t p0 = pass1_load_prev(p, r0, lane);
        const float4 m1 = setup_load_mat(p, r1, lane);
        const uint32_t p1 = pass1_load_prev(p, r1, lane);
        setup_write(p, lds[wave][0], r0, m0, lane, p0);
        setup_write(p, lds[wave][1], r1, m1, lane, p1);
        n_cur = pass1_build_list(lds[wave][0], list_cur, lane); // absent tile: zero records, empty list
        cur = pass1_load(p, lds[wave][0], list_cur, n_cur, 0u, lane);
    }
    uint32_t it = 0;
    while (w0 < ntiles) {
        const uint32_t w3 = w2 + stride;
        WaveTileLds &L = lds[wave][it % 3];
        WaveTileLds &Ln = lds[wave][(it + 1) % 3]; // slab of an absent tile holds zero records
        Payload *payload = p.task_records ? nullptr : p.tile_payload + (size_t)w0 * kPayloadCap;
        const float4 mat2 = setup_load_mat(p, rec2, lane);
        const uint32_t prev2 = pass1_load_prev(p, rec2, lane);
        const uint4 rec3 = setup_load_rec(p, w3, w3 < ntiles, nrec, lane);
        uint32_t total = 0, n_next = 0;
        for (uint32_t chunk = 0;; chunk++) { // at least one trip: an empty tile still hands over to the next one
            const bool last = (chunk + 1u) * 64u >= n_cur;
            Pass1Item next;
            if (!last) {
                next = pass1_load(p, L, list_cur, n_cur, chunk + 1u, lane);
            } else {
                n_next = pass1_build_list(Ln, list_next, lane);
                next = pass1_load(p, Ln, list_next, n_next, 0u, lane);
            }
            total = pass1_eval<PROJ>(p, L, planes, alpha, lane, cur, total, payload);
            cur = next;
            if (last) break;
        }
        setup_write(p, lds[wave][(it + 2) % 3], rec2, mat2, lane, prev2); // also fences the mask atomics of this tile
        if ((p.task_records != nullptr || p.visible_list == 2u || total > kPayloadCap) && lane < 2 * (int)kTileRows) // see meshlet_eval
            p.tile_masks[(size_t)w0 * (2 * kTileRows) + lane] = reinterpret_cast<const uint32_t *>(L.draw_mask)[lane];
        if (lane == 0) p.tile_counts[w0] = total;
        uint16_t *t = list_cur;
        list_cur = list_next;
        list_next = t;
        n_cur = n_next;
        w0 = w1;
        w1 = w2;
        w2 = w3;
        rec2 = rec3;
        it++;
    }
}

template <int PROJ>
__global__ __launch_bounds__(kEvWaves * 64, kEvWavesPerSimd) void meshlet_pass1_kernel(const MeshletCullParams p) {
    meshlet_pass1_body<PROJ>(p);
}
template <int PROJ>
__global__ __launch_bounds__(kEvWaves * 64, kEvWavesPerSimd) void meshlet_pass1_views_kernel(const MeshletCullViews all,
                                                                                            const ViewGroup g) {
    meshlet_pass1_body<PROJ>(all.v[g.idx[blockIdx.y]]);
}

// One wave per tile (grid stride): commands of the tile's survivors at the
// scanned offset (:216-230).  Everything a tile needs is loaded ahead of its
// turn, so no memory round trip is exposed between tiles: the 16 records, count
// and chunk-local base two tiles ahead; the survivor payload (whose extent
// needs the count) one tile ahead.
struct EmitTile {
    uint4 rec; // lanes 0..15: the tile's dispatch records
    uint32_t count, base;
    uint32_t mword; // STREAM: lanes 0..15: the tile's should-draw ballots (every tile is rebuilt from them)
};

template <bool STREAM>
__device__ __forceinline__ EmitTile emit_prefetch(const MeshletCullParams &p, uint32_t tile, uint32_t ntiles,
                                                  uint32_t nrec, int lane) {
    EmitTile e;
    const uint4 *recs = reinterpret_cast<const uint4 *>(p.dispatch_buffer + ORBIT_DISPATCH_HEADER);
    const bool tv = tile < ntiles;
    const uint32_t t = tv ? tile : 0u;
    const uint32_t ri = t * kTileRecords + (lane & 15);
    e.rec = *((tv && ri < nrec) ? recs + ri : reinterpret_cast<const uint4 *>(p.zero_page));
    e.count = tv ? p.tile_counts[t] : 0u;
    e.base = p.tile_base[t]; // exclusive prefix inside the tile's scan chunk
    e.mword = 0u;
    if constexpr (STREAM)
        e.mword = *(tv ? p.tile_masks + ((size_t)t * (2 * kTileRows) + (lane & 15))
                       : reinterpret_cast<const uint32_t *>(p.zero_page));
    return e;
}

// Payload entries [first, first + 64) of a tile; lanes past the count read the zero page.
__device__ __forceinline__ Payload emit_load_payload(const MeshletCullParams &p, uint32_t tile, uint32_t count,
                                                   uint32_t first, int lane) {
    const uint32_t j = first + (uint32_t)lane;
    const bool has = j < count && count <= kPayloadCap;
    return *(has ? p.tile_payload + ((size_t)tile * kPayloadCap + j) : reinterpret_cast<const Payload *>(p.zero_page));
}

constexpr uint32_t kEmitChunkTable = 1024; // chunk bases kept in LDS (1 Mi tiles = 16 Mi dispatch records)

// Lane = survivor.  A tile with at most kPayloadCap survivors (the common case:
// ~50 at 10 % survival) is one contiguous payload run written by meshlet_eval in
// command order; the lane turns its entry into the 7 command words, and the run
// of 7-dword structs goes out through LDS so that each of the 7 stores covers
// 256 contiguous bytes instead of 64 addresses 28 B apart.  A denser tile is
// rebuilt from its ballots: survivors compacted into an LDS list of (row, lane)
// codes (ballot rank = list position, i.e. canonical order), Meshlet bytes
// 16..31 gathered.  The second scan level (chunk sums -> chunk bases, at most a
// few dozen values) is redone by every block in LDS instead of being a launch.
// VISIBLE: instead of the 28-B commands the launch writes the compact visible list — 8 B per survivor
// {meshlet_index, entity_index} after an 8-B {count, 0} header — which is what the sharded engine sends over
// xGMI (visible_expand_kernel turns the gathered lists into commands on the receiving GPU).
// STREAM (derived meshlet streams): meshlet_eval kept no payload; every tile is rebuilt from its ballots.  The visible
// list needs nothing else (this body); draw commands come from meshlet_emit_chain_body below.
template <bool VISIBLE, bool STREAM = false>
__device__ __forceinline__ void meshlet_emit_body(const MeshletCullParams &p) {
    static_assert(VISIBLE || !STREAM, "draw commands from the streams: meshlet_emit_chain_body");
    __shared__ uint16_t s_list[4][kTileRows * 64];
    __shared__ uint32_t s_cmd[4][64 * 7];
    __shared__ uint32_t s_chunk_base[kEmitChunkTable];
    __shared__ uint32_t s_scan[256 / 64 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t *list = s_list[wave];
    uint32_t *cmd = s_cmd[wave];
    const uint32_t nrec = min(*reinterpret_cast<const uint32_t *>(p.dispatch_buffer), p.dispatch_capacity);
    const uint32_t ntiles = (nrec + kTileRecords - 1) / kTileRecords;
    const uint32_t nchunks = (ntiles + kScanChunk - 1) / kScanChunk;
    uint32_t grand_total = 0;
    for (uint32_t c0 = 0; c0 < nchunks; c0 += 256) { // block-uniform trip count
        const uint32_t c = c0 + threadIdx.x;
        const uint32_t v = c < nchunks ? p.chunk_sums[c] : 0u;
        uint32_t sub;
        const uint32_t ex = block_exclusive_scan<4>(v, s_scan, &sub);
        if (c < nchunks && c < kEmitChunkTable) s_chunk_base[c] = grand_total + ex;
        grand_total += sub;
    }
    __syncthreads();
    uint32_t *out = reinterpret_cast<uint32_t *>(p.draw_buffer + ORBIT_DRAW_HEADER);
    const uint32_t stride = gridDim.x * 4;
    uint32_t tile = blockIdx.x * 4 + wave;
    EmitTile cur = emit_prefetch<STREAM>(p, tile, ntiles, nrec, lane);
    EmitTile nxt = emit_prefetch<STREAM>(p, tile + stride, ntiles, nrec, lane);
    Payload cur_pay = Payload{0u, 0u, 0u};
    if constexpr (!STREAM) cur_pay = emit_load_payload(p, tile, cur.count, 0u, lane);
    for (; tile < ntiles; tile += stride) {
        const EmitTile nn = emit_prefetch<STREAM>(p, tile + 2 * stride, ntiles, nrec, lane);
        Payload nxt_pay = Payload{0u, 0u, 0u};
        if constexpr (!STREAM) nxt_pay = emit_load_payload(p, tile + stride, nxt.count, 0u, lane);
        const uint32_t n = cur.count;
        if (n != 0u) { // wave-uniform
            const uint32_t base = cur.base + s_chunk_base[tile / kScanChunk];
            const bool dense = STREAM || n > kPayloadCap; // wave-uniform
            if (dense) {
                const uint32_t mword = STREAM ? cur.mword : p.tile_masks[(size_t)tile * (2 * kTileRows) + (lane & 15)];
                uint32_t row_base = 0;
#pragma unroll
                for (int r = 0; r < (int)kTileRows; r++) {
                    const uint32_t lo = (uint32_t)__shfl((int)mword, 2 * r, 64);
                    const uint32_t hi = (uint32_t)__shfl((int)mword, 2 * r + 1, 64);
                    const uint64_t mask = ((uint64_t)hi << 32) | lo;
                    if ((mask >> lane) & 1ull) list[row_base + lane_prefix(mask)] = (uint16_t)(r * 64 + lane);
                    row_base += (uint32_t)__popcll(mask);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            for (uint32_t j0 = 0; j0 < n; j0 += 64) { // one trip unless more than 64 survivors
                const uint32_t j = j0 + lane;
                Payload pay; // payload layout (make_payload)
                if (dense) {
                    const uint32_t code = j < n ? list[j] : 0u;
                    const uint32_t rid = (code >> 6) * 2u + ((code >> 5) & 1u); // record 2r + half
                    const uint32_t mi = (uint32_t)__shfl((int)cur.rec.y, (int)rid, 64) + (code & 31u);
                    uint4 b = make_uint4(0u, 0u, 0u, 0u); // STREAM: the visible list carries indices only
                    if constexpr (!STREAM)
                        b = *(j < n ? reinterpret_cast<const uint4 *>(p.meshlets + mi) + 1
                                    : reinterpret_cast<const uint4 *>(p.zero_page));
                    pay = make_payload(code & 31u, rid, b);
                } else {
                    pay = j0 == 0u ? cur_pay : emit_load_payload(p, tile, n, j0, lane);
                }
                const uint32_t pay_rid = pay.w & 15u;
                const uint32_t ent = (uint32_t)__shfl((int)cur.rec.x, (int)pay_rid, 64);
                const uint32_t pay_meshlet = (uint32_t)__shfl((int)cur.rec.y, (int)pay_rid, 64) + ((pay.w >> 4) & 31u);
                if (VISIBLE) {
                    const uint32_t slot = base + j;
                    if (j < n && slot < p.draw_capacity)
                        reinterpret_cast<uint2 *>(p.draw_buffer + ORBIT_VISIBLE_HEADER)[slot] = make_uint2(pay_meshlet, ent);
                    continue;
                }
                const uint32_t vertex_count = (pay.w >> 16) & 0xFFu;
                const uint32_t triangle_count = pay.w >> 24;
                uint32_t *c = cmd + lane * 7;
                c[0] = triangle_count * 3u;         // cmd_index_count
                c[1] = 1u;                          // cmd_instance_count
                c[2] = (pay.z + vertex_count) * 4u; // cmd_first_index
                c[3] = pay.z;                       // cmd_vertex_offset = int(data_offset)
                c[4] = ent;                         // cmd_first_instance = entity_index
                c[5] = pay.y;                       // meshlet_vertex_offset
                c[6] = pay_meshlet;                 // meshlet_index
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t first = base + j0; // slot of the trip's first command
                const uint32_t room = first < p.draw_capacity ? p.draw_capacity - first : 0u;
                const uint32_t words = min(min(n - j0, 64u), room) * 7u;
                uint32_t *dst = out + (size_t)first * 7u;
#pragma unroll
                for (uint32_t k = 0; k < 7; k++) {
                    const uint32_t w = k * 64u + (uint32_t)lane;
                    if (w < words) dst[w] = cmd[w];
                }
                __builtin_amdgcn_wave_barrier(); // cmd is rewritten by the next trip
            }
            __builtin_amdgcn_wave_barrier(); // list is reused by the next tile
        }
        cur = nxt;
        cur_pay = nxt_pay;
        nxt = nn;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // `count` header: subsumes the fill_buffer(count = 0) of draw_gen.rs:411-417
        *p.total = grand_total;
        *reinterpret_cast<uint32_t *>(p.draw_buffer) = min(grand_total, p.draw_capacity);
        if (VISIBLE) reinterpret_cast<uint32_t *>(p.draw_buffer)[1] = 0u;
        if (grand_total > p.draw_capacity) latch_status(p.status, ORBIT_E_CAPACITY);
    }
}

// ---------------------------------------------------------------------------
// Draw commands from the derived streams.  meshlet_eval kept only ballots; a survivor's command needs its meshlet's
// vertex_offset, data_offset and the two counts (:216-230).  Gathering them per survivor (12 B out of a 128-B line
// each) moved 3x the bytes the commands hold; this body streams 2 B per meshlet of the records that have survivors
// instead and derives the rest, because the uploader lays a mesh's meshlets out as a chain (kernels.h
// MeshletStreamView): data_offset[i + 1] = data_offset[i] + vertex_count[i] + ceil(3 * triangle_count[i] / 4), one
// vertex_offset per submesh.
// The launch is bound by vector-instruction issue (profiles/r03_notes.md), so a wave works on a GROUP of two
// consecutive wave tiles — 32 dispatch records — at a time: everything that is per record runs on 32 lanes instead of
// 16, and everything that is per tile is paid once per two.
//   phase A  lane = two neighbouring meshlets of a record (a 16-lane DPP row = one record, four records per
//            instruction): data sizes from the counts (one 4-B load per lane), prefix sum inside each record (four
//            DPP steps, two quad-rows per scan as 16-bit halves), {prefix, counts} of every meshlet and the records'
//            totals to LDS; the survivors' (record, lane) codes peeled off the ballots in canonical order.
//   phase B  lane = record: a record's base is the gathered {vertex_offset, data_offset} of its first meshlet where a
//            chain starts (first record of the group with survivors, a jump in meshlet_offset, a predecessor without
//            survivors, a broken link since the predecessor's second meshlet: all known a group ahead, so the gather
//            is prefetched), else the predecessor's base + total — a segmented prefix sum over the 32 records.
//   phase C  lane = survivor: offsets = record base + prefix, valid while the record's link bits up to the lane are
//            all set; the few survivors behind a broken link gather their own 8 B.  Commands go out through LDS as
//            contiguous words (as in meshlet_emit_body).
// Pipeline per wave: records and ballots three groups ahead; link words two; counts and chain-start bases one.
// All loads are range-checked buffer loads (an offset outside the resource reads zero): no pointer selects, no
// branches around loads, 32-bit address arithmetic, and the stream's coverage check comes for free.
constexpr uint32_t kGroupTiles = 2, kGroupRecords = kGroupTiles * kTileRecords; // 32 records = 1024 meshlets
constexpr uint32_t kSlowWords = 32; // 1024 iterations per wave: more than any grid the library launches leaves a wave

struct __attribute__((aligned(16))) ChainLds {
    uint4 rec[kGroupRecords];            // phase B -> C: {data_offset base, vertex_offset, entity_index, meshlet_offset}
    uint2 rec2[kGroupRecords];           // ... behind the record's first broken link: {data_offset base, vertex_offset}
    uint32_t rs[kGroupRecords];          // chain_stage: per record the byte offset of its first count (or "none")
    uint32_t row[kGroupRecords * 32];    // phase A -> C, per (record, lane): size prefix inside the record | counts << 16
    uint32_t link[kGroupRecords];        // link bits of the record's 32 lanes
    uint32_t tot[kGroupRecords / 2];     // [quad-row pair * 4 + sub]: data words of records 4 q + sub / 4 (q + 4) + sub (16:16)
    uint32_t slow[kSlowWords];           // bit i: the wave's i-th group needs the general form (process<!FAST>)
    uint32_t cmd[64 * 7];
    uint16_t list[kGroupRecords * 32];
};


struct ChainRsrc {
    __amdgpu_buffer_rsrc_t recs, counts, bases, masks; // the dispatch records, the scan scratch
    __amdgpu_buffer_rsrc_t cnt, link, cmd;             // the stream (based at its first meshlet)
    uint32_t first, count, first_word;
};

struct ChainGroup {
    uint32_t ent, y, z; // lane = record of the group (lanes 32.. repeat lanes 0..31): entity, meshlet offset, count
    uint32_t mword;     // the record's should-draw ballot
    uint32_t count;     // lane & 1: survivors of the group's tile (lane & 1)
    uint32_t base;      // chunk-local output slot of the group's first survivor
};

__device__ __forceinline__ ChainGroup chain_prefetch(const ChainRsrc &B, uint32_t group, int lane) {
    ChainGroup e;
    const uint32_t g = group < 0x007FFFFFu ? group : 0x007FFFFFu; // keeps g * 512 out of range instead of wrapping
    const uint32_t k = (uint32_t)lane & 31u;
    const auto r = __builtin_amdgcn_raw_buffer_load_b96(B.recs, (g * kGroupRecords + k) * 16u, 0, 0);
    e.ent = r[0], e.y = r[1], e.z = r[2];
    e.mword = __builtin_amdgcn_raw_buffer_load_b32(B.masks, (g * kGroupRecords + k) * 4u, 0, 0);
    e.count = __builtin_amdgcn_raw_buffer_load_b32(B.counts, (g * kGroupTiles + ((uint32_t)lane & 1u)) * 4u, 0, 0);
    e.base = __builtin_amdgcn_raw_buffer_load_b32(B.bases, g * kGroupTiles * 4u, 0, 0);
    return e;
}

struct LinkStage {
    uint32_t l0, l1; // lane = record: the two link words that cover its lanes
};

struct ChainStage {
    uint32_t c2[kGroupRecords / 4]; // per quad-row: the counts of this lane's two meshlets (16:16; 0: not loaded)
    uint32_t link32;                // lane = record: bit l = link bit of meshlet (meshlet_offset + l)
    uint32_t bv, bd;                // lane = record: vertex_offset / data_offset of its first meshlet (chain starts only)
    uint32_t bv2, bd2;              // lane = record: those of the meshlet behind the record's first broken link, if any
    uint32_t flags;                 // lane = record: 1 = has survivors (counts loaded), 2 = chain start (bv, bd loaded)
};

// lane = record: does the emit read this record at all?  Records without survivors are skipped; a record that begins
// outside the stream reads zeros (ORBIT_E_RANGE is the evaluation's to latch).
__device__ __forceinline__ bool chain_has(const ChainRsrc &B, const ChainGroup &e) {
    return e.mword != 0u && e.z != 0u && e.y - B.first < B.count;
}

__device__ __forceinline__ LinkStage link_stage(const ChainRsrc &B, const ChainGroup &e) {
    const uint32_t wo = chain_has(B, e) ? ((e.y >> 5) - B.first_word) * 4u : kNoOffset;
    LinkStage k;
    k.l0 = __builtin_amdgcn_raw_buffer_load_b32(B.link, wo, 0, 0);
    k.l1 = __builtin_amdgcn_raw_buffer_load_b32(B.link, wo, 0 + 4, 0);
    return k;
}

// First lane of a record (1..31) whose link bit is clear, 32 if the links of lanes 1..31 all hold.
__device__ __forceinline__ uint32_t chain_first_break(uint32_t link32) {
    const uint32_t open = ~link32 & 0xFFFFFFFEu;
    return open != 0u ? (uint32_t)__builtin_ctz(open) : 32u;
}

// Lane i <- lane i - 1 of the wave (wave_shr:1); lane 0 gets `first`.
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t first) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xF, 0xF, false);
}

__device__ __forceinline__ ChainStage chain_stage(const ChainRsrc &B, ChainLds &S, const ChainGroup &e,
                                                  const LinkStage &k, int lane) {
    ChainStage s;
    const bool has = chain_has(B, e);
    if (lane < (int)kGroupRecords) S.rs[lane] = has ? (e.y - B.first) * 2u : kNoOffset;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // quad-row q: the wave's four 16-lane rows are records 4 q .. 4 q + 3, lane i of a row holds meshlets 2 i, 2 i + 1.
    // Lanes past a record's count read its neighbours' counts (or zeros past the stream): they feed nothing that is
    // used — a prefix of lanes behind them, and the total of a record that, being short, starts no continuation.
    const uint32_t sub = (uint32_t)lane >> 4, i4 = ((uint32_t)lane & 15u) * 4u;
    uint32_t off[kGroupRecords / 4];
#pragma unroll
    for (int q = 0; q < (int)kGroupRecords / 4; q++) off[q] = S.rs[4 * q + sub];
#pragma unroll
    for (int q = 0; q < (int)kGroupRecords / 4; q++)
        s.c2[q] = __builtin_amdgcn_raw_buffer_load_b32(B.cnt, off[q] + i4, 0, ORBIT_EMIT_CNT_AUX);
    // lane = record.  A record continues its predecessor's chain when that one was read too, ends where this one
    // begins, and every link from the predecessor's second meshlet to this record's first one holds; otherwise it
    // starts a chain from the gathered offsets of its first meshlet.
    s.link32 = __funnelshift_r(k.l0, k.l1, e.y & 31u);
    const bool full = has && e.z == 32u && (s.link32 | 1u) == 0xFFFFFFFFu; // the chain reaches the record's end
    const uint32_t prev_y = wave_shr1(e.y, 0u), prev_full = wave_shr1(full ? 1u : 0u, 0u);
    const bool first = ((uint32_t)lane & 31u) == 0u; // (lane 32 repeats record 0)
    const bool cont = has && !first && prev_full != 0u && e.y == prev_y + 32u && (s.link32 & 1u) != 0u;
    const bool start = has && !cont;
    s.flags = (has ? 1u : 0u) | (start ? 2u : 0u);
    const uint32_t co = start ? (e.y - B.first) * 12u : kNoOffset;
    s.bv = __builtin_amdgcn_raw_buffer_load_b32(B.cmd, co, 0, 0);
    s.bd = __builtin_amdgcn_raw_buffer_load_b32(B.cmd, co, 0 + 4, 0);
    // A submesh or allocation boundary INSIDE the record restarts the chain there: the offsets of the meshlet behind the
    // first broken link are gathered as well (none: nothing is read).  Survivors behind a second one gather their own.
    const uint32_t brk = chain_first_break(s.link32);
    const uint32_t co2 = (has && brk < 32u) ? (e.y + brk - B.first) * 12u : kNoOffset;
    s.bv2 = __builtin_amdgcn_raw_buffer_load_b32(B.cmd, co2, 0, 0);
    s.bd2 = __builtin_amdgcn_raw_buffer_load_b32(B.cmd, co2, 0 + 4, 0);
    return s;
}

// data sizes of two meshlets, packed 16:16, from their packed counts (meshlet_data_words, kernels.h)
__device__ __forceinline__ uint32_t chain_size2(uint32_t c2) {
    const uint32_t vc = c2 & 0x00FF00FFu, tc = (c2 >> 8) & 0x00FF00FFu;
    return vc + (((tc * 3u + 0x00030003u) >> 2) & 0x3FFF3FFFu);
}

// Inclusive scan (sum) inside each row of 16 lanes: row_shr 1 / 2 / 4 / 8.
__device__ __forceinline__ uint32_t row_inclusive_scan(uint32_t v) {
#define ORBIT_ROW_STEP(ctrl) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xF, 0xF, false)
    ORBIT_ROW_STEP(0x111);
    ORBIT_ROW_STEP(0x112);
    ORBIT_ROW_STEP(0x114);
    ORBIT_ROW_STEP(0x118);
#undef ORBIT_ROW_STEP
    return v;
}

__device__ __forceinline__ void meshlet_emit_chain_body(const MeshletCullParams &p) {
    __shared__ ChainLds s_chain[4];
    __shared__ uint32_t s_chunk_base[kEmitChunkTable];
    __shared__ uint32_t s_scan[256 / 64 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ChainLds &S = s_chain[wave];
    const uint32_t nrec = min(*reinterpret_cast<const uint32_t *>(p.dispatch_buffer), p.dispatch_capacity);
    const uint32_t ntiles = (nrec + kTileRecords - 1) / kTileRecords;
    const uint32_t ngroups = (ntiles + kGroupTiles - 1) / kGroupTiles;
    const uint32_t nchunks = (ntiles + kScanChunk - 1) / kScanChunk;
    static_assert(kScanChunk % kGroupTiles == 0, "a group's tiles share a scan chunk: its survivors are contiguous");
    uint32_t grand_total = 0;
    for (uint32_t c0 = 0; c0 < nchunks; c0 += 256) { // block-uniform trip count
        const uint32_t c = c0 + threadIdx.x;
        const uint32_t v = c < nchunks ? p.chunk_sums[c] : 0u;
        uint32_t sub;
        const uint32_t ex = block_exclusive_scan<4>(v, s_scan, &sub);
        if (c < nchunks && c < kEmitChunkTable) s_chunk_base[c] = grand_total + ex;
        grand_total += sub;
    }
    __syncthreads();
    ChainRsrc B;
    B.recs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.dispatch_buffer) + ORBIT_DISPATCH_HEADER, 0,
                                               nrec * 16u, kBufFlags);
    B.masks = __builtin_amdgcn_make_buffer_rsrc(p.tile_masks, 0, ntiles * 64u, kBufFlags);
    B.counts = __builtin_amdgcn_make_buffer_rsrc(p.tile_counts, 0, ntiles * 4u, kBufFlags);
    B.bases = __builtin_amdgcn_make_buffer_rsrc(p.tile_base, 0, ntiles * 4u, kBufFlags);
    B.first = p.ms.first;
    B.count = p.ms.count;
    B.first_word = p.ms.first >> 5;
    // + 2 bytes: the 4-B load of a lane whose first meshlet is the stream's last one (the arrays are padded)
    B.cnt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(p.ms.cnt + p.ms.first), 0, p.ms.count * 2u + 2u, kBufFlags);
    B.link = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.ms.link + B.first_word), 0,
                                               (((p.ms.first + p.ms.count + 31u) >> 5) - B.first_word + 2u) * 4u, kBufFlags);
    B.cmd = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.ms.cmd + (size_t)p.ms.first * 3u), 0,
                                              p.ms.count * 12u, kBufFlags);
    uint32_t *out = reinterpret_cast<uint32_t *>(p.draw_buffer + ORBIT_DRAW_HEADER);
    const uint32_t stride = gridDim.x * 4;
    const uint32_t sub = (uint32_t)lane >> 4, i16 = (uint32_t)lane & 15u;
    // One group's phases, from its prefetched records (`cur`) and staged counts / bases (`cs`).
    // FAST (the pipelined loop): straight-line vector-memory code.  The wave's vector-memory operations retire in
    // issue order, so waiting for a prefetched load also waits for every store issued before it — unless the compiler
    // knows exactly how many operations lie in between.  Hence, in that loop, no load or store in any branch: the same
    // fourteen stores per group whatever its survivors (a group without any stores into an empty window).  A group
    // that needs more — over 128 survivors, or a survivor behind a SECOND boundary inside one record — stores nothing
    // there, returns true, and is done again by the general form (!FAST) after the loop.
    auto process = [&](auto fast_tag, const ChainGroup &cur, const ChainStage &cs, uint32_t group) -> bool {
        constexpr bool FAST = decltype(fast_tag)::value;
        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)cur.count, 0) +
                           (uint32_t)__builtin_amdgcn_readlane((int)cur.count, 1);
        // wave-uniform, and provably so for the compiler (the store window's descriptor lives in SGPRs)
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)(cur.base + s_chunk_base[min(group * kGroupTiles / kScanChunk, kEmitChunkTable - 1u)]));
        // ---- phase A: quad-rows q and q + 4 share a scan (16-bit halves: a record's sizes sum to < 2^14)
#pragma unroll
        for (int q = 0; q < (int)kGroupRecords / 8; q++) {
            const uint32_t ca = cs.c2[q], cb = cs.c2[q + 4];
            const uint32_t sa = chain_size2(ca), sb = chain_size2(cb);         // {first, second} meshlet of the lane
            const uint32_t pa = (sa & 0xFFFFu) + (sa >> 16), pb = (sb & 0xFFFFu) + (sb >> 16);
            const uint32_t pair = pa | (pb << 16);
            const uint32_t incl = row_inclusive_scan(pair);
            const uint32_t excl = incl - pair;
            const uint32_t ea = excl & 0xFFFFu, eb = excl >> 16;
            // per meshlet: prefix | counts << 16 (the second meshlet's prefix includes the first one's size)
            uint2 wa, wb;
            wa.x = ea | (ca << 16);
            wa.y = ((ea + (sa & 0xFFFFu)) & 0xFFFFu) | (ca & 0xFFFF0000u);
            wb.x = eb | (cb << 16);
            wb.y = ((eb + (sb & 0xFFFFu)) & 0xFFFFu) | (cb & 0xFFFF0000u);
            *reinterpret_cast<uint2 *>(&S.row[q * 128 + lane * 2]) = wa;       // record 4 q + sub, lanes 2 i, 2 i + 1
            *reinterpret_cast<uint2 *>(&S.row[(q + 4) * 128 + lane * 2]) = wb;
            if (i16 == 15u) S.tot[q * 4 + sub] = incl;
        }
        // the survivors' codes (record * 32 + lane) in canonical order: lane i owns half a record's ballot and
        // peels its bits off one per trip
        {
            const uint32_t w = (uint32_t)__shfl((int)cur.mword, lane >> 1, 64);
            uint32_t bits = (w >> (16u * ((uint32_t)lane & 1u))) & 0xFFFFu;
            const uint32_t pc = (uint32_t)__popc(bits);
            uint32_t pos = wave_inclusive_scan(pc) - pc;
            const uint32_t code0 = (uint32_t)lane * 16u;
            while (__ballot(bits != 0u) != 0ull) {
                if (bits != 0u) {
                    S.list[pos] = (uint16_t)(code0 + (uint32_t)__builtin_ctz(bits));
                    pos++;
                    bits &= bits - 1u;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- phase B: lane = record (32 of them; lanes 32.. repeat)
        {
            const uint32_t k = (uint32_t)lane & 31u, kq = k >> 2;
            const uint32_t tot = (S.tot[(kq & 3u) * 4u + (k & 3u)] >> (16u * (kq >> 2))) & 0xFFFFu;
            const uint32_t prev_tot = wave_shr1(tot, 0u);
            const uint32_t brk = chain_first_break(cs.link32);
            const uint32_t brk_prefix = S.row[k * 32u + (brk & 31u)] & 0xFFFFu; // data words in front of that meshlet
            const bool cont = (cs.flags & 3u) == 1u; // has survivors and continues the predecessor's chain
            uint32_t f = cont ? 0u : 1u;             // segment head
            uint32_t a_d = f ? cs.bd : prev_tot, a_v = cs.bv; // bv / bd are zero unless the record starts a chain
#define ORBIT_SEG_STEP(ctrl, rows)                                                                              \
    {                                                                                                           \
        const uint32_t sd = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_d, ctrl, rows, 0xF, false);         \
        const uint32_t sv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a_v, ctrl, rows, 0xF, false);         \
        const uint32_t sf = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, ctrl, rows, 0xF, false);           \
        a_d += f ? 0u : sd;                                                                                     \
        a_v += f ? 0u : sv;                                                                                     \
        f |= sf;                                                                                                \
    }
            // (a lane without a source in a step keeps what it has: record 16 has no predecessor in its row of 16
            // lanes and is still no head — its segment arrives with the row_bcast15 step)
            ORBIT_SEG_STEP(0x111, 0xF)
            ORBIT_SEG_STEP(0x112, 0xF)
            ORBIT_SEG_STEP(0x114, 0xF)
            ORBIT_SEG_STEP(0x118, 0xF)
            ORBIT_SEG_STEP(0x142, 0xA) // row_bcast15: records 16..31 continue from record 15 (lanes 48.. from 47)
#undef ORBIT_SEG_STEP
            if (lane < (int)kGroupRecords) {
                S.rec[lane] = make_uint4(a_d, a_v, cur.ent, cur.y);
                S.rec2[lane] = make_uint2(cs.bd2 - brk_prefix, cs.bv2);
                S.link[lane] = cs.link32;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- phase C: lane = survivor; one trip = up to 64 commands
        // is survivor j behind a second broken link of its record?  (then only its own gathered words are right)
        auto second_break = [&](uint32_t j) -> bool {
            const uint32_t code = j < n ? S.list[j] : 0u;
            const uint32_t lk = S.link[code >> 5], l = code & 31u;
            const uint32_t brk = chain_first_break(lk);
            const uint32_t need = ((2u << l) - 1u) & ~((2u << (brk & 31u)) - 1u); // links of lanes brk + 1 .. l
            return j < n && l >= brk && (lk & need) != need;
        };
        bool slow = false;
        if constexpr (FAST)
            slow = n > 128u || __ballot(second_break((uint32_t)lane) || second_break(64u + (uint32_t)lane)) != 0ull;
        auto trip = [&](uint32_t j0) {
            const uint32_t j = j0 + (uint32_t)lane;
            const uint32_t code = j < n ? S.list[j] : 0u;
            const uint32_t rid = code >> 5, l = code & 31u;
            const uint32_t pk = S.row[code];
            const uint4 R = S.rec[rid];
            const uint2 R2 = S.rec2[rid];
            const uint32_t mi = R.w + l;
            // in front of the record's first broken link: the record's base; behind it: the base gathered there
            const bool behind = l >= chain_first_break(S.link[rid]);
            uint32_t data = (behind ? R2.x : R.x) + (pk & 0xFFFFu), vtx = behind ? R2.y : R.y;
            if constexpr (!FAST) {
                const bool broken = second_break(j);
                if (__ballot(broken) != 0ull) { // rare
                    const uint32_t co = broken ? (mi - B.first) * 12u : kNoOffset;
                    const uint32_t gv = __builtin_amdgcn_raw_buffer_load_b32(B.cmd, co, 0, 0);
                    const uint32_t gd = __builtin_amdgcn_raw_buffer_load_b32(B.cmd, co, 0 + 4, 0);
                    vtx = broken ? gv : vtx;
                    data = broken ? gd : data;
                }
            }
            uint32_t vertex_count = (pk >> 16) & 0xFFu, triangle_count = pk >> 24;
            // ORBIT_E_RANGE (latched by the evaluation): the command of a meshlet outside the stream carries zeros
            if (!p.ms.covers(mi)) vtx = data = vertex_count = triangle_count = 0u;
            uint32_t *c = S.cmd + lane * 7;
            c[0] = triangle_count * 3u;         // cmd_index_count
            c[1] = 1u;                          // cmd_instance_count
            c[2] = (data + vertex_count) * 4u;  // cmd_first_index
            c[3] = data;                        // cmd_vertex_offset = int(data_offset)
            c[4] = R.z;                         // cmd_first_instance = entity_index
            c[5] = vtx;                         // meshlet_vertex_offset
            c[6] = mi;                          // meshlet_index
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // the trip's commands as one range-checked window: words past the list or the capacity are dropped
            const uint32_t first = base + j0; // slot of the trip's first command
            const uint32_t room = first < p.draw_capacity ? p.draw_capacity - first : 0u;
            const uint32_t left = (j0 < n && !slow) ? n - j0 : 0u;
            const uint32_t bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(min(min(left, 64u), room) * 28u));
            const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)first * 7u, 0, bytes, kBufFlags);
#pragma unroll
            for (uint32_t k = 0; k < 7; k++) {
                const uint32_t w = k * 64u + (uint32_t)lane;
                __builtin_amdgcn_raw_buffer_store_b32(S.cmd[w], win, w * 4u, 0, ORBIT_EMIT_STORE_AUX);
            }
            __builtin_amdgcn_wave_barrier(); // cmd is rewritten by the next trip
        };
        if constexpr (FAST) {
            trip(0u);
            trip(64u);
        } else {
            for (uint32_t j0 = 0u; j0 < n; j0 += 64u) trip(j0);
        }
        __builtin_amdgcn_wave_barrier(); // list / row / rec are reused by the next group
        return slow;
    };

    // The pipelined loop over the wave's groups: records and ballots three groups ahead, link words two, counts and
    // bases one.  A group past the end reads zero records: nothing else is loaded for it.
    const uint32_t group0 = blockIdx.x * 4 + wave;
    uint32_t group = group0, it = 0;
    ChainGroup cur = chain_prefetch(B, group, lane);
    ChainGroup nxt = chain_prefetch(B, group + stride, lane);
    ChainGroup nn = chain_prefetch(B, group + 2 * stride, lane);
    ChainStage cs = chain_stage(B, S, cur, link_stage(B, cur), lane);
    LinkStage nk = link_stage(B, nxt);
    if (lane < (int)kSlowWords) S.slow[lane] = 0u;
    auto step = [&]() __attribute__((always_inline)) {
        const ChainGroup n3 = chain_prefetch(B, group + 3 * stride, lane);
        const LinkStage nnk = link_stage(B, nn);
        const ChainStage ns = chain_stage(B, S, nxt, nk, lane);
        const bool slow = process(std::true_type{}, cur, cs, group);
        // (the launchers size the grid so that a wave never has more than 32 * kSlowWords iterations)
        if (slow && lane == 0) S.slow[(it >> 5) & (kSlowWords - 1u)] |= 1u << (it & 31u);
        cur = nxt;
        cs = ns;
        nxt = nn;
        nk = nnk;
        nn = n3;
        group += stride;
        it++;
    };
    // The first group outside the loop: the loop is then entered with its steady-state queue of loads and stores in
    // flight, and the compiler's wait counts inside it are the steady state's (from the prologue alone the link
    // words would look six operations old, and every iteration would drain its predecessor's stores to be safe).
    if (group < ngroups) {
        step();
        while (group < ngroups) step();
    }
    // The groups the loop left: one at a time, loads awaited as they come.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t w = 0; w < kSlowWords; w++) {
        uint32_t bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.slow[w]);
        while (bits != 0u) {
            const uint32_t g = group0 + (w * 32u + (uint32_t)__builtin_ctz(bits)) * stride;
            bits &= bits - 1u;
            const ChainGroup e = chain_prefetch(B, g, lane);
            const ChainStage st = chain_stage(B, S, e, link_stage(B, e), lane);
            process(std::false_type{}, e, st, g);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // `count` header: subsumes the fill_buffer(count = 0) of draw_gen.rs:411-417
        *p.total = grand_total;
        *reinterpret_cast<uint32_t *>(p.draw_buffer) = min(grand_total, p.draw_capacity);
        if (grand_total > p.draw_capacity) latch_status(p.status, ORBIT_E_CAPACITY);
    }
}

template <bool VISIBLE, bool STREAM = false>
__global__ __launch_bounds__(256) void meshlet_emit_kernel(const MeshletCullParams p) {
    meshlet_emit_body<VISIBLE, STREAM>(p);
}
template <bool STREAM>
__global__ __launch_bounds__(256) void meshlet_emit_views_kernel(const MeshletCullViews all, const ViewGroup g) {
    meshlet_emit_body<false, STREAM>(all.v[g.idx[blockIdx.y]]);
}
// the chain emit is bound by instruction issue: kChainWavesPerSimd resident waves per SIMD (the register budget the
// compiler gets) and exactly that many blocks per CU in the grid — a block more per CU would run as a second round
constexpr int kChainWavesPerSimd = 4;
__global__ __launch_bounds__(256, kChainWavesPerSimd) void meshlet_emit_chain_kernel(const MeshletCullParams p) {
    meshlet_emit_chain_body(p);
}
__global__ __launch_bounds__(256, kChainWavesPerSimd) void meshlet_emit_chain_views_kernel(const MeshletCullViews all,
                                                                                          const ViewGroup g) {
    meshlet_emit_chain_body(all.v[g.idx[blockIdx.y]]);
}

// Visible list -> MeshletDrawCommandBuffer (meshlet_cull.comp:216-230 for survivors that were culled on another GPU):
// lane = survivor, Meshlet bytes 16..31 gathered from the (replicated) meshlet buffer, commands staged through LDS
// so that every store covers 256 contiguous bytes.  Grid-stride over chunks of 64 survivors per wave, software
// pipelined: the items of chunk i+2 and the gathers of chunk i+1 are in flight while chunk i is written.  All loads
// are unconditional (lanes past the list read the zero page) and the stores are buffer stores whose offset is out
// of range for the words past the list, so the code is straight-line and every vmcnt wait is exact.
struct ExpandChunk {
    uint2 item;  // {meshlet_index, entity_index}
    bool valid;
};

__device__ __forceinline__ ExpandChunk expand_load_items(const uint2 *items, uint32_t first, uint32_t n, int lane,
                                                         const void *zero_page) {
    ExpandChunk c;
    const uint32_t i = first + (uint32_t)lane;
    c.valid = first < n && i < n; // first can wrap past 2^32 only with n near 2^32: excluded by draw_capacity
    c.item = *(c.valid ? items + i : reinterpret_cast<const uint2 *>(zero_page));
    return c;
}

// STREAM: the three command words come from the derived 12-B stream (`cmd`, orbit_meshlet_stream_*) instead of bytes
// 16..31 of the Meshlet: a third of the lines per survivor.
template <bool STREAM>
__device__ __forceinline__ uint4 expand_gather(const OrbitMeshlet *meshlets, const MeshletStreamView &ms,
                                               const ExpandChunk &c, const void *zero_page, int32_t *status) {
    if constexpr (STREAM) {
        const bool in = c.valid && ms.covers(c.item.x);
        if (c.valid && !in) latch_status(status, ORBIT_E_RANGE);
        const uint32_t *c3 = in ? ms.cmd + (size_t)c.item.x * 3u : reinterpret_cast<const uint32_t *>(zero_page);
        return make_uint4(0u, c3[0], c3[1], c3[2]);
    } else {
        return *(c.valid ? reinterpret_cast<const uint4 *>(meshlets + c.item.x) + 1
                         : reinterpret_cast<const uint4 *>(zero_page));
    }
}

template <bool STREAM>
__global__ __launch_bounds__(256) void visible_expand_kernel(const uint8_t *visible, const OrbitMeshlet *meshlets,
                                                             const MeshletStreamView ms, uint8_t *draw_buffer,
                                                             uint32_t draw_capacity, const void *zero_page,
                                                             int32_t *status) {
    __shared__ uint32_t s_cmd[4][64 * 7];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *cmd = s_cmd[wave];
    const uint32_t total = *reinterpret_cast<const uint32_t *>(visible);
    const uint32_t n = min(total, draw_capacity);
    const uint2 *items = reinterpret_cast<const uint2 *>(visible + ORBIT_VISIBLE_HEADER);
    uint8_t *out = draw_buffer + ORBIT_DRAW_HEADER;
    const uint32_t stride = gridDim.x * 4u * 64u;
    // wave-uniform, and provably so for the compiler (buffer descriptors live in SGPRs)
    uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * 4u + wave) * 64u));
    ExpandChunk c0 = expand_load_items(items, first, n, lane, zero_page);
    ExpandChunk c1 = expand_load_items(items, first + stride, n, lane, zero_page);
    uint4 b0 = expand_gather<STREAM>(meshlets, ms, c0, zero_page, status);
    for (; first < n; first += stride) {
        const ExpandChunk c2 = expand_load_items(items, first + 2u * stride, n, lane, zero_page);
        const uint4 b1 = expand_gather<STREAM>(meshlets, ms, c1, zero_page, status);
        uint32_t *c = cmd + lane * 7;
        c[0] = (b0.w >> 24) * 3u;                     // cmd_index_count = triangle_count * 3
        c[1] = 1u;                                    // cmd_instance_count
        c[2] = (b0.z + ((b0.w >> 16) & 0xFFu)) * 4u;  // cmd_first_index = (data_offset + vertex_count) * 4
        c[3] = b0.z;                                  // cmd_vertex_offset = int(data_offset)
        c[4] = c0.item.y;                             // cmd_first_instance = entity_index
        c[5] = b0.y;                                  // meshlet_vertex_offset
        c[6] = c0.item.x;                             // meshlet_index
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the chunk's commands as one range-checked window: words past the list fall outside and are dropped
        const uint32_t bytes = min(n - first, 64u) * 28u;
        const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)first * 28u, 0, bytes, 0x00020000);
#pragma unroll
        for (uint32_t k = 0; k < 7; k++) {
            const uint32_t w = k * 64u + (uint32_t)lane;
            __builtin_amdgcn_raw_buffer_store_b32(cmd[w], win, w * 4u, 0, 0);
        }
        __builtin_amdgcn_wave_barrier(); // cmd is rewritten by the next chunk
        c0 = c1;
        c1 = c2;
        b0 = b1;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *reinterpret_cast<uint32_t *>(draw_buffer) = n;
        if (total > draw_capacity) latch_status(status, ORBIT_E_CAPACITY);
    }
}

// ---------------------------------------------------------------------------
// The visible list compacted at record granularity (sharded engine, orbit_meshlet_cull_visible_records): 12 B per
// dispatch record that has a survivor — {entity_index, meshlet_offset, should-draw ballot of its (up to) 32 meshlets}
// — in record order, behind an 8-B {records, survivors} header.  The same ordered list of (meshlet, entity) pairs as
// the 8-B items (bit b of record r = item {meshlet_offset + b, entity_index}), in half the bytes at config 5 (3.2
// survivors per surviving record): what travels over xGMI.  Built straight from the evaluation's per-record
// ballots — no survivor scan, no emit: a count launch (block sums) and a write launch that adds up the sums of the
// blocks before it.
constexpr uint32_t kVrBlock = 1024; // records per block: 256 threads x 4

__global__ __launch_bounds__(256) void visible_records_count_kernel(const MeshletCullParams p, uint32_t max_blocks) {
    __shared__ uint32_t smem[8];
    const uint32_t nrec = min(*reinterpret_cast<const uint32_t *>(p.dispatch_buffer), p.dispatch_capacity);
    if (blockIdx.x == 0 && threadIdx.x < kTicketPools) p.tickets[threadIdx.x * kTicketStride] = 0u; // next evaluation
    const uint32_t r0 = blockIdx.x * kVrBlock + threadIdx.x * 4u;
    uint32_t nz = 0, pop = 0;
    if (blockIdx.x * kVrBlock < nrec) { // tile_masks is padded to whole tiles: the uint4 of a thread is always readable
        const uint4 m = r0 < nrec ? *reinterpret_cast<const uint4 *>(p.tile_masks + r0) : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t w[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t v = r0 + k < nrec ? w[k] : 0u;
            nz += v != 0u ? 1u : 0u;
            pop += (uint32_t)__popc(v);
        }
    }
    // (pop < 2^15 and nz < 2^11 per block: one reduction for both)
    const uint32_t both = wave_reduce_add(nz << 16 | pop);
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = both;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t s = smem[0] + smem[1] + smem[2] + smem[3];
        p.tile_base[blockIdx.x] = s >> 16;
        p.tile_base[max_blocks + blockIdx.x] = s & 0xFFFFu;
    }
}

__global__ __launch_bounds__(256) void visible_records_write_kernel(const MeshletCullParams p, uint32_t max_blocks) {
    __shared__ uint32_t smem[8];
    const uint32_t nrec = min(*reinterpret_cast<const uint32_t *>(p.dispatch_buffer), p.dispatch_capacity);
    const uint32_t nblocks = (nrec + kVrBlock - 1u) / kVrBlock;
    if (blockIdx.x >= nblocks && blockIdx.x != 0) return;
    // records of the blocks before this one; block 0 adds up everything for the header
    const uint32_t upto = blockIdx.x == 0 ? nblocks : blockIdx.x;
    uint32_t part_nz = 0, part_pop = 0;
    for (uint32_t i = threadIdx.x; i < upto; i += 256u) {
        part_nz += p.tile_base[i];
        if (blockIdx.x == 0) part_pop += p.tile_base[max_blocks + i];
    }
    part_nz = wave_reduce_add(part_nz);
    part_pop = wave_reduce_add(part_pop);
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = part_nz, smem[4 + (threadIdx.x >> 6)] = part_pop;
    __syncthreads();
    const uint32_t sum_nz = smem[0] + smem[1] + smem[2] + smem[3], sum_pop = smem[4] + smem[5] + smem[6] + smem[7];
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint32_t *hdr = reinterpret_cast<uint32_t *>(p.draw_buffer);
        hdr[0] = min(sum_nz, p.draw_capacity);
        hdr[1] = sum_pop;
        *p.total = sum_nz;
        if (sum_nz > p.draw_capacity) latch_status(p.status, ORBIT_E_CAPACITY);
    }
    if (blockIdx.x >= nblocks) return;
    const uint32_t base = blockIdx.x == 0 ? 0u : sum_nz;
    const uint32_t r0 = blockIdx.x * kVrBlock + threadIdx.x * 4u;
    const uint4 m = r0 < nrec ? *reinterpret_cast<const uint4 *>(p.tile_masks + r0) : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t w[4] = {m.x, m.y, m.z, m.w};
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) mine += (r0 + k < nrec && w[k] != 0u) ? 1u : 0u;
    uint32_t total;
    uint32_t slot = base + block_exclusive_scan<4>(mine, smem, &total);
    const uint4 *recs = reinterpret_cast<const uint4 *>(p.dispatch_buffer + ORBIT_DISPATCH_HEADER);
    uint32_t *out = reinterpret_cast<uint32_t *>(p.draw_buffer + ORBIT_VISIBLE_HEADER);
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (r0 + k < nrec && w[k] != 0u) {
            if (slot < p.draw_capacity) {
                const uint4 rec = recs[r0 + k];
                out[3 * (size_t)slot + 0] = rec.x; // entity_index
                out[3 * (size_t)slot + 1] = rec.y; // meshlet_offset
                out[3 * (size_t)slot + 2] = w[k];
            }
            slot++;
        }
    }
}

// Commands from a (gathered) record list, in list order = the canonical order: orbit_expand_visible_records.  A count
// launch sums the survivors of every block of 1024 records; here a block adds up the blocks before it, a wave takes 64
// records at a time, lays their survivors out in LDS as (record lane, bit) codes at the records' prefix offsets and
// then writes them 64 commands at a time exactly like visible_expand_kernel.
__global__ __launch_bounds__(256) void visible_records_pop_kernel(const uint8_t *records, uint32_t *block_pop,
                                                                  uint32_t max_blocks) {
    __shared__ uint32_t smem[4];
    const uint32_t n = *reinterpret_cast<const uint32_t *>(records);
    if (blockIdx.x * kVrBlock >= n || blockIdx.x >= max_blocks) return;
    const uint32_t *recs = reinterpret_cast<const uint32_t *>(records + ORBIT_VISIBLE_HEADER);
    uint32_t pop = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t r = blockIdx.x * kVrBlock + k * 256u + threadIdx.x;
        pop += r < n ? (uint32_t)__popc(recs[3 * (size_t)r + 2]) : 0u;
    }
    pop = wave_reduce_add(pop);
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = pop;
    __syncthreads();
    if (threadIdx.x == 0) block_pop[blockIdx.x] = smem[0] + smem[1] + smem[2] + smem[3];
}

template <bool STREAM>
__global__ __launch_bounds__(256) void visible_records_expand_kernel(const uint8_t *records, const uint32_t *block_pop,
                                                                     uint32_t max_blocks, const OrbitMeshlet *meshlets,
                                                                     const MeshletStreamView ms, uint8_t *draw_buffer,
                                                                     uint32_t draw_capacity, const void *zero_page,
                                                                     int32_t *status) {
    __shared__ uint32_t s_cmd[4][64 * 7];
    __shared__ uint16_t s_code[4][64 * 32];
    __shared__ uint32_t smem[8];
    __shared__ uint32_t s_chunk[kVrBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = *reinterpret_cast<const uint32_t *>(records);
    // the grid covers draw_capacity / 1024 blocks of records (every record holds a survivor: more records than that
    // cannot fit) and the scratch max_blocks: a longer list is cut there, and says so
    const uint32_t all_blocks = (n + kVrBlock - 1u) / kVrBlock;
    const uint32_t nblocks = min(min(all_blocks, max_blocks), gridDim.x);
    if (all_blocks > nblocks && blockIdx.x == 0 && threadIdx.x == 0) latch_status(status, ORBIT_E_CAPACITY);
    if (blockIdx.x >= nblocks && blockIdx.x != 0) return;
    const uint32_t upto = blockIdx.x == 0 ? nblocks : blockIdx.x;
    uint32_t part = 0;
    for (uint32_t i = threadIdx.x; i < upto; i += 256u) part += block_pop[i];
    part = wave_reduce_add(part);
    if (lane == 0) smem[wave] = part;
    __syncthreads();
    const uint32_t before = smem[0] + smem[1] + smem[2] + smem[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) { // `count` header of the command list
        *reinterpret_cast<uint32_t *>(draw_buffer) = min(before, draw_capacity);
        if (before > draw_capacity) latch_status(status, ORBIT_E_CAPACITY);
    }
    if (blockIdx.x >= nblocks) return;
    const uint32_t *recs = reinterpret_cast<const uint32_t *>(records + ORBIT_VISIBLE_HEADER);
    // the block's 16 chunks of 64 records: wave w takes chunks 4w .. 4w+3; their survivor counts first
    uint32_t ent[4], off[4], mask[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t r = blockIdx.x * kVrBlock + (wave * 4u + k) * 64u + (uint32_t)lane;
        const bool in = r < n;
        ent[k] = in ? recs[3 * (size_t)r + 0] : 0u;
        off[k] = in ? recs[3 * (size_t)r + 1] : 0u;
        mask[k] = in ? recs[3 * (size_t)r + 2] : 0u;
        const uint32_t cs = wave_reduce_add((uint32_t)__popc(mask[k]));
        if (lane == 0) s_chunk[wave * 4u + k] = cs;
    }
    __syncthreads();
    uint32_t chunk_base = blockIdx.x == 0 ? 0u : before;
    for (uint32_t c = 0; c < wave * 4u; c++) chunk_base += s_chunk[c];
    uint32_t *cmd = s_cmd[wave];
    uint16_t *code = s_code[wave];
    uint8_t *out = draw_buffer + ORBIT_DRAW_HEADER;
#pragma unroll 1
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t pc = (uint32_t)__popc(mask[k]);
        const uint32_t inc = wave_inclusive_scan(pc);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        {   // lane = record: its survivors' codes at its prefix offset, ascending bits
            uint32_t m = mask[k], at = inc - pc;
            while (m != 0u) {
                code[at++] = (uint16_t)((uint32_t)lane << 5 | (uint32_t)__builtin_ctz(m));
                m &= m - 1u;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t j0 = 0; j0 < total; j0 += 64u) {
            const uint32_t j = j0 + (uint32_t)lane;
            const bool valid = j < total;
            const uint32_t cd = valid ? code[j] : 0u;
            const uint32_t mi = (uint32_t)__shfl((int)off[k], (int)(cd >> 5), 64) + (cd & 31u);
            const uint32_t en = (uint32_t)__shfl((int)ent[k], (int)(cd >> 5), 64);
            ExpandChunk ch;
            ch.item = make_uint2(mi, en);
            ch.valid = valid;
            const uint4 b = expand_gather<STREAM>(meshlets, ms, ch, zero_page, status);
            uint32_t *c = cmd + lane * 7;
            c[0] = (b.w >> 24) * 3u;                    // cmd_index_count = triangle_count * 3
            c[1] = 1u;                                  // cmd_instance_count
            c[2] = (b.z + ((b.w >> 16) & 0xFFu)) * 4u;  // cmd_first_index = (data_offset + vertex_count) * 4
            c[3] = b.z;                                 // cmd_vertex_offset = int(data_offset)
            c[4] = en;                                  // cmd_first_instance = entity_index
            c[5] = b.y;                                 // meshlet_vertex_offset
            c[6] = mi;                                  // meshlet_index
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // the chunk's commands as one range-checked window: words past the list or the capacity are dropped
            const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(chunk_base + j0));
            const uint32_t room = first < draw_capacity ? draw_capacity - first : 0u;
            const uint32_t bytes = min(min(total - j0, 64u), room) * 28u;
            const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)first * 28u, 0, bytes, 0x00020000);
#pragma unroll
            for (uint32_t q = 0; q < 7; q++) {
                const uint32_t w = q * 64u + (uint32_t)lane;
                __builtin_amdgcn_raw_buffer_store_b32(cmd[w], win, w * 4u, 0, 0);
            }
            __builtin_amdgcn_wave_barrier(); // cmd is rewritten by the next trip
        }
        chunk_base += total;
        __builtin_amdgcn_wave_barrier(); // code is rewritten by the next chunk
    }
}

// Sharded engine, exchange without a host round trip (orbit_scatter_list): this rank's items go to every rank's
// output buffer — peers' buffers are mapped over xGMI — at the position the all-gathered counts give them.  A plain
// dword copy: list items start 4 (commands) or 8 (visible items) bytes into their buffers and land at 28- or 8-byte
// multiples, so 4 B is the common alignment; a wave moves 4 x 256 contiguous bytes per trip, loads first.
// blockIdx.y = destination rank: the copies to all peers run side by side, each on its own link.
__global__ __launch_bounds__(256) void scatter_list_kernel(const ScatterListParams p) {
    const uint32_t dst_rank = blockIdx.y;
    uint32_t first = 0, total = 0;
    for (uint32_t r = 0; r < p.world; r++) {
        const uint32_t c = p.counts[r];
        if (r < p.rank) first += c;
        total += c;
    }
    const uint32_t mine = p.counts[p.rank];
    const uint32_t room = first < p.out_capacity ? p.out_capacity - first : 0u;
    const uint32_t n_items = min(mine, room);
    const uint64_t words = (uint64_t)n_items * p.stride / 4u;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p.local_list + p.header_bytes);
    uint32_t *dst = reinterpret_cast<uint32_t *>(p.out[dst_rank] + p.header_bytes + (uint64_t)first * p.stride);
    const uint64_t step = (uint64_t)gridDim.x * 256u * 4u;
    for (uint64_t w = ((uint64_t)blockIdx.x * 256u + (threadIdx.x & ~63u)) * 4u + (threadIdx.x & 63u); w < words;
         w += step) {
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = w + 64u * k < words ? src[w + 64u * k] : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (w + 64u * k < words) dst[w + 64u * k] = v[k];
    }
    if (dst_rank == p.rank && blockIdx.x == 0 && threadIdx.x == 0) {
        uint32_t *hdr = reinterpret_cast<uint32_t *>(p.out[p.rank]);
        hdr[0] = min(total, p.out_capacity);
        for (uint32_t k = 1; k < p.header_bytes / 4u; k++) hdr[k] = 0u;
        if (total > p.out_capacity) latch_status(p.status, ORBIT_E_CAPACITY);
    }
}

// orbit_meshlet_stream_update: Meshlet records -> the derived streams (thread = meshlet; 32 B read, 38 B written).
// A wave owns 64 consecutive, 64-aligned global meshlet indices, i.e. two whole words of link bits: bits of meshlets
// outside the updated range keep their value.  Link bit of meshlet i (kernels.h): i continues i - 1's chain.  It is
// re-derived for i in [first, first + count] — the meshlet behind the range chains from the range's last one; its own
// offsets (and those of the meshlet in front of the range) come from the stream's `cmd` copy, not from the buffer,
// of which only [first, first + count) may be read.
struct StreamBuildArrays {
    uint4 *sphere;
    uint32_t *cone;
    uint16_t *mat;
    uint32_t *cmd;
    uint16_t *cnt;
    uint32_t *link, *cls0, *cls1;
    uint64_t s_first, s_end; // the stream's range
    const OrbitMaterialData *materials; // alpha classes come from here (nullptr: class 3 everywhere)
    uint32_t material_count;
    uint32_t *other; // set to 1 when a meshlet of class 3 is derived while `materials` is set (abi.hip: such a stream
                     // evaluates through its material indices)
};

// alpha class of a meshlet (kernels.h MeshletStreamView): its material's alpha_mode where the two bit planes can hold
// it, 3 = "look the material up" otherwise
__device__ __forceinline__ uint32_t alpha_class(const StreamBuildArrays &A, uint32_t material_index) {
    if (A.materials == nullptr || material_index >= A.material_count) return 3u;
    const uint32_t mode = A.materials[material_index].alpha_mode;
    return mode < 3u ? mode : 3u;
}

__global__ __launch_bounds__(256) void meshlet_stream_build_kernel(const OrbitMeshlet *meshlets, uint64_t first,
                                                                   uint64_t count, const StreamBuildArrays A) {
    const int lane = threadIdx.x & 63;
    const uint64_t lo = first & ~63ull, end = first + count;
    for (uint64_t base = lo + ((uint64_t)blockIdx.x * 256u + (threadIdx.x & ~63u)); base <= end;
         base += (uint64_t)gridDim.x * 256u) {
        const uint64_t m = base + (uint64_t)lane;
        const bool in = m >= first && m < end;
        uint32_t vtx = 0, data = 0, cls = 3u;
        if (in) {
            const uint4 *src = reinterpret_cast<const uint4 *>(meshlets + m);
            const uint4 a = src[0], b = src[1];
            A.sphere[m] = a;
            A.cone[m] = b.x;
            A.mat[m] = (uint16_t)(b.w & 0xFFFFu);
            A.cmd[m * 3u] = b.y;
            A.cmd[m * 3u + 1u] = b.z;
            A.cmd[m * 3u + 2u] = b.w;
            A.cnt[m] = (uint16_t)(b.w >> 16);
            vtx = b.y, data = b.z;
            cls = alpha_class(A, b.w & 0xFFFFu);
            if (cls == 3u && A.materials != nullptr) *A.other = 1u;
        } else if (m == end && m < A.s_end) { // derived earlier (or never: then the bit is as arbitrary as the rest)
            vtx = A.cmd[m * 3u], data = A.cmd[m * 3u + 1u];
        }
        const bool in_stream = m >= A.s_first && m < A.s_end;
        bool bit = false;
        if (in_stream) bit = ((A.link[m >> 5] >> (m & 31u)) & 1u) != 0; // kept unless re-derived below
        if (in_stream && m >= first && m <= end) {
            bit = false;
            if (m > A.s_first) {
                uint32_t pv, pd, p7;
                if (m > first) { // predecessor inside the updated range: from the buffer
                    const uint4 pb = reinterpret_cast<const uint4 *>(meshlets + (m - 1u))[1];
                    pv = pb.y, pd = pb.z, p7 = pb.w;
                } else { // the meshlet in front of the range: the stream's own copy
                    pv = A.cmd[(m - 1u) * 3u], pd = A.cmd[(m - 1u) * 3u + 1u], p7 = A.cmd[(m - 1u) * 3u + 2u];
                }
                bit = vtx == pv && data == pd + meshlet_data_words((p7 >> 16) & 0xFFu, p7 >> 24);
            }
        }
        // the class planes: meshlets outside the updated range keep their bits
        bool p0 = (cls & 1u) != 0u, p1 = (cls & 2u) != 0u;
        if (!in && in_stream) {
            p0 = ((A.cls0[m >> 5] >> (m & 31u)) & 1u) != 0;
            p1 = ((A.cls1[m >> 5] >> (m & 31u)) & 1u) != 0;
        }
        const uint64_t mask = __ballot(bit), m0 = __ballot(p0), m1 = __ballot(p1);
        // both words exist whenever any of their meshlets is in the stream (orbit_meshlet_stream_create)
        if (lane == 0 && base + 32u > A.s_first && base < A.s_end) {
            A.link[base >> 5] = (uint32_t)mask;
            A.cls0[base >> 5] = (uint32_t)m0;
            A.cls1[base >> 5] = (uint32_t)m1;
        }
        if (lane == 32 && base + 64u > A.s_first && base + 32u < A.s_end) {
            A.link[(base >> 5) + 1u] = (uint32_t)(mask >> 32);
            A.cls0[(base >> 5) + 1u] = (uint32_t)(m0 >> 32);
            A.cls1[(base >> 5) + 1u] = (uint32_t)(m1 >> 32);
        }
    }
}

// orbit_meshlet_stream_set_materials: the class planes of the whole stream from its material indices
__global__ __launch_bounds__(256) void meshlet_stream_classes_kernel(const StreamBuildArrays A) {
    const int lane = threadIdx.x & 63;
    for (uint64_t base = (A.s_first & ~63ull) + ((uint64_t)blockIdx.x * 256u + (threadIdx.x & ~63u)); base < A.s_end;
         base += (uint64_t)gridDim.x * 256u) {
        const uint64_t m = base + (uint64_t)lane;
        const bool in = m >= A.s_first && m < A.s_end;
        const uint32_t cls = in ? alpha_class(A, (uint32_t)A.mat[m]) : 3u;
        if (in && cls == 3u && A.materials != nullptr) *A.other = 1u;
        const uint64_t m0 = __ballot((cls & 1u) != 0u), m1 = __ballot((cls & 2u) != 0u);
        if (lane == 0 && base + 32u > A.s_first) {
            A.cls0[base >> 5] = (uint32_t)m0;
            A.cls1[base >> 5] = (uint32_t)m1;
        }
        if (lane == 32 && base + 32u < A.s_end) {
            A.cls0[(base >> 5) + 1u] = (uint32_t)(m0 >> 32);
            A.cls1[(base >> 5) + 1u] = (uint32_t)(m1 >> 32);
        }
    }
}

// Mesh-shading path: one OrbitMeshTaskRecord per dispatch record from the record's should-draw ballot
// (forward_depth_prepass.task:225-228,255-257).  Thread = record; the drawn lanes in ascending order.
__global__ __launch_bounds__(256) void task_records_kernel(const MeshletCullParams p) {
    const uint32_t nrec = min(*reinterpret_cast<const uint32_t *>(p.dispatch_buffer), p.dispatch_capacity);
    if (blockIdx.x == 0 && threadIdx.x < kTicketPools) p.tickets[threadIdx.x * kTicketStride] = 0u; // next evaluation
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= nrec) return;
    const uint4 rec = reinterpret_cast<const uint4 *>(p.dispatch_buffer + ORBIT_DISPATCH_HEADER)[r];
    uint32_t m = p.tile_masks[r]; // word r = tile r/16, record r%16
    const uint32_t count = (uint32_t)__popc(m);
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 32; k++) {
        if (m != 0u) {
            w[k / 4] |= (uint32_t)__builtin_ctz(m) << (8 * (k % 4));
            m &= m - 1u;
        }
    }
    uint32_t *out = reinterpret_cast<uint32_t *>(p.task_records + r);
    out[0] = count;
    out[1] = rec.x; // entity_index
    out[2] = rec.y; // meshlet_offset
#pragma unroll
    for (int k = 0; k < 8; k++) out[3 + k] = w[k];
}

template <int PASS>
hipError_t launch_pass(const MeshletCullParams &p, uint32_t num_cus, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    const uint32_t max_tiles = (p.dispatch_capacity + kTileRecords - 1) / kTileRecords;
    // nothing is enqueued for a capacity the emit launch cannot address (orbit_ctx_create rejects such caps already)
    if ((max_tiles + kScanChunk - 1) / kScanChunk > kEmitChunkTable) return hipErrorInvalidValue;
    // eval -> scan -> emit
    // 4-wave blocks, one wave per SIMD each.  The fifth wave per SIMD pays once a wave has enough tiles to stream (8 and
    // more); a small launch is mostly its pipeline's ramp, which more, shorter waves only lengthen (a 1/4 and a 1/8
    // shard of config 5: 0.095 / 0.085 ms at four waves, 0.1025 / 0.090 at five)
    uint32_t wps = (uint32_t)ev_waves_per_simd(PASS, p.ms.sphere != nullptr);
    if (wps > (uint32_t)kEvWavesPerSimd && max_tiles < 8u * num_cus * wps * (uint32_t)kEvWaves) wps = (uint32_t)kEvWavesPerSimd;
    uint32_t blocks = num_cus * wps;
    const uint32_t need = (max_tiles + kEvWaves - 1) / kEvWaves;
    if (blocks > need) blocks = need;
    if (blocks == 0) blocks = 1;
    if (ev0) {
        const hipError_t ee = hipEventRecord(ev0, s);
        if (ee != hipSuccess) return ee;
    }
    const dim3 grid(blocks), block(kEvWaves * 64);
    const bool ortho = p.ci.projection_type == 1;
    if constexpr (PASS == 1) { // list-driven: only last frame's visible meshlets are read
        if (ortho)
            hipLaunchKernelGGL((meshlet_pass1_kernel<1>), grid, block, 0, s, p);
        else
            hipLaunchKernelGGL((meshlet_pass1_kernel<0>), grid, block, 0, s, p);
    } else if (p.ms.sphere != nullptr && p.ms.cls0 != nullptr) { // derived streams + alpha classes: 20.25 B per meshlet
        if (ortho)
            hipLaunchKernelGGL((meshlet_eval_kernel<PASS, 1, false, true, true>), grid, block, 0, s, p);
        else
            hipLaunchKernelGGL((meshlet_eval_kernel<PASS, 0, false, true, true>), grid, block, 0, s, p);
    } else if (p.ms.sphere != nullptr) { // derived streams: 22 B per meshlet, no payload
        if (ortho)
            hipLaunchKernelGGL((meshlet_eval_kernel<PASS, 1, false, true>), grid, block, 0, s, p);
        else
            hipLaunchKernelGGL((meshlet_eval_kernel<PASS, 0, false, true>), grid, block, 0, s, p);
    } else if constexpr (PASS == 0) {
#ifdef ORBIT_TRIAGE
        if (p.debug_flags & 8u) { // phase-stamped kernel (tools/phase.py); | 16: only the per-wave wall-clock stamps
            if (ortho)
                hipLaunchKernelGGL((meshlet_eval_kernel<0, 1, true>), grid, block, 0, s, p);
            else
                hipLaunchKernelGGL((meshlet_eval_kernel<0, 0, true>), grid, block, 0, s, p);
        } else
#endif
        if (ortho) {
            hipLaunchKernelGGL((meshlet_eval_kernel<0, 1, false>), grid, block, 0, s, p);
        } else {
            hipLaunchKernelGGL((meshlet_eval_kernel<0, 0, false>), grid, block, 0, s, p);
        }
    } else {
        if (ortho)
            hipLaunchKernelGGL((meshlet_eval_kernel<PASS, 1, false>), grid, block, 0, s, p);
        else
            hipLaunchKernelGGL((meshlet_eval_kernel<PASS, 0, false>), grid, block, 0, s, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (ev1) {
        e = hipEventRecord(ev1, s);
        if (e != hipSuccess) return e;
    }
    if (p.task_records) { // mesh-shading path: expand the per-record ballots, no draw list
        const uint32_t tblocks = (p.dispatch_capacity + 255u) / 256u;
        if (tblocks) hipLaunchKernelGGL(task_records_kernel, dim3(tblocks), dim3(256), 0, s, p);
        return hipGetLastError();
    }
    if (p.visible_list == 2u) { // the visible list at record granularity: straight from the per-record ballots
        const uint32_t vblocks = max((p.dispatch_capacity + kVrBlock - 1u) / kVrBlock, 1u);
        hipLaunchKernelGGL(visible_records_count_kernel, dim3(vblocks), dim3(256), 0, s, p, vblocks);
        hipLaunchKernelGGL(visible_records_write_kernel, dim3(vblocks), dim3(256), 0, s, p, vblocks);
        return hipGetLastError();
    }
    // first scan level only (chunk-local prefixes + chunk sums); meshlet_emit redoes the second in LDS
    e = launch_scan_chunks(p.tile_counts, p.tile_base, reinterpret_cast<const uint32_t *>(p.dispatch_buffer),
                           kTileRecords, max_tiles, p.chunk_sums, s, p.tickets, kTicketPools * kTicketStride);
    if (e != hipSuccess) return e;
    uint32_t eblocks = num_cus * 8u;
    const uint32_t eneed = (max_tiles + 3) / 4;
    if (eblocks > eneed) eblocks = eneed;
    if (eblocks == 0) eblocks = 1;
    const bool stream = PASS != 1 && p.ms.sphere != nullptr; // the list-driven pass 1 keeps its payload
    if (p.visible_list) {
        if (stream) hipLaunchKernelGGL((meshlet_emit_kernel<true, true>), dim3(eblocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((meshlet_emit_kernel<true, false>), dim3(eblocks), dim3(256), 0, s, p);
    } else if (stream) {
        // a wave per group of two tiles; never more than 32 * kSlowWords groups per wave
        const uint32_t gneed = (eneed + kGroupTiles - 1) / kGroupTiles;
        const uint32_t cblocks = max(max(min(num_cus * (uint32_t)kChainWavesPerSimd, gneed), (gneed + 32u * kSlowWords - 1u) / (32u * kSlowWords)), 1u);
        hipLaunchKernelGGL(meshlet_emit_chain_kernel, dim3(cblocks), dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL((meshlet_emit_kernel<false, false>), dim3(eblocks), dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

} // namespace

// orbit_cull_views, meshlet stage.  The evaluation launches are grouped by (occlusion pass, projection) — a frame's
// early pass and its four cascades are two launches — scan and emit take all views at once.
hipError_t launch_meshlet_cull_views(const MeshletCullViews &all, uint32_t n, uint32_t num_cus, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const MeshletCullParams *host = all.v;
    uint32_t max_tiles_all = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t t = (host[i].dispatch_capacity + kTileRecords - 1) / kTileRecords;
        if ((t + kScanChunk - 1) / kScanChunk > kEmitChunkTable) return hipErrorInvalidValue;
        max_tiles_all = max(max_tiles_all, t);
    }
    // a view evaluates from the derived meshlet streams when its block carries them and its pass streams (0 / 2):
    // 0 = Meshlet buffer, 1 = streams, 2 = streams with alpha classes
    auto source = [&](uint32_t i) -> int {
        if (host[i].ci.occlusion_pass == 1 || host[i].ms.sphere == nullptr) return 0;
        return host[i].ms.cls0 != nullptr ? 2 : 1;
    };
    auto from_stream = [&](uint32_t i) { return source(i) != 0; };
    bool done[ORBIT_MAX_CULL_VIEWS] = {};
    for (uint32_t i = 0; i < n; i++) {
        if (done[i]) continue;
        const uint32_t pass = host[i].ci.occlusion_pass, proj = host[i].ci.projection_type;
        const int src = source(i);
        ViewGroup g{};
        uint32_t m = 0, need = 1;
        for (uint32_t j = i; j < n; j++)
            if (!done[j] && host[j].ci.occlusion_pass == pass && host[j].ci.projection_type == proj && source(j) == src) {
                done[j] = true;
                g.idx[m++] = (uint8_t)j;
                const uint32_t t = (host[j].dispatch_capacity + kTileRecords - 1) / kTileRecords;
                need = max(need, (t + kEvWaves - 1) / kEvWaves);
            }
        // the views of a group share the device: each gets an equal part of the resident-wave budget
        uint32_t blocks = max(num_cus * (uint32_t)ev_waves_per_simd((int)pass, src != 0) / m, 1u);
        if (blocks > need) blocks = need;
        const dim3 grid(blocks, m), block(kEvWaves * 64);
        const bool ortho = proj == 1;
#define ORBIT_EVAL_VIEWS(PASS)                                                                                        \
    if (src == 2) {                                                                                                   \
        if (ortho) hipLaunchKernelGGL((meshlet_eval_views_kernel<PASS, 1, true, true>), grid, block, 0, s, all, g);   \
        else hipLaunchKernelGGL((meshlet_eval_views_kernel<PASS, 0, true, true>), grid, block, 0, s, all, g);         \
    } else if (src == 1) {                                                                                            \
        if (ortho) hipLaunchKernelGGL((meshlet_eval_views_kernel<PASS, 1, true>), grid, block, 0, s, all, g);         \
        else hipLaunchKernelGGL((meshlet_eval_views_kernel<PASS, 0, true>), grid, block, 0, s, all, g);               \
    } else {                                                                                                          \
        if (ortho) hipLaunchKernelGGL((meshlet_eval_views_kernel<PASS, 1, false>), grid, block, 0, s, all, g);        \
        else hipLaunchKernelGGL((meshlet_eval_views_kernel<PASS, 0, false>), grid, block, 0, s, all, g);              \
    }
        if (pass == 1) {
            if (ortho) hipLaunchKernelGGL((meshlet_pass1_views_kernel<1>), grid, block, 0, s, all, g);
            else hipLaunchKernelGGL((meshlet_pass1_views_kernel<0>), grid, block, 0, s, all, g);
        } else if (pass == 0) {
            ORBIT_EVAL_VIEWS(0)
        } else {
            ORBIT_EVAL_VIEWS(2)
        }
#undef ORBIT_EVAL_VIEWS
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    ScanViews sv{};
    for (uint32_t i = 0; i < n; i++)
        sv.v[i] = {host[i].tile_counts, host[i].tile_base, reinterpret_cast<const uint32_t *>(host[i].dispatch_buffer),
                   host[i].chunk_sums, host[i].tickets,
                   (host[i].dispatch_capacity + kTileRecords - 1) / kTileRecords};
    hipError_t e = launch_scan_chunks_views(sv, n, kTileRecords, max_tiles_all, kTicketPools * kTicketStride, s);
    if (e != hipSuccess) return e;
    uint32_t eblocks = max(num_cus * 8u / n, 1u);
    const uint32_t eneed = (max_tiles_all + 3) / 4;
    if (eblocks > eneed) eblocks = eneed;
    if (eblocks == 0) eblocks = 1;
    // the emit launch of the views whose evaluation kept a payload, and of those that kept only ballots
    for (int stream = 0; stream < 2; stream++) {
        ViewGroup g{};
        uint32_t m = 0;
        for (uint32_t i = 0; i < n; i++)
            if (from_stream(i) == (stream != 0)) g.idx[m++] = (uint8_t)i;
        if (m == 0) continue;
        if (stream)
            hipLaunchKernelGGL(meshlet_emit_chain_views_kernel,
                               dim3(max(max(min(num_cus * (uint32_t)kChainWavesPerSimd / m, (eneed + kGroupTiles - 1) / kGroupTiles),
                                            ((eneed + kGroupTiles - 1) / kGroupTiles + 32u * kSlowWords - 1u) / (32u * kSlowWords)), 1u), m),
                               dim3(256), 0, s, all, g);
        else hipLaunchKernelGGL(meshlet_emit_views_kernel<false>, dim3(eblocks, m), dim3(256), 0, s, all, g);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_scatter_list(const ScatterListParams &p, uint32_t num_cus, hipStream_t s) {
    // enough blocks per destination to keep a link busy, few enough that `world` copies share the device
    const uint32_t per_dst = max(num_cus * 2u / max(p.world, 1u), 8u);
    hipLaunchKernelGGL(scatter_list_kernel, dim3(per_dst, p.world), dim3(256), 0, s, p);
    return hipGetLastError();
}

static StreamBuildArrays stream_build_arrays(const MeshletStreamView &ms, const OrbitMaterialData *materials,
                                             uint32_t material_count, uint32_t *other = nullptr) {
    return StreamBuildArrays{const_cast<uint4 *>(ms.sphere),   const_cast<uint32_t *>(ms.cone),
                             const_cast<uint16_t *>(ms.mat),   const_cast<uint32_t *>(ms.cmd),
                             const_cast<uint16_t *>(ms.cnt),   const_cast<uint32_t *>(ms.link),
                             const_cast<uint32_t *>(ms.cls0),  const_cast<uint32_t *>(ms.cls1),
                             (uint64_t)ms.first,               (uint64_t)ms.first + ms.count,
                             materials,                        material_count,
                             other};
}

hipError_t launch_meshlet_stream_build(const OrbitMeshlet *meshlets, uint64_t first, uint64_t count,
                                       const MeshletStreamView &ms, const OrbitMaterialData *materials,
                                       uint32_t material_count, uint32_t *other, hipStream_t s) {
    if (count == 0) return hipSuccess;
    const uint64_t need = (count + 64u + 255u) / 256u + 1u;
    const uint32_t blocks = (uint32_t)(need < 8192u ? need : 8192u);
    hipLaunchKernelGGL(meshlet_stream_build_kernel, dim3(blocks), dim3(256), 0, s, meshlets, first, count,
                       stream_build_arrays(ms, materials, material_count, other));
    return hipGetLastError();
}

// orbit_meshlet_stream_validate: thread = meshlet of the stream's range; the Meshlet as the arrays hold it against the
// Meshlet in the buffer, the alpha class against the material's alpha_mode.
__global__ __launch_bounds__(256) void meshlet_stream_validate_kernel(const OrbitMeshlet *meshlets, const StreamBuildArrays A,
                                                                      int32_t *status) {
    for (uint64_t m = A.s_first + (uint64_t)blockIdx.x * 256u + threadIdx.x; m < A.s_end; m += (uint64_t)gridDim.x * 256u) {
        const uint4 *src = reinterpret_cast<const uint4 *>(meshlets + m);
        const uint4 a = src[0], b = src[1], sa = A.sphere[m];
        bool same = a.x == sa.x && a.y == sa.y && a.z == sa.z && a.w == sa.w && b.x == A.cone[m] &&
                    (b.w & 0xFFFFu) == (uint32_t)A.mat[m] && b.y == A.cmd[m * 3u] && b.z == A.cmd[m * 3u + 1u] &&
                    b.w == A.cmd[m * 3u + 2u] && (b.w >> 16) == (uint32_t)A.cnt[m];
        if (A.materials) {
            const uint32_t cls = ((A.cls0[m >> 5] >> (m & 31u)) & 1u) | (((A.cls1[m >> 5] >> (m & 31u)) & 1u) << 1);
            same = same && cls == alpha_class(A, b.w & 0xFFFFu);
        }
        if (m > A.s_first) { // the link bit against the two meshlets it relates
            const uint4 pb = reinterpret_cast<const uint4 *>(meshlets + (m - 1u))[1];
            const bool link = b.y == pb.y && b.z == pb.z + meshlet_data_words((pb.w >> 16) & 0xFFu, pb.w >> 24);
            same = same && link == (((A.link[m >> 5] >> (m & 31u)) & 1u) != 0u);
        }
        if (!same) latch_status(status, ORBIT_E_STALE);
    }
}

hipError_t launch_meshlet_stream_validate(const OrbitMeshlet *meshlets, const MeshletStreamView &ms,
                                          const OrbitMaterialData *materials, uint32_t material_count, int32_t *status,
                                          hipStream_t s) {
    if (ms.count == 0) return hipSuccess;
    const uint64_t need = ((uint64_t)ms.count + 255u) / 256u;
    const uint32_t blocks = (uint32_t)(need < 8192u ? need : 8192u);
    hipLaunchKernelGGL(meshlet_stream_validate_kernel, dim3(blocks), dim3(256), 0, s, meshlets,
                       stream_build_arrays(ms, materials, material_count), status);
    return hipGetLastError();
}

hipError_t launch_meshlet_stream_classes(const MeshletStreamView &ms, const OrbitMaterialData *materials,
                                         uint32_t material_count, uint32_t *other, hipStream_t s) {
    const uint64_t need = ((uint64_t)ms.count + 64u + 255u) / 256u + 1u;
    const uint32_t blocks = (uint32_t)(need < 8192u ? need : 8192u);
    hipLaunchKernelGGL(meshlet_stream_classes_kernel, dim3(blocks), dim3(256), 0, s,
                       stream_build_arrays(ms, materials, material_count, other));
    return hipGetLastError();
}

uint32_t max_dispatch_capacity() {
    const uint64_t cap = (uint64_t)kEmitChunkTable * kScanChunk * kTileRecords;
    return cap > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cap;
}

hipError_t launch_visible_expand(const uint8_t *visible, const OrbitMeshlet *meshlets, const MeshletStreamView &ms,
                                 uint8_t *draw_buffer, uint32_t draw_capacity, uint32_t num_cus, const void *zero_page,
                                 int32_t *status, hipStream_t s) {
    uint32_t blocks = num_cus * 8u;
    const uint32_t need = (draw_capacity + 255u) / 256u;
    if (blocks > need) blocks = need;
    if (blocks == 0) blocks = 1;
    if (ms.cmd)
        hipLaunchKernelGGL(visible_expand_kernel<true>, dim3(blocks), dim3(256), 0, s, visible, meshlets, ms,
                           draw_buffer, draw_capacity, zero_page, status);
    else
        hipLaunchKernelGGL(visible_expand_kernel<false>, dim3(blocks), dim3(256), 0, s, visible, meshlets, ms,
                           draw_buffer, draw_capacity, zero_page, status);
    return hipGetLastError();
}

hipError_t launch_visible_records_expand(const uint8_t *records, uint32_t *block_pop, uint32_t max_blocks,
                                         const OrbitMeshlet *meshlets, const MeshletStreamView &ms, uint8_t *draw_buffer,
                                         uint32_t draw_capacity, const void *zero_page, int32_t *status, hipStream_t s) {
    // a command list of draw_capacity entries has at most that many records behind it
    const uint32_t blocks = max(min((draw_capacity + kVrBlock - 1u) / kVrBlock, max_blocks), 1u);
    hipLaunchKernelGGL(visible_records_pop_kernel, dim3(blocks), dim3(256), 0, s, records, block_pop, max_blocks);
    if (ms.cmd)
        hipLaunchKernelGGL(visible_records_expand_kernel<true>, dim3(blocks), dim3(256), 0, s, records, block_pop,
                           max_blocks, meshlets, ms, draw_buffer, draw_capacity, zero_page, status);
    else
        hipLaunchKernelGGL(visible_records_expand_kernel<false>, dim3(blocks), dim3(256), 0, s, records, block_pop,
                           max_blocks, meshlets, ms, draw_buffer, draw_capacity, zero_page, status);
    return hipGetLastError();
}

hipError_t launch_meshlet_cull(const MeshletCullParams &p, uint32_t num_cus, hipStream_t s, hipEvent_t ev0,
                               hipEvent_t ev1) {
    switch (p.ci.occlusion_pass) {
    case 0: return launch_pass<0>(p, num_cus, s, ev0, ev1);
    case 1: return launch_pass<1>(p, num_cus, s, ev0, ev1);
    default: return launch_pass<2>(p, num_cus, s, ev0, ev1);
    }
}

} // namespace orbit
