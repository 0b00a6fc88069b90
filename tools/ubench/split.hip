// Micro-benchmark: would a derived, split meshlet stream pay?  The 32-B Meshlet (types.glsl:124-134) is static asset
// data; the evaluation needs only its bounding sphere + cone (+ the word with the material index), the command words
// (vertex_offset, data_offset, counts) only for the ~10 % that survive.  Skeletons, 50 M meshlets, 9.5 % scattered
// survivors:
//   eval  AOS      : 32 B per meshlet read, 12-B payload per survivor written into the tile's slot (today's kernel)
//   eval  SPLIT<W> : 16 B (spheres) + W B (cone [+ word 7]) per meshlet read, 64 B of ballot masks per tile written
//   emit  PAYLOAD  : tile's payload run read, 28-B commands written at the tile's offset (today's launch)
//   emit  GATHER<S>: masks read, every survivor gathers S bytes (12: the derived command-word stream, 32: the
//                    original AoS buffer) and writes its command
// build: hipcc -O3 -w --offload-arch=gfx950 tools/ubench/split.hip -o tools/ubench/split.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr unsigned kKeep = 97; // of 1024

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ void fill(uint4 *aos, uint4 *a1, uint2 *a2, unsigned *b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned h = hash32((unsigned)i);
        aos[2 * i] = make_uint4(h, 1, 2, 3);
        aos[2 * i + 1] = make_uint4(0, (unsigned)i, (unsigned)i * 3, 0x40300000u);
        a1[i] = make_uint4(h, 1, 2, 3);
        a2[i] = make_uint2(0, 0x40300000u);
        b[3 * i] = (unsigned)i; b[3 * i + 1] = (unsigned)i * 3; b[3 * i + 2] = 0x40300000u;
    }
}

struct P3 { unsigned y, z, w; };

__device__ __forceinline__ unsigned lane_prefix(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

__global__ __launch_bounds__(256) void eval_aos(const uint4 *src, size_t n_tiles, P3 *pay, unsigned *counts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * 4;
    for (size_t t = (size_t)blockIdx.x * 4 + wave; t < n_tiles; t += stride) {
        const uint4 *p = src + (t * 512 + lane) * 2;
        uint4 a0 = p[0], b0 = p[1];
        unsigned total = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint4 a1 = a0, b1 = b0;
            if (r < 7) { a1 = p[(r + 1) * 128]; b1 = p[(r + 1) * 128 + 1]; }
            const bool keep = ((a0.x + (b0.x >> 31)) & 1023u) < kKeep;
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const unsigned rank = total + lane_prefix(m);
                if (rank < 128) pay[t * 128 + rank] = P3{b0.y, b0.z, b0.w | lane | (r << 6)};
            }
            total += __popcll(m);
            a0 = a1; b0 = b1;
        }
        if (lane == 0) counts[t] = total;
    }
}

// OUT 0: ballots (lanes 0..7, 8 B each) and the count (lane 0) as two stores per tile (the product);
//     1: one store: lanes 0..8 write 8 B each into a 72-B block (count in the ninth); 3: ballots only
template <int W, int OUT = 0>
__global__ __launch_bounds__(256) void eval_split(const uint4 *a1, const void *a2v, size_t n_tiles,
                                                  unsigned long long *masks, unsigned *counts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * 4;
    for (size_t t = (size_t)blockIdx.x * 4 + wave; t < n_tiles; t += stride) {
        const uint4 *p = a1 + t * 512 + lane;
        uint4 s0 = p[0];
        uint2 c0;
        if (W == 8) c0 = ((const uint2 *)a2v)[t * 512 + lane];
        else c0 = make_uint2(((const unsigned *)a2v)[t * 512 + lane], 0);
        unsigned total = 0;
        unsigned long long mine = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint4 s1 = s0; uint2 c1 = c0;
            if (r < 7) {
                s1 = p[(r + 1) * 64];
                if (W == 8) c1 = ((const uint2 *)a2v)[t * 512 + (r + 1) * 64 + lane];
                else c1 = make_uint2(((const unsigned *)a2v)[t * 512 + (r + 1) * 64 + lane], 0);
            }
            const bool keep = ((s0.x + (c0.x >> 31) + (c0.y & 1u)) & 1023u) < kKeep;
            const unsigned long long m = __ballot(keep);
            if (lane == r) mine = m;
            total += __popcll(m);
            s0 = s1; c0 = c1;
        }
        if (OUT == 0) {
            if (lane < 8) masks[t * 8 + lane] = mine;
            if (lane == 0) counts[t] = total;
        } else if (OUT == 1) {
            if (lane == 8) mine = total;
            if (lane < 9) masks[t * 9 + lane] = mine;
        } else if (OUT == 3) {
            if (lane < 8) masks[t * 8 + lane] = mine;
        }
    }
}

// WIDE: the tile's 512 cones (2 KiB) come in as two dwordx4 loads per lane issued one tile ahead instead of one
// dword load per row (8 per tile) — fewer, full-width requests for the same bytes.  (Bandwidth probe: the lane mapping
// a real kernel would need — through LDS — is not reproduced, the words only feed the predicate.)
__global__ __launch_bounds__(256) void eval_split_wide(const uint4 *a1, const uint4 *cones, size_t n_tiles,
                                                       unsigned long long *masks, unsigned *counts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * 4;
    size_t t = (size_t)blockIdx.x * 4 + wave;
    uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0;
    if (t < n_tiles) { c0 = cones[t * 128 + lane]; c1 = cones[t * 128 + 64 + lane]; }
    for (; t < n_tiles; t += stride) {
        const uint4 *p = a1 + t * 512 + lane;
        uint4 s0 = p[0];
        uint4 n0 = c0, n1 = c1;
        if (t + stride < n_tiles) { n0 = cones[(t + stride) * 128 + lane]; n1 = cones[(t + stride) * 128 + 64 + lane]; }
        const unsigned cw[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        unsigned total = 0;
        unsigned long long mine = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint4 s1 = s0;
            if (r < 7) s1 = p[(r + 1) * 64];
            const bool keep = ((s0.x + (cw[r] >> 31)) & 1023u) < kKeep;
            const unsigned long long m = __ballot(keep);
            if (lane == r) mine = m;
            total += __popcll(m);
            s0 = s1;
        }
        if (lane < 8) masks[t * 8 + lane] = mine;
        if (lane == 0) counts[t] = total;
        c0 = n0; c1 = n1;
    }
}

// AoSoA: the same 16 + 4 B per meshlet, but interleaved per group of 32 meshlets (512 B of spheres, then 128 B of
// cones: 640 B per group) instead of two arrays — one DRAM stream per wave instead of two.
__global__ __launch_bounds__(256) void eval_split_aosoa(const unsigned char *g, size_t n_tiles,
                                                        unsigned long long *masks, unsigned *counts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * 4;
    for (size_t t = (size_t)blockIdx.x * 4 + wave; t < n_tiles; t += stride) {
        // meshlet t*512 + r*64 + lane: group = t*16 + r*2 + lane/32, k = lane % 32
        const unsigned char *base = g + (t * 16 + (lane >> 5)) * 640 + (lane & 31) * 16;
        const unsigned char *cbase = g + (t * 16 + (lane >> 5)) * 640 + 512 + (lane & 31) * 4;
        uint4 s0 = *(const uint4 *)base;
        unsigned c0 = *(const unsigned *)cbase;
        unsigned total = 0;
        unsigned long long mine = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint4 s1 = s0;
            unsigned c1 = c0;
            if (r < 7) {
                s1 = *(const uint4 *)(base + (r + 1) * 1280);
                c1 = *(const unsigned *)(cbase + (r + 1) * 1280);
            }
            const bool keep = ((s0.x + (c0 >> 31)) & 1023u) < kKeep;
            const unsigned long long m = __ballot(keep);
            if (lane == r) mine = m;
            total += __popcll(m);
            s0 = s1; c0 = c1;
        }
        if (lane < 8) masks[t * 8 + lane] = mine;
        if (lane == 0) counts[t] = total;
    }
}

__global__ void fill_aosoa(unsigned char *g, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t grp = i >> 5, k = i & 31;
        *(uint4 *)(g + grp * 640 + k * 16) = make_uint4(hash32((unsigned)i), 1, 2, 3);
        *(unsigned *)(g + grp * 640 + 512 + k * 4) = 0u;
    }
}

struct Cmd { unsigned w[7]; };

__device__ __forceinline__ void put_cmd(unsigned *out, size_t slot, unsigned y, unsigned z, unsigned w, unsigned idx) {
    unsigned *o = out + slot * 7;
    o[0] = (w >> 24) * 3; o[1] = 1; o[2] = z * 4 + ((w >> 16) & 255u); o[3] = y; o[4] = 0; o[5] = y; o[6] = idx;
}

__global__ __launch_bounds__(256) void emit_payload(const P3 *pay, const unsigned *counts, const unsigned *offs,
                                                    size_t n_tiles, unsigned *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * 4;
    for (size_t t = (size_t)blockIdx.x * 4 + wave; t < n_tiles; t += stride) {
        const unsigned n = counts[t], off = offs[t];
        for (unsigned j = lane; j < n && j < 128; j += 64) {
            const P3 v = pay[t * 128 + j];
            put_cmd(out, off + j, v.y, v.z, v.w, (unsigned)t * 512 + (v.w & 511u));
        }
    }
}

template <int S>
__global__ __launch_bounds__(256) void emit_gather(const unsigned long long *masks, const unsigned *offs,
                                                   const void *src, size_t n_tiles, unsigned *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * 4;
    for (size_t t = (size_t)blockIdx.x * 4 + wave; t < n_tiles; t += stride) {
        const unsigned off = offs[t];
        const unsigned long long mine = lane < 8 ? masks[t * 8 + lane] : 0ull;
        unsigned total = 0;
        // all eight rows' gathers are issued before any command is stored
        unsigned y[8], z[8], w[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const unsigned long long m = __shfl(mine, r, 64);
            if ((m >> lane) & 1ull) {
                const size_t i = t * 512 + r * 64 + lane;
                if (S == 12) {
                    const unsigned *b = (const unsigned *)src + 3 * i;
                    y[r] = b[0]; z[r] = b[1]; w[r] = b[2];
                } else {
                    const uint4 v = ((const uint4 *)src)[2 * i + 1];
                    y[r] = v.y; z[r] = v.z; w[r] = v.w;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const unsigned long long m = __shfl(mine, r, 64);
            if ((m >> lane) & 1ull)
                put_cmd(out, off + total + lane_prefix(m), y[r], z[r], w[r], (unsigned)(t * 512 + r * 64 + lane));
            total += __popcll(m);
        }
    }
}

template <class F> static float time_kernel(F &&launch, int reps = 10) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const size_t n_tiles = 97657, n = n_tiles * 512;
    uint4 *aos, *a1; uint2 *a2; unsigned *a2n, *b, *counts, *offs, *out; P3 *pay; unsigned long long *masks;
    hipMalloc(&aos, n * 32); hipMalloc(&a1, n * 16); hipMalloc(&a2, n * 8); hipMalloc(&a2n, n * 4);
    hipMalloc(&b, n * 12); hipMalloc(&counts, n_tiles * 4); hipMalloc(&offs, n_tiles * 4);
    hipMalloc(&pay, n_tiles * 128 * 12); hipMalloc(&masks, n_tiles * 72); hipMalloc(&out, (n / 8) * 28);
    hipMemset(a2n, 0, n * 4);
    unsigned char *gso;
    hipMalloc(&gso, n * 20);
    hipLaunchKernelGGL(fill_aosoa, dim3(4096), dim3(256), 0, 0, gso, n);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, aos, a1, a2, b, n);
    const int blocks = 256 * 4;
    hipLaunchKernelGGL(eval_aos, dim3(blocks), dim3(256), 0, 0, aos, n_tiles, pay, counts);
    std::vector<unsigned> hc(n_tiles), ho(n_tiles);
    hipMemcpy(hc.data(), counts, n_tiles * 4, hipMemcpyDeviceToHost);
    size_t tot = 0; unsigned mx = 0;
    for (size_t i = 0; i < n_tiles; i++) { ho[i] = (unsigned)tot; tot += hc[i]; if (hc[i] > mx) mx = hc[i]; }
    hipMemcpy(offs, ho.data(), n_tiles * 4, hipMemcpyHostToDevice);
    printf("meshlets %zu survivors %zu (%.2f %%) max per tile %u\n", n, tot, 100.0 * tot / n, mx);
    // same commands from every emit variant?
    std::vector<unsigned> ref(tot * 7), got(tot * 7);
    hipLaunchKernelGGL((eval_split<8>), dim3(blocks), dim3(256), 0, 0, a1, a2, n_tiles, masks, counts);
    hipLaunchKernelGGL(emit_payload, dim3(blocks * 2), dim3(256), 0, 0, pay, counts, offs, n_tiles, out);
    hipMemcpy(ref.data(), out, tot * 28, hipMemcpyDeviceToHost);
    hipMemset(out, 0, tot * 28);
    hipLaunchKernelGGL((emit_gather<12>), dim3(blocks * 2), dim3(256), 0, 0, masks, offs, b, n_tiles, out);
    hipMemcpy(got.data(), out, tot * 28, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < tot * 7; i++) bad += (ref[i] != got[i]);
    printf("gather<12> vs payload commands: %zu differing words\n", bad);
    for (int rep = 0; rep < 3; rep++) {
        const float ea = time_kernel([&] { hipLaunchKernelGGL(eval_aos, dim3(blocks), dim3(256), 0, 0, aos, n_tiles, pay, counts); });
        const float e8 = time_kernel([&] { hipLaunchKernelGGL((eval_split<8>), dim3(blocks), dim3(256), 0, 0, a1, a2, n_tiles, masks, counts); });
        const float e4 = time_kernel([&] { hipLaunchKernelGGL((eval_split<4>), dim3(blocks), dim3(256), 0, 0, a1, a2n, n_tiles, masks, counts); });
        const float o1 = time_kernel([&] { hipLaunchKernelGGL((eval_split<4, 1>), dim3(blocks), dim3(256), 0, 0, a1, a2n, n_tiles, masks, counts); });
        const float o3 = time_kernel([&] { hipLaunchKernelGGL((eval_split<4, 3>), dim3(blocks), dim3(256), 0, 0, a1, a2n, n_tiles, masks, counts); });
        printf("split 16+4, outputs per tile: ballots + count (2 stores) %.4f | one 72-B store %.4f | ballots only %.4f ms\n", e4, o1, o3);
        const float ew = time_kernel([&] { hipLaunchKernelGGL(eval_split_wide, dim3(blocks), dim3(256), 0, 0, a1, (const uint4 *)a2n, n_tiles, masks, counts); });
        printf("split 16+4: cones as one dword load per row %.4f | as two dwordx4 loads per tile %.4f ms\n", e4, ew);
        const float ea2 = time_kernel([&] { hipLaunchKernelGGL(eval_split_aosoa, dim3(blocks), dim3(256), 0, 0, gso, n_tiles, masks, counts); });
        printf("split 16+4: two arrays %.4f | interleaved per 32 meshlets (640-B groups) %.4f ms\n", e4, ea2);
        float mp[3], g12[3], g32[3];
        const int grids[3] = {blocks, blocks * 2, blocks * 4};
        for (int g = 0; g < 3; g++) {
            mp[g] = time_kernel([&] { hipLaunchKernelGGL(emit_payload, dim3(grids[g]), dim3(256), 0, 0, pay, counts, offs, n_tiles, out); });
            g12[g] = time_kernel([&] { hipLaunchKernelGGL((emit_gather<12>), dim3(grids[g]), dim3(256), 0, 0, masks, offs, b, n_tiles, out); });
            g32[g] = time_kernel([&] { hipLaunchKernelGGL((emit_gather<32>), dim3(grids[g]), dim3(256), 0, 0, masks, offs, aos, n_tiles, out); });
        }
        printf("eval: aos+payload %.4f | split 16+8 %.4f | split 16+4 %.4f ms || emit (grid x4/x8/x16 blocks per CU): payload %.4f %.4f %.4f | gather 12 B %.4f %.4f %.4f | gather AoS %.4f %.4f %.4f ms\n",
               ea, e8, e4, mp[0], mp[1], mp[2], g12[0], g12[1], g12[2], g32[0], g32[1], g32[2]);
        printf("  sums: today %.4f | 16+8 & 12-B gather %.4f | 16+4 & 12-B gather %.4f | 16+8 & AoS gather %.4f ms\n",
               ea + mp[1], e8 + g12[1], e4 + g12[1], e8 + g32[1]);
    }
    return 0;
}
