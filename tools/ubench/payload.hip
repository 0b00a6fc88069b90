// Micro-benchmark: WHERE the survivor payload goes.  The load/store skeleton of meshlet_eval (tile-order stream of
// 1.6 GB + 784 B of payload per 16-KiB tile) with the payload written to different placements — is the cost of the
// payload (12-17 % of the read rate for 4.8 % of the bytes, profiles/r01_notes.md session 5) a property of the bytes
// or of their placement (4096 sparse write fronts, one per wave)?
// build: hipcc -O3 -w --offload-arch=gfx950 tools/ubench/payload.hip -o tools/ubench/payload.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// MODE 0: stream only
//      1: pay[t * 128 + lane]           2-KiB slot per tile, 784 B used (the kernel's layout)
//      2: pay[t * 49 + lane]            tiles packed back to back (dense 3.2-MB window over the running waves)
//      3: one global log, base from a returning atomicAdd per tile (issued at tile start, used at its end)
//      4: NLOG logs selected by blockIdx % NLOG
//      5: full 2-KiB slot written (128 entries): sparse vs dense at the slot level
//      6: like 1 but the tile's 784 B are written as 4 x 196 B... no: as 13 rows of 64 B by lanes 0..3 (serialised)
template <int MODE, int NLOG>
__global__ __launch_bounds__(256) void k(const uint4 *src, size_t n_rows, uint4 *pay, unsigned *ctr, size_t log_cap,
                                         unsigned *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * 4;
    const size_t n_tiles = n_rows / 8;
    unsigned acc = 0;
    const size_t wg = (size_t)blockIdx.x * 4 + wave;
    const unsigned lg = MODE == 4 ? blockIdx.x % NLOG : 0;
    for (size_t t = wg; t < n_tiles; t += stride) {
        unsigned base = 0;
        if (MODE == 3 || MODE == 4) {
            if (lane == 0) base = atomicAdd(&ctr[lg * 64], 49u);
        }
        const uint4 *p = src + (t * 8 * 64 + lane) * 2;
        uint4 a0 = p[0], b0 = p[1];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint4 a1 = a0, b1 = b0;
            if (r < 7) {
                a1 = p[(r + 1) * 128];
                b1 = p[(r + 1) * 128 + 1];
            }
            acc += a0.x ^ b0.w;
            a0 = a1;
            b0 = b1;
        }
        const uint4 v = make_uint4(acc, lane, 3, 7);
        if (MODE == 1 && lane < 49) pay[t * 128 + lane] = v;
        if (MODE == 2 && lane < 49) pay[t * 49 + lane] = v;
        if (MODE == 3 || MODE == 4) {
            base = (unsigned)__shfl((int)base, 0, 64);
            if (lane < 49) pay[(size_t)lg * log_cap + base + lane] = v;
        }
        if (MODE == 5) {
            pay[t * 128 + lane] = v;
            pay[t * 128 + 64 + lane] = v;
        }
        if (MODE == 6) { // NLOG tiles' payload as ONE contiguous run in the wave's own dense log, every NLOG-th tile
            const size_t it = (t - wg) / stride, per_wave = n_tiles / stride + 1;
            if ((it % NLOG) == NLOG - 1) {
                uint4 *dst = pay + (wg * per_wave + (it / NLOG) * NLOG) * 49;
                for (int j = lane; j < 49 * NLOG; j += 64) dst[j] = v;
            }
        }
        if (MODE == 7) { // 4-B entries (196 B per tile), NLOG tiles batched
            const size_t it = (t - wg) / stride, per_wave = n_tiles / stride + 1;
            if ((it % NLOG) == NLOG - 1) {
                unsigned *dst = reinterpret_cast<unsigned *>(pay) + (wg * per_wave + (it / NLOG) * NLOG) * 49;
                for (int j = lane; j < 49 * NLOG; j += 64) dst[j] = acc;
            }
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <typename K>
float time_kernel(K launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 8; i++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 8;
}

int main() {
    const size_t bytes = 1600ull << 20;
    const size_t n_rows = bytes / 2048, n_tiles = n_rows / 8;
    uint4 *src, *pay;
    unsigned *sink, *ctr;
    hipMalloc(&src, bytes);
    hipMalloc(&sink, 4);
    hipMalloc(&ctr, 64 * 256 * 4);
    hipMalloc(&pay, (n_tiles + 64) * 2048);
    hipMemset(src, 1, bytes);
    const int blocks = 256 * 4;
    auto gb = [&](float ms) { return bytes / (ms * 1e-3) / 1e9; };
#define RUN(MODE, NLOG) gb(time_kernel([&] { hipMemsetAsync(ctr, 0, 64 * 256 * 4, 0); \
    hipLaunchKernelGGL((k<MODE, NLOG>), dim3(blocks), dim3(256), 0, 0, src, n_rows, pay, ctr, (n_tiles * 49 + 4096) / NLOG + 4096, sink); }))
    for (int rep = 0; rep < 3; rep++) {
        printf("dense per-wave log, N tiles per flush (16-B entries): N=1 %.0f N=2 %.0f N=4 %.0f N=8 %.0f N=16 %.0f | 4-B entries: N=1 %.0f N=4 %.0f N=16 %.0f GB/s\n",
               RUN(6, 1), RUN(6, 2), RUN(6, 4), RUN(6, 8), RUN(6, 16), RUN(7, 1), RUN(7, 4), RUN(7, 16));
        printf("none %.0f | slot/tile %.0f | packed tiles %.0f | 1 log %.0f | 8 logs %.0f | 64 logs %.0f | 256 logs %.0f | full 2-KiB slot %.0f GB/s\n",
               RUN(0, 1), RUN(1, 1), RUN(2, 1), RUN(3, 1), RUN(4, 8), RUN(4, 64), RUN(4, 256), RUN(5, 1));
    }
    return 0;
}
