// Micro-benchmark: is an XCD's streaming rate a function of WHICH addresses it reads?  In meshlet_eval a block b
// (XCD b % 8) reads the 64-KiB groups g with g % 8 == b % 8, always — if the memory system places 64-KiB groups on
// stacks / channels by those address bits, every XCD streams from a fixed subset of the HBM.  Here XCD x reads the
// groups with g % 8 == (x + shift) % 8 for shift = 0..7 (all XCDs at the same time), and the per-XCD median block
// finish time is printed: an 8 x 8 (XCD, residue) table under load.
// build: hipcc -O3 -w --offload-arch=gfx950 tools/ubench/xcd_numa.hip -o tools/ubench/xcd_numa.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int GROUP = 65536; // bytes

// MODE 0: residue = (xcd + shift) % 8 (fixed per launch); MODE 1: residue rotates with the step (xcd + step) % 8
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4 *src, size_t n_groups, int shift, unsigned long long *stamps,
                                         unsigned *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, nbx = gridDim.x >> 3; // block index inside its XCD
    unsigned acc = 0;
    const unsigned long long t0 = wall_clock64();
    // a block takes one 64-KiB group per step: 4 waves x 16 KiB, each wave 8 rows of 2 KiB (the meshlet_eval shape)
    size_t step = 0;
    for (size_t j = bx; j * 8 + 7 < n_groups; j += nbx, step++) {
        const unsigned res = MODE == 0 ? (xcd + shift) & 7 : (xcd + shift + (unsigned)step) & 7;
        const size_t g = j * 8 + res;
        const uint4 *p = src + (g * GROUP + (size_t)wave * 16384) / 16 + lane * 2;
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
    }
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = wall_clock64();
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const size_t bytes = 1600ull << 20;
    const size_t n_groups = bytes / GROUP;
    uint4 *src;
    unsigned *sink;
    unsigned long long *stamps;
    const int blocks = 1024;
    hipMalloc(&src, bytes);
    hipMalloc(&sink, 4);
    hipMalloc(&stamps, blocks * 16);
    hipMemset(src, 1, bytes);
    std::vector<unsigned long long> h(blocks * 2);
    for (int mode = 0; mode < 2; mode++) {
        printf(mode == 0 ? "fixed residue per XCD: rows = shift, columns = XCD 0..7: median block end (ticks) | kernel ms\n"
                         : "residue rotating per step:\n");
        for (int shift = 0; shift < (mode == 0 ? 8 : 2); shift++) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, src, n_groups, shift, stamps, sink);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, src, n_groups, shift, stamps, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), stamps, blocks * 16, hipMemcpyDeviceToHost);
            unsigned long long tmin = ~0ull;
            for (int b = 0; b < blocks; b++) tmin = std::min(tmin, h[2 * b]);
            printf("shift %d:", shift);
            for (int x = 0; x < 8; x++) {
                std::vector<double> e;
                for (int b = x; b < blocks; b += 8) e.push_back((double)(h[2 * b + 1] - tmin));
                std::sort(e.begin(), e.end());
                printf(" %6.0f", e[e.size() / 2]);
            }
            printf(" | %.4f ms (%.0f GB/s)\n", ms, bytes / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
