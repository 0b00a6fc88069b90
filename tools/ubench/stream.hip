// Micro-benchmark: HBM streaming read bandwidth on MI355X as a function of 16-B loads in flight per lane,
// with the meshlet_eval access pattern (lane reads 32 contiguous bytes = 2 x dwordx4, rows of 2 KiB per wave).
// build (here, cross-compiles): hipcc -O3 -w --offload-arch=gfx950 tools/ubench/stream.hip -o tools/ubench/stream.bin
// run (GPU box):                 ./tools/ubench/stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// LINEAR: the two loads of a row are two fully contiguous 1 KiB reads (lane * 16 B) instead of the
// record-interleaved pattern of meshlet_eval (lane * 32 B and lane * 32 B + 16)
template <int ROWS, bool LINEAR = false>  // rows (2 x dwordx4 per lane) in flight per wave
__global__ __launch_bounds__(256) void stream_kernel(const uint4 *src, size_t n_rows, unsigned *sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t stride = (size_t)gridDim.x * 4;
    unsigned acc = 0;
    // each "row" = 64 lanes x 32 B; a wave takes ROWS consecutive rows per step
    for (size_t r0 = wave * ROWS; r0 + ROWS <= n_rows; r0 += stride * ROWS) {
        uint4 a[ROWS], b[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; k++) {
            if (LINEAR) {
                const uint4 *p = src + (r0 + k) * 128 + lane;
                a[k] = p[0];
                b[k] = p[64];
            } else {
                const uint4 *p = src + ((r0 + k) * 64 + lane) * 2;
                a[k] = p[0];
                b[k] = p[1];
            }
        }
#pragma unroll
        for (int k = 0; k < ROWS; k++) acc += a[k].x ^ b[k].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// TILED: the access order of meshlet_eval — a wave walks the 8 consecutive rows of a 16-KiB tile one after the other
// (two row buffers), then jumps a whole grid stride of tiles; at any moment neighbouring waves touch addresses 16 KiB
// apart, not adjacent rows.
// ROTATE: wave w starts its tile at row w % 8 and wraps around, so that concurrently running waves are at different
// 2-KiB offsets inside their 16-KiB tiles.
template <bool ROTATE>
__global__ __launch_bounds__(256) void stream_tiled_kernel(const uint4 *src, size_t n_rows, unsigned *sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t stride = (size_t)gridDim.x * 4;
    const size_t n_tiles = n_rows / 8;
    unsigned acc = 0;
    for (size_t t = wave; t < n_tiles; t += stride) {
        const uint4 *p = src + (t * 8 * 64 + lane) * 2;
        const int r0 = ROTATE ? (int)(wave & 7) : 0;
        uint4 a0 = p[r0 * 128], b0 = p[r0 * 128 + 1];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            uint4 a1 = a0, b1 = b0;
            if (r < 7) {
                const int rr = (r0 + r + 1) & 7;
                a1 = p[rr * 128];
                b1 = p[rr * 128 + 1];
            }
            acc += a0.x ^ b0.w;
            a0 = a1;
            b0 = b1;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}


// BLOCK-TILE: the assignment a row-interleaved meshlet_eval would use — a block of 8 waves owns one 16-KiB tile per
// step (wave = row), blocks take tiles grid-stride, one block barrier per step (the slab hand-over), next tile's row
// in flight during the "evaluation" of the current one.  TR = rows per tile for the plain tile order (TR < 8 probes
// whether smaller wave tiles recover the row-interleaved rate).
template <bool BARRIER>
__global__ __launch_bounds__(512) void stream_blocktile_kernel(const uint4 *src, size_t n_rows, unsigned *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t n_tiles = n_rows / 8;
    unsigned acc = 0;
    size_t t = blockIdx.x;
    if (t >= n_tiles) return;
    const uint4 *p = src + ((t * 8 + wave) * 64 + lane) * 2;
    uint4 a0 = p[0], b0 = p[1];
    for (; t < n_tiles; t += gridDim.x) {
        const size_t tn = t + gridDim.x;
        uint4 a1 = a0, b1 = b0;
        if (tn < n_tiles) {
            const uint4 *q = src + ((tn * 8 + wave) * 64 + lane) * 2;
            a1 = q[0];
            b1 = q[1];
        }
        acc += a0.x ^ b0.w;
        if (BARRIER) __syncthreads();
        a0 = a1;
        b0 = b1;
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int TR>
__global__ __launch_bounds__(256) void stream_tiled_n_kernel(const uint4 *src, size_t n_rows, unsigned *sink) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t stride = (size_t)gridDim.x * 4;
    const size_t n_tiles = n_rows / TR;
    unsigned acc = 0;
    for (size_t t = wave; t < n_tiles; t += stride) {
        const uint4 *p = src + (t * TR * 64 + lane) * 2;
        uint4 a0 = p[0], b0 = p[1];
#pragma unroll
        for (int r = 0; r < TR; r++) {
            uint4 a1 = a0, b1 = b0;
            if (r < TR - 1) {
                a1 = p[(r + 1) * 128];
                b1 = p[(r + 1) * 128 + 1];
            }
            acc += a0.x ^ b0.w;
            a0 = a1;
            b0 = b1;
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}


// PAYLOAD: does a small store per tile beside the stream cost what meshlet_eval's survivor payload costs (8 %)?
//   MODE 0: tile-order stream only (as stream_tiled_n_kernel<8>)
//   MODE 1: + one 784-B store per tile (49 lanes x 16 B at tile * 2 KiB), issued by the streaming wave itself
//   MODE 2: the same bytes, issued by a fifth wave of the block that follows the streaming waves' progress through
//           an LDS counter (the streaming waves issue no store at all)
template <int MODE>
__global__ __launch_bounds__(320) void stream_payload_kernel(const uint4 *src, size_t n_rows, uint4 *pay, unsigned *sink) {
    __shared__ unsigned s_done[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t stride = (size_t)gridDim.x * 4;
    const size_t n_tiles = n_rows / 8;
    if (threadIdx.x < 4) s_done[threadIdx.x] = 0;
    __syncthreads();
    if (wave == 4) {
        if (MODE != 2) return;
        unsigned seen[4] = {0, 0, 0, 0};
        size_t t[4];
        for (int w = 0; w < 4; w++) t[w] = (size_t)blockIdx.x * 4 + w;
        bool live = true;
        while (live) {
            live = false;
            for (int w = 0; w < 4; w++) {
                if (t[w] >= n_tiles) continue;
                live = true;
                const unsigned d = *(volatile unsigned *)&s_done[w];
                while (seen[w] < d) {
                    if (lane < 49) pay[t[w] * 128 + lane] = make_uint4(lane, seen[w], d, 7);
                    seen[w]++;
                    t[w] += stride;
                }
            }
            __builtin_amdgcn_s_sleep(8);
        }
        return;
    }
    unsigned acc = 0;
    const size_t wg = (size_t)blockIdx.x * 4 + wave;
    for (size_t t = wg; t < n_tiles; t += stride) {
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
        if (MODE == 1 && lane < 49) pay[t * 128 + lane] = make_uint4(acc, lane, 3, 7);
        if (MODE == 2 && lane == 0) atomicAdd(&s_done[wave], 1u);
        if (MODE == 3 && lane < 49) { // non-temporal
            typedef unsigned v4 __attribute__((ext_vector_type(4)));
            v4 v = {acc, (unsigned)lane, 3u, 7u};
            __builtin_nontemporal_store(v, reinterpret_cast<v4 *>(pay + t * 128 + lane));
        }
        if (MODE == 4 && lane < 49) reinterpret_cast<unsigned *>(pay + t * 128)[lane] = acc; // 196 B per tile
        if (MODE == 5) { // the stores of 8 tiles issued together, every 8th tile
            const size_t it = (t - wg) / stride;
            if ((it & 7) == 7 && lane < 49)
                for (int k = 0; k < 8; k++) pay[(t - (size_t)k * stride) * 128 + lane] = make_uint4(acc, lane, k, 7);
        }
        if (MODE == 6 && lane < 49) { // dense: a wave's records back to back
            const size_t it = (t - wg) / stride, per_wave = n_tiles / stride + 1;
            pay[(wg * per_wave + it) * 49 + lane] = make_uint4(acc, lane, 3, 7);
        }
        if (MODE == 7 && lane < 49) pay[(t & 1023) * 128 + lane] = make_uint4(acc, lane, 3, 7); // 2 MB window: stays in L2
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <typename K>
float time_kernel(K launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

template <bool ROTATE>
float run_tiled(const uint4 *src, size_t n_rows, unsigned *sink, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(stream_tiled_kernel<ROTATE>, dim3(blocks), dim3(256), 0, 0, src, n_rows, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(stream_tiled_kernel<ROTATE>, dim3(blocks), dim3(256), 0, 0, src, n_rows, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

template <int ROWS, bool LINEAR = false>
float run(const uint4 *src, size_t n_rows, unsigned *sink, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((stream_kernel<ROWS, LINEAR>), dim3(blocks), dim3(256), 0, 0, src, n_rows, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((stream_kernel<ROWS, LINEAR>), dim3(blocks), dim3(256), 0, 0, src, n_rows, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main() {
    const size_t bytes = 1600ull << 20;
    const size_t n_rows = bytes / 2048;
    uint4 *src;
    unsigned *sink;
    hipMalloc(&src, bytes);
    hipMalloc(&sink, 4);
    hipMemset(src, 1, bytes);
    for (int bpc : {4, 5, 6, 8}) {
        const int blocks = 256 * bpc;
        printf("blocks/CU %d (waves/CU %d):", bpc, bpc * 4);
        printf(" rows=1 %.0f", bytes / (run<1>(src, n_rows, sink, blocks) * 1e-3) / 1e9);
        printf(" rows=2 %.0f", bytes / (run<2>(src, n_rows, sink, blocks) * 1e-3) / 1e9);
        printf(" rows=3 %.0f", bytes / (run<3>(src, n_rows, sink, blocks) * 1e-3) / 1e9);
        printf(" rows=4 %.0f", bytes / (run<4>(src, n_rows, sink, blocks) * 1e-3) / 1e9);
        printf(" rows=6 %.0f", bytes / (run<6>(src, n_rows, sink, blocks) * 1e-3) / 1e9);
        printf(" rows=8 %.0f GB/s\n", bytes / (run<8>(src, n_rows, sink, blocks) * 1e-3) / 1e9);
        printf("   tile order (8 rows of a 16-KiB tile per wave, then a grid stride): %.0f GB/s; rotated start row: %.0f GB/s\n",
               bytes / (run_tiled<false>(src, n_rows, sink, blocks) * 1e-3) / 1e9,
               bytes / (run_tiled<true>(src, n_rows, sink, blocks) * 1e-3) / 1e9);
        {
            const int b8 = blocks / 2; // 8-wave blocks, same waves per CU
            auto gb = [&](float ms) { return bytes / (ms * 1e-3) / 1e9; };
            printf("   block-tile (8-wave block = tile, wave = row): barrier %.0f, no barrier %.0f GB/s\n",
                   gb(time_kernel([&] { hipLaunchKernelGGL(stream_blocktile_kernel<true>, dim3(b8), dim3(512), 0, 0, src, n_rows, sink); })),
                   gb(time_kernel([&] { hipLaunchKernelGGL(stream_blocktile_kernel<false>, dim3(b8), dim3(512), 0, 0, src, n_rows, sink); })));
            printf("   tile order with TR rows per tile: TR=2 %.0f TR=4 %.0f TR=8 %.0f TR=16 %.0f GB/s\n",
                   gb(time_kernel([&] { hipLaunchKernelGGL(stream_tiled_n_kernel<2>, dim3(blocks), dim3(256), 0, 0, src, n_rows, sink); })),
                   gb(time_kernel([&] { hipLaunchKernelGGL(stream_tiled_n_kernel<4>, dim3(blocks), dim3(256), 0, 0, src, n_rows, sink); })),
                   gb(time_kernel([&] { hipLaunchKernelGGL(stream_tiled_n_kernel<8>, dim3(blocks), dim3(256), 0, 0, src, n_rows, sink); })),
                   gb(time_kernel([&] { hipLaunchKernelGGL(stream_tiled_n_kernel<16>, dim3(blocks), dim3(256), 0, 0, src, n_rows, sink); })));
        }
        if (bpc == 4) {
            uint4 *pay;
            hipMalloc(&pay, (bytes / 16384 + 1) * 2048);
            auto gb = [&](float ms) { return bytes / (ms * 1e-3) / 1e9; };
            printf("   + payload store per tile (784 B): none %.0f, own wave %.0f, fifth wave %.0f GB/s (read bytes only)\n",
                   gb(time_kernel([&] { hipLaunchKernelGGL(stream_payload_kernel<0>, dim3(blocks), dim3(320), 0, 0, src, n_rows, pay, sink); })),
                   gb(time_kernel([&] { hipLaunchKernelGGL(stream_payload_kernel<1>, dim3(blocks), dim3(320), 0, 0, src, n_rows, pay, sink); })),
                   gb(time_kernel([&] { hipLaunchKernelGGL(stream_payload_kernel<2>, dim3(blocks), dim3(320), 0, 0, src, n_rows, pay, sink); })));
            printf("     own wave: non-temporal %.0f, 196 B %.0f, 8 tiles batched %.0f, dense per wave %.0f, 2-MB window (L2-resident) %.0f GB/s\n",
                   gb(time_kernel([&] { hipLaunchKernelGGL(stream_payload_kernel<3>, dim3(blocks), dim3(320), 0, 0, src, n_rows, pay, sink); })),
                   gb(time_kernel([&] { hipLaunchKernelGGL(stream_payload_kernel<4>, dim3(blocks), dim3(320), 0, 0, src, n_rows, pay, sink); })),
                   gb(time_kernel([&] { hipLaunchKernelGGL(stream_payload_kernel<5>, dim3(blocks), dim3(320), 0, 0, src, n_rows, pay, sink); })),
                   gb(time_kernel([&] { hipLaunchKernelGGL(stream_payload_kernel<6>, dim3(blocks), dim3(320), 0, 0, src, n_rows, pay, sink); })),
                   gb(time_kernel([&] { hipLaunchKernelGGL(stream_payload_kernel<7>, dim3(blocks), dim3(320), 0, 0, src, n_rows, pay, sink); })));
            hipFree(pay);
        }
        printf("   linear 1 KiB loads:   rows=1 %.0f rows=2 %.0f rows=4 %.0f GB/s\n",
               bytes / (run<1, true>(src, n_rows, sink, blocks) * 1e-3) / 1e9,
               bytes / (run<2, true>(src, n_rows, sink, blocks) * 1e-3) / 1e9,
               bytes / (run<4, true>(src, n_rows, sink, blocks) * 1e-3) / 1e9);
    }
    return 0;
}
