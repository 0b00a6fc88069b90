// scan.hip — device-wide exclusive prefix sums used by the ordered compactions.
//
// Every append on the path is a prefix sum (canonical sequential order, never atomics): a single-block scan for short
// inputs and a two-level form for long ones — chunk-local prefixes + chunk sums per block, then a scan of the (few)
// chunk sums which most consumers redo themselves in LDS instead of paying a launch for it.
#include "kernels.h"
#include "scan.h"

namespace orbit {

namespace {

// Single-block exclusive scan with carry over chunks of 1024.
__global__ __launch_bounds__(1024) void scan_exclusive_kernel(const uint32_t *in, uint32_t *out, uint32_t n,
                                                              const uint32_t *n_src, uint32_t div, uint32_t *total) {
    __shared__ uint32_t smem[1024 / 64 + 1];
    if (n_src) n = min((*n_src + div - 1u) / div, n);
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? in[i] : 0u;
        uint32_t chunk_total;
        const uint32_t ex = block_exclusive_scan<16>(v, smem, &chunk_total);
        if (i < n) out[i] = carry + ex;
        carry += chunk_total;
    }
    if (threadIdx.x == 0) *total = carry;
}

// Two-level scan for long inputs: each block scans a chunk of kScanChunk words
// (exclusive, chunk-local) and emits the chunk sum; a second single-block launch
// scans the chunk sums.  Consumers add chunk_base[i / kScanChunk] themselves.
__device__ __forceinline__ void scan_chunks_body(const uint32_t *in, uint32_t *out, uint32_t n, const uint32_t *n_src,
                                                 uint32_t div, uint32_t *chunk_sums, uint32_t *clear,
                                                 uint32_t clear_words) {
    __shared__ uint32_t smem[1024 / 64 + 1];
    if (blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < clear_words; i += 1024) clear[i] = 0u;
    if (n_src) n = min((*n_src + div - 1u) / div, n);
    const uint32_t base = blockIdx.x * kScanChunk;
    if (base >= n) { // block-uniform
        if (threadIdx.x == 0) chunk_sums[blockIdx.x] = 0u;
        return;
    }
    uint32_t carry = 0;
#pragma unroll
    for (uint32_t k = 0; k < kScanChunk / 1024; k++) {
        const uint32_t i = base + k * 1024 + threadIdx.x;
        const uint32_t v = i < n ? in[i] : 0u;
        uint32_t sub_total;
        const uint32_t ex = block_exclusive_scan<16>(v, smem, &sub_total);
        if (i < n) out[i] = carry + ex;
        carry += sub_total;
    }
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = carry;
}

__global__ __launch_bounds__(1024) void scan_chunks_kernel(const uint32_t *in, uint32_t *out, uint32_t n,
                                                           const uint32_t *n_src, uint32_t div, uint32_t *chunk_sums,
                                                           uint32_t *clear, uint32_t clear_words) {
    scan_chunks_body(in, out, n, n_src, div, chunk_sums, clear, clear_words);
}

__global__ __launch_bounds__(1024) void scan_chunks_views_kernel(const ScanViews sv, uint32_t div, uint32_t clear_words) {
    const ScanViews::Item &v = sv.v[blockIdx.y];
    scan_chunks_body(v.in, v.out_local, v.n_max, v.n_src, div, v.chunk_sums, v.clear, clear_words);
}

} // namespace

hipError_t launch_scan_chunks_views(const ScanViews &sv, uint32_t count, uint32_t div, uint32_t n_max_all,
                                    uint32_t clear_words, hipStream_t s) {
    uint32_t chunks = (n_max_all + kScanChunk - 1) / kScanChunk;
    if (chunks == 0) chunks = 1;
    hipLaunchKernelGGL(scan_chunks_views_kernel, dim3(chunks, count), dim3(1024), 0, s, sv, div, clear_words);
    return hipGetLastError();
}

namespace {
} // namespace

hipError_t launch_scan_chunks(const uint32_t *in, uint32_t *out_local, const uint32_t *n_src, uint32_t div,
                              uint32_t n_max, uint32_t *chunk_sums, hipStream_t s, uint32_t *clear,
                              uint32_t clear_words) {
    uint32_t chunks = (n_max + kScanChunk - 1) / kScanChunk;
    if (chunks == 0) {
        if (!clear_words) return hipSuccess;
        chunks = 1; // the clear must happen: block 0 finds its chunk empty
    }
    hipLaunchKernelGGL(scan_chunks_kernel, dim3(chunks), dim3(1024), 0, s, in, out_local, n_max, n_src, div, chunk_sums,
                       clear, clear_words);
    return hipGetLastError();
}

hipError_t launch_scan_two_level(const uint32_t *in, uint32_t *out_local, const uint32_t *n_src, uint32_t div,
                                 uint32_t n_max, uint32_t *chunk_sums, uint32_t *chunk_base, uint32_t *total,
                                 hipStream_t s) {
    const uint32_t chunks = (n_max + kScanChunk - 1) / kScanChunk;
    if (chunks == 0) return hipSuccess;
    const hipError_t e = launch_scan_chunks(in, out_local, n_src, div, n_max, chunk_sums, s);
    if (e != hipSuccess) return e;
    return launch_scan_exclusive(chunk_sums, chunk_base, chunks, total, s);
}

hipError_t launch_scan_exclusive(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *total, hipStream_t s) {
    hipLaunchKernelGGL(scan_exclusive_kernel, dim3(1), dim3(1024), 0, s, in, out, n, (const uint32_t *)nullptr, 1u,
                       total);
    return hipGetLastError();
}

} // namespace orbit
