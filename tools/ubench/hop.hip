// hop.hip — what one link of a dependent chain costs on this box: the denominator of the per-config "floor" lines of
// tools/bench_configs.py (configs 2-4 are latency chains, bytes / time against 8 TB/s says nothing).
//   launch_us       an empty kernel inside its own HIP event pair (what every timed call carries), median
//   boundary_us     what one more (empty) launch adds on a stream: (N launches in one event pair) / N
//   load_hop_us     one dependent global load that misses every cache: a pointer chase over 256 MB, one lane
//   atomic_hop_us   one returning device-scope atomic on a line of its own (a ticket)
//   handoff_hop_us  store (write-through, drained) in one workgroup -> seen by the polling sc1 load of another: half the
//                   round trip of a ping-pong between two workgroups (a hand-over inside a launch, handoff.h)
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/hop.bin tools/ubench/hop.hip && tools/ubench/hop.bin
// Prints one JSON object.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            return 1;                                                             \
        }                                                                         \
    } while (0)

__global__ void empty_kernel() {}

__global__ void chase_kernel(const uint32_t *next, uint32_t start, uint32_t hops, uint32_t *out) {
    uint32_t i = start;
    for (uint32_t k = 0; k < hops; k++) i = next[(size_t)i * 32u]; // one 128-B line per element
    *out = i;
}

__global__ void atomic_kernel(uint32_t *ctr, uint32_t hops, uint32_t *out) {
    uint32_t v = 0;
    for (uint32_t k = 0; k < hops; k++) v += __hip_atomic_fetch_add(ctr + (v >> 31), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (the address depends on the last result)
    *out = v;
}

// two workgroups on (most likely) two XCDs: A stores k to flag[0], B waits for it and stores k to flag[32], A waits ...
__global__ void pingpong_kernel(uint32_t *flags, uint32_t rounds, uint32_t *timeout) {
    if (threadIdx.x != 0) return;
    const __amdgpu_buffer_rsrc_t R = __builtin_amdgcn_make_buffer_rsrc(flags, 0, 64 * 4 * 2, 0x00020000);
    const uint32_t me = blockIdx.x == 0 ? 0u : 1u;
    for (uint32_t k = 1; k <= rounds; k++) {
        if (me == 0u) {
            __builtin_amdgcn_raw_buffer_store_b32(k, R, 0, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        uint32_t spins = 0;
        while (__builtin_amdgcn_raw_buffer_load_b32(R, me == 0u ? 128 : 0, 0, 16) < k) {
            asm volatile("s_sleep 1" ::: "memory");
            if (++spins > (1u << 22)) {
                *timeout = 1u;
                return;
            }
        }
        if (me == 1u) {
            __builtin_amdgcn_raw_buffer_store_b32(k, R, 128, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
}

static float median(std::vector<float> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main() {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    // keep the device busy for a moment first (clocks)
    for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(256), 0, s);
    CHECK(hipStreamSynchronize(s));
    std::vector<float> t;
    float ms;
    for (int i = 0; i < 101; i++) {
        CHECK(hipEventRecord(a, s));
        hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
        CHECK(hipEventRecord(b, s));
        CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms * 1e3f);
    }
    const float launch_us = median(t);
    t.clear();
    for (int i = 0; i < 21; i++) {
        CHECK(hipEventRecord(a, s));
        for (int k = 0; k < 64; k++) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
        CHECK(hipEventRecord(b, s));
        CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms * 1e3f / 64.0f);
    }
    const float boundary_us = median(t);

    // pointer chase: a random cycle over 2 M lines (256 MB)
    const uint32_t n = 1u << 21, hops = 256;
    std::vector<uint32_t> perm(n), next_h((size_t)n * 32u, 0u);
    std::iota(perm.begin(), perm.end(), 0u);
    uint64_t rng = 88172645463325252ull;
    for (uint32_t i = n - 1; i > 0; i--) {
        rng ^= rng << 13, rng ^= rng >> 7, rng ^= rng << 17;
        std::swap(perm[i], perm[rng % (i + 1)]);
    }
    for (uint32_t i = 0; i < n; i++) next_h[(size_t)perm[i] * 32u] = perm[(i + 1) % n];
    uint32_t *next_d, *out_d, *ctr_d, *flags_d, *to_d;
    CHECK(hipMalloc(&next_d, (size_t)n * 128u));
    CHECK(hipMalloc(&out_d, 256));
    CHECK(hipMalloc(&ctr_d, 4096));
    CHECK(hipMalloc(&flags_d, 4096));
    CHECK(hipMalloc(&to_d, 256));
    CHECK(hipMemcpy(next_d, next_h.data(), (size_t)n * 128u, hipMemcpyHostToDevice));
    CHECK(hipMemset(ctr_d, 0, 4096));
    CHECK(hipMemset(to_d, 0, 256));
    t.clear();
    for (int i = 0; i < 21; i++) {
        CHECK(hipEventRecord(a, s));
        hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(1), 0, s, next_d, perm[(i * 7919u) % n], hops, out_d);
        CHECK(hipEventRecord(b, s));
        CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms, a, b));
        t.push_back((ms * 1e3f - launch_us) / hops);
    }
    const float load_hop_us = median(t);
    t.clear();
    for (int i = 0; i < 21; i++) {
        CHECK(hipEventRecord(a, s));
        hipLaunchKernelGGL(atomic_kernel, dim3(1), dim3(1), 0, s, ctr_d, hops, out_d);
        CHECK(hipEventRecord(b, s));
        CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms, a, b));
        t.push_back((ms * 1e3f - launch_us) / hops);
    }
    const float atomic_hop_us = median(t);
    t.clear();
    const uint32_t rounds = 200;
    for (int i = 0; i < 11; i++) {
        CHECK(hipMemsetAsync(flags_d, 0, 4096, s));
        CHECK(hipEventRecord(a, s));
        hipLaunchKernelGGL(pingpong_kernel, dim3(2), dim3(64), 0, s, flags_d, rounds, to_d);
        CHECK(hipEventRecord(b, s));
        CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms, a, b));
        t.push_back((ms * 1e3f - launch_us) / (2.0f * rounds));
    }
    uint32_t timed_out = 0;
    CHECK(hipMemcpy(&timed_out, to_d, 4, hipMemcpyDeviceToHost));
    const float handoff_hop_us = median(t);
    printf("{\"launch_us\": %.3f, \"boundary_us\": %.3f, \"load_hop_us\": %.3f, \"atomic_hop_us\": %.3f, \"handoff_hop_us\": %.3f, "
           "\"handoff_timed_out\": %u}\n",
           launch_us, boundary_us, load_hop_us, atomic_hop_us, handoff_hop_us, timed_out);
    return 0;
}
