// What a second active queue costs a chain of dependent launches on the first: K small kernels back to back on stream s
// (a) alone, (b) with an unrelated kernel enqueued on a second stream at the same time, (c) with that kernel forked off s
// by an event (not joined), (d) forked and joined.  K = 2, 6, 12: a cost per fork is constant in K, a cost per launch
// boundary grows with it.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/forktax.bin tools/ubench/forktax.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void small(unsigned *p, unsigned n) {
    // a few dependent global round trips, like the launches of a latency-bound chain
    unsigned v = p[threadIdx.x];
    for (unsigned i = 0; i < n; i++) v = p[(v + i) & 1023u];
    if (v == 0xdeadbeefu) p[0] = v;
}
__global__ void side_work(unsigned long long ticks, unsigned *out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out[0] == 0xdeadbeefu) out[1] = 1;
}

int main() {
    hipStream_t s, t;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&t, hipStreamNonBlocking);
    hipEvent_t a, b, f, j;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventCreateWithFlags(&f, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&j, hipEventDisableTiming);
    unsigned *d, *e;
    (void)hipMalloc(&d, 4096 * 4);
    (void)hipMalloc(&e, 4096 * 4);
    (void)hipMemset(d, 0, 4096 * 4);
    (void)hipMemset(e, 0, 4096 * 4);
    (void)hipDeviceSynchronize();
    printf("{");
    for (int K : {2, 6, 12}) {
        for (int mode = 0; mode < 4; mode++) {
            float best = 1e9f, sum = 0;
            for (int rep = 0; rep < 30; rep++) {
                hipLaunchKernelGGL(small, dim3(64), dim3(256), 0, s, d, 8u); // something in front
                (void)hipEventRecord(a, s);
                if (mode >= 2) {
                    (void)hipEventRecord(f, s);
                    (void)hipStreamWaitEvent(t, f, 0);
                }
                if (mode >= 1) hipLaunchKernelGGL(side_work, dim3(8), dim3(256), 0, t, 1000ull, e); // 10 us, eight workgroups
                if (mode == 3) (void)hipEventRecord(j, t);
                for (int k = 0; k < K; k++) hipLaunchKernelGGL(small, dim3(64), dim3(256), 0, s, d, 8u);
                if (mode == 3) (void)hipStreamWaitEvent(s, j, 0);
                (void)hipEventRecord(b, s);
                (void)hipStreamSynchronize(s);
                (void)hipStreamSynchronize(t);
                float ms;
                (void)hipEventElapsedTime(&ms, a, b);
                if (rep >= 5) {
                    sum += ms;
                    if (ms < best) best = ms;
                }
            }
            printf("%s\"K%d_%s_us\": [%.1f, %.1f]", (K == 2 && mode == 0) ? "" : ", ", K,
                   mode == 0 ? "alone" : mode == 1 ? "unrelated_side_kernel" : mode == 2 ? "forked" : "forked_joined", best * 1e3f,
                   sum / 25 * 1e3f);
        }
    }
    printf("}\n");
    return 0;
}
