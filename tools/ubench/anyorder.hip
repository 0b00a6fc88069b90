// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  (hip_ext.h says the flag is not supported on
// GFX9 boards.)  Two single-workgroup kernels that each spin for ~20 us: back to back 40 us; overlapped 20.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/anyorder.bin tools/ubench/anyorder.hip
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void spin(unsigned long long ticks, unsigned *out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out) out[0] = 1;
}

int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    unsigned *d;
    hipMalloc(&d, 64);
    const unsigned long long ticks = 2000; // wall_clock64: 100 MHz -> 20 us
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 20; rep++) {
            hipEventRecord(a, s);
            if (mode == 0) {
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, d);
                hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, d);
            } else if (mode == 1) {
                hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, ticks, d);
                hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d);
            } else {
                hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d);
                hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d);
            }
            hipEventRecord(b, s);
            hipStreamSynchronize(s);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("%s\"%s\": %.1f", mode ? ", " : "{", mode == 0 ? "in_order_us" : mode == 1 ? "second_any_order_us" : "both_any_order_us", best * 1e3f);
    }
    printf("}\n");
    return 0;
}
