// Does a raw buffer dword load at a byte offset that is not a multiple of 4 return the bytes AT that offset on this
// device (gfx950, the alignment mode ROCm sets for compute)?  Prints "unaligned buffer loads: ok" or what came back.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void k(const uint8_t *src, uint32_t *out) {
    const __amdgpu_buffer_rsrc_t R = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, 4096, 0x00020000);
    const uint32_t off = threadIdx.x * 3u + 1u; // 1, 4, 7, 10, ...
    out[threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(R, off, 0, 0);
    out[64 + threadIdx.x] = *reinterpret_cast<const uint32_t *>(src + off); // the same through a global load
}
int main() {
    uint8_t h[4096];
    for (int i = 0; i < 4096; i++) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *d; uint32_t *o, r[128];
    hipMalloc(&d, 4096); hipMalloc(&o, 512);
    hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    if (hipMemcpy(r, o, 512, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 1; }
    int bad_buf = 0, bad_glob = 0;
    for (int t = 0; t < 64; t++) {
        const int off = t * 3 + 1;
        const uint32_t want = h[off] | h[off + 1] << 8 | h[off + 2] << 16 | (uint32_t)h[off + 3] << 24;
        bad_buf += r[t] != want; bad_glob += r[64 + t] != want;
    }
    printf("unaligned buffer loads: %s (%d of 64 wrong); unaligned global loads: %s (%d wrong)\n", bad_buf ? "NOT ok" : "ok", bad_buf,
           bad_glob ? "NOT ok" : "ok", bad_glob);
    return 0;
}
