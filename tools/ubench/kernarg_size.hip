#include <hip/hip_runtime.h>
struct P { float m[200]; int *out; int pad[20]; };
struct All { P v[8]; };
__global__ void k(const All a) { const P &p = a.v[blockIdx.y]; p.out[threadIdx.x] = (int)p.m[threadIdx.x % 200] + p.pad[3]; }
int main() { All a{}; int *d; hipMalloc(&d, 4096); for (int i = 0; i < 8; i++) { a.v[i].out = d; a.v[i].m[5] = i + 1; a.v[i].pad[3] = 7; }
  hipLaunchKernelGGL(k, dim3(1, 8), dim3(64), 0, 0, a); hipError_t e = hipDeviceSynchronize(); printf("size %zu err %d\n", sizeof(All), (int)e); return 0; }
