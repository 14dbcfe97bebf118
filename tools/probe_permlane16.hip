#include <hip/hip_runtime.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *o) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  u2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r.x; o[64 + threadIdx.x] = r.y;
}
int main(){ unsigned *d; hipMalloc(&d, 512); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
 for (int i=0;i<64;i+=8) printf("lane %2d: P=%u Q=%u\n", i, h[i], h[64+i]); return 0; }
