// Lane maps of v_mfma_f64_4x4x4_4b_f64 on gfx950 (calibration only, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double *out)
{
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la) {
    const double a = lane == la ? 1.0 : 0.0, b = lane + 1.0;
    out[la * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  }
}
int main()
{
  double *d; hipMalloc(&d, 64 * 64 * 8);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  static double h[64 * 64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int l = 0; l < 64; ++l) if (h[la * 64 + l] != 0.0) printf(" D%d<-B%d", l, (int)h[la * 64 + l] - 1);
    printf("\n");
  }
  return 0;
}
