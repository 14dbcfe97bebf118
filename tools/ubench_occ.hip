// How many 256-thread workgroups fit a CU at a given VGPR count / LDS size? (calibration only)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V> __global__ __launch_bounds__(256) void spin(long long cycles, int *out)
{
  extern __shared__ double l[];
  if (V == 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
  if (V == 160) asm volatile("v_mov_b32 v159, 0" ::: "v159");
  if (V == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (V == 176) asm volatile("v_mov_b32 v175, 0" ::: "v175");
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (threadIdx.x == 0 && out) l[0] = 1.0;
}
template <int V> void run(int lds)
{
  hipFuncSetAttribute(reinterpret_cast<const void *>(&spin<V>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  int nb = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&spin<V>), 256, lds);
  for (int per_cu : {1, 2, 3, 4}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<V>, dim3(256 * per_cu), dim3(256), lds, 0, 1000000LL, nullptr);
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin<V>, dim3(256 * per_cu), dim3(256), lds, 0, 1000000LL, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("vgpr %3d lds %6d: %d workgroups per CU worth of grid -> %.2f ms   (runtime says %d blocks/CU)\n", V, lds, per_cu, ms, nb);
  }
}
int main()
{
  run<128>(35632); run<160>(35632); run<168>(35632); run<176>(35632); run<168>(53632); run<168>(69632);
  return 0;
}
