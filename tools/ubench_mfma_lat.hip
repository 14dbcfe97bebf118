// Dependent-issue latency of v_mfma_f64_4x4x4_4b_f64: one wave per SIMD, CH independent accumulation chains.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH> __global__ void chains(double *out, double a, double b, int iters)
{
  double acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c];
  if (s == 123.456) out[0] = s;
}
template <int CH> void run(double *out, int threads)
{
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(chains<CH>, dim3(256), dim3(threads), 0, 0, out, 1.0000001, 1e-9, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(chains<CH>, dim3(256), dim3(threads), 0, 0, out, 1.0000001, 1e-9, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = double(iters) * 8 * CH * (threads / 256);
  printf("chains %d, waves/SIMD %d: %.1f cycles per MFMA per SIMD (2.4 GHz)\n", CH, threads / 256, ms * 1e-3 * 2.4e9 / per_simd);
}
int main()
{
  double *out; hipMalloc(&out, 64);
  run<1>(out, 256); run<2>(out, 256); run<3>(out, 256); run<4>(out, 256);
  run<1>(out, 512); run<2>(out, 512); run<1>(out, 768); run<2>(out, 768);
  return 0;
}
