// Micro-benchmarks used to calibrate the kernel cost model (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o gpurun_out/ubench && gpurun_out/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int CH> __global__ void fma_chain(double *out, double a, double b, int iters)
{
  double acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = threadIdx.x + c;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = fma(acc[c], a, b);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c];
  if (s == 123.456) out[0] = s;
}

// ds_read_b128 with a per-lane element stride (in 16-byte elements, fixed-point /1024)
__global__ void lds_read(double *out, int stride1024, int iters)
{
  extern __shared__ double2 l2[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) l2[i] = make_double2(i, -i);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int base = ((lane * stride1024) >> 10) + (threadIdx.x >> 6) * 8;
  double sx = 0, sy = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 25; ++u) {
      double2 v = l2[(base + u + i) & 4095];
      sx += v.x;
      sy += v.y;
    }
  }
  if (sx == 123.456) out[0] = sx + sy;
}

// ds_read_b64 (forced single reads through inline asm), planar layout, two channels = two reads per tap
__global__ void lds_read_b64(double *out, int stride1024, int iters)
{
  extern __shared__ double2 l2[];
  double *l1 = reinterpret_cast<double *>(l2);
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) l1[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int base = ((lane * stride1024) >> 10) + (threadIdx.x >> 6) * 8;
  double sx = 0, sy = 0;
  for (int i = 0; i < iters; ++i) {
    double va[25], vb[25];
    const unsigned a0 = (unsigned)(((base + i) & 2047) * 8), b0 = a0 + 4096 * 8;
#pragma unroll
    for (int u = 0; u < 25; ++u) {
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(va[u]) : "v"(a0), "n"(0 + 8 * u));
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(vb[u]) : "v"(b0), "n"(0 + 8 * u));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 25; ++u) { sx += va[u]; sy += vb[u]; }
  }
  if (sx == 123.456) out[0] = sx + sy;
}

template <class F> float timeit(F f, int reps = 5)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main()
{
  double *d;
  CK(hipMalloc(&d, 1024));
  const int cus = 256, iters = 2000;
  printf("fp64 FMA: chains x waves/SIMD -> cycles per wave-FMA per SIMD (assuming 2.4 GHz) and TFLOP/s\n");
  for (int wps : {1, 2, 4}) {
    const int threads = 256, blocks = cus * wps; // 256 threads = 4 waves = 1 wave/SIMD per block
    auto report = [&](int ch, float ms) {
      double fmas = double(blocks) * threads * iters * 16.0 * ch;
      double tf = 2 * fmas / (ms * 1e-3) / 1e12;
      double wave_instr_per_simd = double(wps) * iters * 16.0 * ch;
      printf("  waves/SIMD %d chains %d : %.3f ms  %.1f TF  %.2f cyc/instr/SIMD@2.4GHz\n", wps, ch, ms, tf,
             ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
    };
    report(1, timeit([&] { hipLaunchKernelGGL(fma_chain<1>, dim3(blocks), dim3(threads), 0, 0, d, 1.0000001, 1e-9, iters); }));
    report(2, timeit([&] { hipLaunchKernelGGL(fma_chain<2>, dim3(blocks), dim3(threads), 0, 0, d, 1.0000001, 1e-9, iters); }));
    report(4, timeit([&] { hipLaunchKernelGGL(fma_chain<4>, dim3(blocks), dim3(threads), 0, 0, d, 1.0000001, 1e-9, iters); }));
    report(8, timeit([&] { hipLaunchKernelGGL(fma_chain<8>, dim3(blocks), dim3(threads), 0, 0, d, 1.0000001, 1e-9, iters); }));
  }
  printf("ds_read_b128, 256 threads/block, 1 or 2 blocks per CU: cycles per wave-instruction per CU @2.4GHz\n");
  for (int bpc : {1, 2})
    for (int stride : {1024, 1882, 2048, 3763, 147 * 1024, 320 * 1024}) {
      const int blocks = cus * bpc, li = 400;
      float ms = timeit([&] { hipLaunchKernelGGL(lds_read, dim3(blocks), dim3(256), 65536, 0, d, stride, li); });
      double instr_per_cu = double(bpc) * 4 * li * 25;
      printf("  blocks/CU %d lane stride %.3f elems: %.3f ms  %.2f cyc/wave-instr/CU\n", bpc, stride / 1024.0, ms,
             ms * 1e-3 * 2.4e9 / instr_per_cu);
    }
  printf("ds_read_b64 x2 per tap (planar), cycles per TAP (2 wave-instr) per CU @2.4GHz\n");
  for (int bpc : {1, 2})
    for (int stride : {1024, 1882, 3763, 147 * 1024}) {
      const int blocks = cus * bpc, li = 400;
      float ms = timeit([&] { hipLaunchKernelGGL(lds_read_b64, dim3(blocks), dim3(256), 65536, 0, d, stride, li); });
      double taps_per_cu = double(bpc) * 4 * li * 25;
      printf("  blocks/CU %d lane stride %.3f elems: %.3f ms  %.2f cyc/tap/CU\n", bpc, stride / 1024.0, ms,
             ms * 1e-3 * 2.4e9 / taps_per_cu);
    }
  return 0;
}
