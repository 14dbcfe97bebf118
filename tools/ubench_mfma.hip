// Does the fp64 matrix pipe run beside the fp64 vector pipe on gfx950?  (calibration only, not product code)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o gpurun_out/ubench_mfma && gpurun_out/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// mode bit0: waves with (wave/4)&1 == 0 run MFMA chains; bit1: the others run VALU chains.
// mode 4: every wave interleaves both.
template <int CH> __global__ void mix(double *out, double a, double b, int iters, int mode)
{
  const int wave = threadIdx.x >> 6;
  const int role = (wave >> 2) & 1;
  double s = 0;
  if (mode == 4) {
    d4 acc[CH];
    double v[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { acc[c] = (d4){0, 0, 0, 0}; v[c] = threadIdx.x + c; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[(c + k) % CH] = fma(v[(c + k) % CH], a, b);
        }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3] + v[c];
  } else if (role == 0) {
    if (!(mode & 1)) return;
    d4 acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = (d4){0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  } else {
    if (!(mode & 2)) return;
    double v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = threadIdx.x + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4 * CH; ++u)
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = fma(v[c], a, b);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) s += v[c];
  }
  if (s == 123.456) out[0] = s;
}

// 4x4x4 (4 blocks) variant alone
template <int CH> __global__ void mfma4(double *out, double a, double b, int iters)
{
  double acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c];
  if (s == 123.456) out[0] = s;
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

// D layout probe: A[m][k] = m*10+k (lane -> m = lane&15, k = lane>>4), B[k][n] = delta(k==0)*1 + ... we use B = e_{k0} rows
__global__ void layout_probe(double *out)
{
  const int lane = threadIdx.x;
  // A[m][k]: value 100*m + k ; B[k][n]: 1 if k == 0 else 0  -> D[m][n] = 100*m  (tells the row map)
  // second product: A[m][k] = 1 if k==0 ; B[k][n] = n -> D[m][n] = n (tells the col map)
  d4 z = {0, 0, 0, 0};
  const double a1 = 100.0 * (lane & 15) + (lane >> 4), b1 = ((lane >> 4) == 0) ? 1.0 : 0.0;
  d4 r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, z, 0, 0, 0);
  const double a2 = ((lane >> 4) == 0) ? 1.0 : 0.0, b2 = (double)(lane & 15) + 0.25 * (lane >> 4);
  d4 r2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, z, 0, 0, 0);
  // k-order probe: terms of very different magnitude so that the rounding order shows
  const int k = lane >> 4;
  const double ak = (k == 0) ? 1.0 : (k == 1) ? 0x1p-53 : (k == 2) ? 0x1p-53 : -1.0;
  d4 r3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ak, 1.0, z, 0, 0, 0);
  for (int j = 0; j < 4; ++j) {
    out[(0 * 4 + j) * 64 + lane] = r1[j];
    out[(1 * 4 + j) * 64 + lane] = r2[j];
    out[(2 * 4 + j) * 64 + lane] = r3[j];
  }
}

int main()
{
  double *out;
  CK(hipMalloc(&out, 1 << 20));
  const double clk = 2.4e9;
  const int iters = 2000;
  const int grid = 256 * 8;
  // MFMA flops per wave-instruction: 16*16*4*2 = 2048
  for (int wpb : {256, 512, 1024}) {  // 4, 8, 16 waves per workgroup; one workgroup per CU at a time is not forced
    for (int mode : {1, 2, 3, 4}) {
      float ms = timeit([&] { hipLaunchKernelGGL(mix<4>, dim3(grid), dim3(wpb), 0, 0, out, 1.0000001, 1e-9, iters, mode); });
      const double waves = (double)grid * (wpb / 64);
      const double mf_waves = mode == 4 ? waves : (mode & 1) ? (wpb >= 512 ? waves / 2 : waves) : 0;
      const double va_waves = mode == 4 ? waves : (mode & 2) ? (wpb >= 512 ? waves / 2 : 0) : 0;
      const double mf_instr = mf_waves * iters * 16.0, va_instr = va_waves * iters * (mode == 4 ? 16.0 * 8 : 16.0 * 8);
      const double tf = (mf_instr * 2048 + va_instr * 128) / (ms * 1e-3) * 1e-12;
      printf("threads %4d mode %d: %.3f ms  mfma %.2f TF  valu %.2f TF  total %.2f TF\n", wpb, mode, ms,
             mf_instr * 2048 / (ms * 1e-3) * 1e-12, va_instr * 128 / (ms * 1e-3) * 1e-12, tf);
    }
  }
  {
    float ms = timeit([&] { hipLaunchKernelGGL(mfma4<8>, dim3(grid), dim3(512), 0, 0, out, 1.0000001, 1e-9, iters); });
    const double instr = (double)grid * 8 * iters * 32.0;
    printf("mfma_f64_4x4x4 x8 chains: %.3f ms  %.2f TF (512 flops per instr)  %.1f cycles per wave-instr per SIMD\n", ms,
           instr * 512 / (ms * 1e-3) * 1e-12, ms * 1e-3 * clk / (instr / (256 * 4)));
  }
  hipLaunchKernelGGL(layout_probe, dim3(1), dim3(64), 0, 0, out);
  CK(hipDeviceSynchronize());
  static double h[3 * 4 * 64];
  CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  printf("row map (D/100) lanes 0,1,16,17,32,48 regs 0..3:\n");
  for (int lane : {0, 1, 16, 17, 32, 48}) {
    printf("  lane %2d:", lane);
    for (int j = 0; j < 4; ++j) printf(" row %g col %g |", h[(0 * 4 + j) * 64 + lane] / 100.0, h[(1 * 4 + j) * 64 + lane]);
    printf("\n");
  }
  printf("k-order probe (1 + 2^-53 + 2^-53 - 1): %a  (sequential fma chain k=0..3 gives 0 ; exact sum gives 0x1p-52)\n", h[(2 * 4 + 0) * 64]);
  return 0;
}
