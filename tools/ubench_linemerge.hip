// Do loads of the same 128-byte line by different waves of ONE workgroup merge in L1 (one L2 request), where the same loads
// from different workgroups do not?  The access pattern of the lean kernel's input phase on 32-channel frames (128 bytes per
// frame, a channel pair = 8 of them): every lane reads 8 bytes of its own line.
//   mode 0: one pair per 256-thread workgroup (today): grid = groups * 16
//   mode 1: three pairs per 768-thread workgroup (sub-group k reads pair 3 j + k of the same frames): grid = groups * 6 (16 pairs as 5 x 3 + 1)
//   mode 2: one pair per workgroup, but 64-byte-contiguous reads (what planar input would cost): lower bound
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench_linemerge tools/ubench_linemerge.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(768) void k(const float2 *in, float2 *out, int frames, int mode, int pairs_per_frame)
{
  const int sub = threadIdx.x >> 8, tid = threadIdx.x & 255;
  int group, pair;
  if (mode == 1) {
    group = blockIdx.x / 6;
    pair = (blockIdx.x % 6) * 3 + sub;
    if (pair >= pairs_per_frame) return;
  } else {
    if (sub) return;
    group = blockIdx.x / pairs_per_frame;
    pair = blockIdx.x % pairs_per_frame;
  }
  const float2 *base = in + (size_t)group * frames * pairs_per_frame;
  float2 acc = make_float2(0.f, 0.f);
  float2 v[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int f = tid + 256 * s;
    v[s] = mode == 2 ? base[(size_t)pair * frames + f] : base[(size_t)f * pairs_per_frame + pair];
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) { acc.x += v[s].x; acc.y += v[s].y; }
  // a few microseconds of dependent arithmetic, so that the loads are a phase of the workgroup and not all of it
  for (int i = 0; i < 2000; ++i) acc.x = fmaf(acc.x, 1.0000001f, acc.y);
  if (acc.x == 1.2345f) out[blockIdx.x * 256 + tid] = acc;
}
int main()
{
  const int frames = 4096, ppf = 16, groups = 4096; // 4096 x 512 KB = 2 GB of input
  float2 *in, *out;
  hipMalloc(&in, (size_t)groups * frames * ppf * sizeof(float2));
  hipMalloc(&out, (size_t)groups * 16 * 256 * sizeof(float2));
  hipMemset(in, 0, (size_t)groups * frames * ppf * sizeof(float2));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      const int grid = mode == 1 ? groups * 6 : groups * ppf, block = mode == 1 ? 768 : 256;
      hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, in, out, frames, mode, ppf);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, in, out, frames, mode, ppf);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d: %.3f ms  (%.0f GB/s of useful input)\n", mode, ms, (double)groups * frames * ppf * 8 / ms / 1e6);
    }
  return 0;
}
