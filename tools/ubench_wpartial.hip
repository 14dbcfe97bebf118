// Cost of writing interleaved float frames in pieces (calibration only): 8-channel frames (32 bytes), every workgroup
// owns PIECE bytes of each frame of its 7152-frame block (PIECE = 8: one channel pair; 16: two pairs; 32: the whole frame);
// the workgroups that share frames sit on one XCD as item_map places them.  MODE 0: consecutive lanes -> consecutive
// frames; MODE 1: consecutive lanes -> every 4th frame (what an unstaged x4 component store does).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PIECE, int MODE> __global__ __launch_bounds__(256) void wr(char *out, int nblocks)
{
  constexpr int NP = 32 / PIECE, V = 7152;
  const int lin = blockIdx.x, xcd = lin & 7, t = lin >> 3, slot = t / NP, within = t - slot * NP, g = slot * 8 + xcd;
  if (g >= nblocks) return;
  char *base = out + (long long)g * V * 32 + within * PIECE;
  for (int i = threadIdx.x; i < 8192; i += 256) {
    const int m = MODE == 0 ? i : 4 * (i & 2047) + (i >> 11);
    if (m >= V) continue;
    char *p = base + (long long)m * 32;
    if (PIECE == 8) *reinterpret_cast<float2 *>(p) = make_float2(1.f, 2.f);
    else if (PIECE == 16) *reinterpret_cast<float4 *>(p) = make_float4(1.f, 2.f, 3.f, 4.f);
    else {
      reinterpret_cast<float4 *>(p)[0] = make_float4(1.f, 2.f, 3.f, 4.f);
      reinterpret_cast<float4 *>(p)[1] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
  }
}
template <int PIECE, int MODE> static void run(char *out, int nblocks)
{
  constexpr int NP = 32 / PIECE;
  const int grid = (nblocks + 7) / 8 * 8 * NP;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((wr<PIECE, MODE>), dim3(grid), dim3(256), 0, 0, out, nblocks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("piece %2d B  mode %d: %.3f ms for %.2f GB (%.2f TB/s)\n", PIECE, MODE, ms, nblocks * 7152.0 * 32 / 1e9, nblocks * 7152.0 * 32 / 1e9 / ms);
  }
}
int main()
{
  const int nblocks = 4690; // x 7152 frames x 32 B = 1.07 GB, the 44.1k->192k chain's output per step
  char *out; hipMalloc(&out, (size_t)nblocks * 7152 * 32);
  run<8, 0>(out, nblocks); run<8, 1>(out, nblocks); run<16, 0>(out, nblocks); run<16, 1>(out, nblocks); run<32, 0>(out, nblocks);
  return 0;
}
