// Cost of the polyphase epilogue's store pattern (calibration only): every wave-instruction writes 4 periods x
// 16 residues of stereo float frames; 8 bytes per lane (one frame) vs 16 bytes per lane (two adjacent frames).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W> __global__ void st(float *out, long long frames_per_wg, int iters_per_wave)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *base = out + (long long)blockIdx.x * frames_per_wg * 2;
  // lane -> (hi = lane >> 4, bq = (lane >> 2) & 3, jq = lane & 3): residue 4*bq + hi (+16*g), period jq (+4*cs)
  const int hi = lane >> 4, bq = (lane >> 2) & 3, jq = lane & 3;
  for (int it = wave; it < iters_per_wave * 4; it += 4) {
    const int g = it % 10, cs = it / 10;
    if (W == 8) {
      const long long frame = (long long)(4 * cs + jq) * 160 + 16 * g + 4 * bq + hi;
      *reinterpret_cast<float2 *>(base + 2 * frame) = make_float2(1.f, 2.f);
    } else { // two column steps per instruction: even hi -> frames (r, r+1) of step 2cs', odd hi -> of step 2cs'+1
      const int r = 16 * g + 4 * bq + (hi & 2);
      const long long frame = (long long)(4 * (2 * (cs >> 1) + (hi & 1)) + jq) * 160 + r;
      if (cs & 1) continue;
      *reinterpret_cast<float4 *>(base + 2 * frame) = make_float4(1.f, 2.f, 3.f, 4.f);
    }
  }
}
int main()
{
  const long long frames_per_wg = 3840, wgs = 69632; // 24 periods x 160 residues per workgroup
  float *out; hipMalloc(&out, wgs * frames_per_wg * 8);
  for (int w : {8, 16, 8, 16}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (w == 8) hipLaunchKernelGGL(st<8>, dim3(wgs), dim3(256), 0, 0, out, frames_per_wg, 15);
    else hipLaunchKernelGGL(st<16>, dim3(wgs), dim3(256), 0, 0, out, frames_per_wg, 15);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%2d bytes per lane: %.3f ms for %.2f GB  (%.2f TB/s)\n", w, ms, wgs * frames_per_wg * 8 / 1e9, wgs * frames_per_wg * 8 / 1e9 / ms);
  }
  return 0;
}
