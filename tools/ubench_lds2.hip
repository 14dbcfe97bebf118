// LDS read cost of candidate lane->address maps for the matrix-pipe polyphase loop (calibration only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

template <int WIDTH> __global__ void rd(const int *tab, double *out, int iters)
{
  extern __shared__ double l[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) l[i] = i;
  __syncthreads();
  const unsigned a0 = (unsigned)tab[threadIdx.x & 63];
  double s0 = 0, s1 = 0;
  for (int i = 0; i < iters; ++i) {
    if (WIDTH == 16) {
      double2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[u]) : "v"(a0), "n"(0 + 64 * u));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) { s0 += v[u].x; s1 += v[u].y; }
    } else {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[u]) : "v"(a0), "n"(0 + 64 * u));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) s0 += v[u];
    }
  }
  if (s0 == 123.456) out[0] = s0 + s1;
}

int main()
{
  int *dt; double *out;
  hipMalloc(&dt, 256); hipMalloc(&out, 64);
  auto run = [&](const char *name, int width, const std::vector<int> &t) {
    hipMemcpy(dt, t.data(), 256, hipMemcpyHostToDevice);
    const int iters = 2000, grid = 256, waves = 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (width == 16) hipLaunchKernelGGL(rd<16>, dim3(grid), dim3(64 * waves), 65536, 0, dt, out, iters);
      else hipLaunchKernelGGL(rd<8>, dim3(grid), dim3(64 * waves), 65536, 0, dt, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // one workgroup per CU (64 KB LDS x 2 fit, grid = 256 -> 1 per CU): cycles per wave-instruction on the CU's LDS
    printf("%-46s b%-3d %.2f cycles per wave-instruction\n", name, width * 8, ms * 1e-3 * 2.4e9 / (iters * 8.0 * waves));
  };
  const int Q[4] = {0, 3, 7, 11}, Q2[4] = {0, 4, 7, 11};
  std::vector<int> t(64);
  for (int l = 0; l < 64; ++l) t[l] = 16 * l;
  run("linear 16B stride (reference)", 16, t);
  for (int l = 0; l < 64; ++l) t[l] = 8 * l;
  run("linear 8B stride (reference)", 8, t);
  for (int l = 0; l < 64; ++l) { int k = l >> 4, b = (l >> 2) & 3, qq = (l >> 1) & 1, ch = l & 1; t[l] = 16 * (k + Q[b] + 147 * qq) + 8 * ch; }
  run("current: b=block j=(2 periods x ch)", 8, t);
  for (int l = 0; l < 64; ++l) { int k = l >> 4, b = (l >> 2) & 3, j = l & 3; t[l] = 16 * (k + Q[b] + 147 * j); }
  run("P1: b=block j=period (Q 0,3,7,11)", 16, t);
  for (int l = 0; l < 64; ++l) { int k = l >> 4, b = (l >> 2) & 3, j = l & 3; t[l] = 16 * (k + Q2[b] + 147 * j); }
  run("P1: b=block j=period (Q 0,4,7,11)", 16, t);
  for (int l = 0; l < 64; ++l) { int k = l >> 4, b = (l >> 2) & 3, j = l & 3; t[l] = 16 * (k + Q[j] + 147 * b); }
  run("P2: b=period j=block", 16, t);
  for (int sp = 1; sp <= 12; ++sp) { // P3: blocks 4*sp residues apart
    for (int l = 0; l < 64; ++l) { int k = l >> 4, b = (l >> 2) & 3, j = l & 3; t[l] = 16 * (k + (4 * sp * b * 147) / 160 + 147 * j); }
    char nm[64]; snprintf(nm, 64, "P3: b=block spaced %d residues, j=period", 4 * sp);
    run(nm, 16, t);
  }
  for (int pstep : {80 * 0 + 147, 320, 160}) for (int L : {160, 147, 80}) {
    if ((pstep == 320) != (L == 147)) continue;
    for (int l = 0; l < 64; ++l) { int k = l >> 4, b = (l >> 2) & 3, j = l & 3; t[l] = 16 * (k + (4 * b * pstep) / L + (pstep % 256) * j); }
    char nm[64]; snprintf(nm, 64, "P1 for step %d / L %d", pstep, L);
    run(nm, 16, t);
  }
  return 0;
}
