// Overlap-save FFT-FIR stage for the reference's LONG blocks, N = 32768 / 65536 / 131072 points (lsx_set_dft_length,
// rate/effects_i_dsp.c:64-73: filters of 4096 taps and more; dft_stage_fn, rate/dft_filter.h:60-190).  Such a block
// does not fit one workgroup's LDS, so the transform is split N = 16 x M ("four-step" FFT, M = N/16 <= 8192) and runs as
// three launches that hand the data over through an HBM/L2 workspace:
//
//   big_cols_fwd_kernel   thread = column n2: x[n1*Mp + n2], n1 < 16  -> radix-16 butterfly in registers -> twiddle
//                         W_P^(n2 k1) -> W1[k1][n2]                                (P = forward length = 16*Mp)
//   big_rows_kernel       workgroup = row k1 (the bins k1 + 16*k2): forward FFT of Mp points over n2 (fft_regs, in
//                         LDS), spectrum replication for xL in the frequency domain (row-local: bin k1 + 16*k2 of the
//                         N-point spectrum is bin k1 + 16*(k2 mod Mp) of the P-point one), x G, frequency-domain
//                         decimation (row-local as well), inverse FFT of Md points over k2, twiddle W_Nd^(k1 m2)
//                         -> W2[k1][m2]                                              (Nd = inverse length = 16*Md)
//   big_cols_inv_kernel   thread = column m2: W2[k1][m2], k1 < 16 -> radix-16 butterfly -> y[m2 + Md*m1], m1 < 16,
//                         the valid part of which goes to the stage's output fifo.
//
// Blocks are the reference's own blocks, anchored at absolute stream positions like every other stage, so results are
// bit-invariant to push size and within fp64 rounding of the reference's Ooura transform (no "decoupled" shorter GPU
// blocks any more).  Two channels ride as real / imaginary part of one complex transform, as in dft_kernel.
#include "fft_device.hpp"
#include "fifo_device.hpp"
#include "kernels.hpp"

#include <algorithm>
#include <atomic>

namespace rsmp {

// twN[j] = exp(+2 pi i j / N); W_P^t = twN[t * (N / P)]
__device__ __forceinline__ c64 big_tw(const double2 *__restrict__ twN, long long idx)
{
  const double2 w = twN[idx];
  return {w.x, w.y};
}

__global__ __launch_bounds__(256) void big_cols_fwd_kernel(AnyView in, BigDftArgs a)
{
  const int n2 = blockIdx.x * 256 + threadIdx.x;
  const int item = blockIdx.y; // within this launch
  const int Mp = 1 << a.log2mp;
  if (n2 >= Mp) return;
  const int git = a.item0 + item, npairs = a.d.npairs;
  const int bl = git / npairs, pair = git - bl * npairs;
  const long long B = a.d.B0 + bl;
  const PairCh pc = pair_channels(pair, a.d.C, a.d.nchs, a.d.pps_magic);
  const int ca = pc.ca, cb = pc.cb;
  const bool hasb = pc.hasb;
  const ChanRef ia = chan_ref(in, ca), ib = chan_ref(in, hasb ? cb : ca);

  c64 v[16];
  if (a.fdomain_in) { // L = 1, or x2 / x4 in the frequency domain: the block is Mp*16 consecutive stage inputs
    const long long base = B * a.d.q;
    const PairSpan sp = pair_span(in, pair, hasb, base, 16LL * Mp, ca);
    if (sp.kind) span_load<16>(sp, n2, Mp, v); // the column's 16 loads issued together
    else {
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        const long long e = base + (long long)n1 * Mp + n2;
        v[n1].x = fifo_get(ia, e);
        v[n1].y = hasb ? fifo_get(ib, e) : 0.0;
      }
    }
  } else { // time-domain zero stuffing (dft_filter.h:109-115) in absolute coordinates, as in dft_kernel
    const long long U = B * a.d.V;
    const long long j0 = (U - a.d.c0 + a.d.L - 1) / a.d.L;
    const int remL = (int)(j0 * a.d.L + a.d.c0 - U);
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const int d = n1 * Mp + n2 - remL;
      v[n1] = {0.0, 0.0};
      if (d >= 0 && d % a.d.L == 0) {
        const long long e = j0 + d / a.d.L;
        v[n1].x = fifo_get(ia, e);
        v[n1].y = hasb ? fifo_get(ib, e) : 0.0;
      }
    }
  }
  Bfly<16, -1>::run(v); // v[k1] = sum_n1 x[n1] e^{-2 pi i n1 k1 / 16}
  double2 *w1 = a.w1 + ((size_t)item * 16) * Mp + n2;
  const int tstride = 1 << (a.log2n - a.log2mp - 4); // N / P
  w1[0] = make_double2(v[0].x, v[0].y);
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) {
    const c64 t = cmulc(v[k1], big_tw(a.twN, (long long)n2 * k1 * tstride)); // e^{-2 pi i n2 k1 / P}
    w1[(size_t)k1 * Mp] = make_double2(t.x, t.y);
  }
}

// One workgroup = one row k1 of one (block, channel pair).  LOG2M: row length of the N-point spectrum, LOG2MP: forward
// row length (M / L for frequency-domain upsampling), LOG2MD: inverse row length (M >> m for frequency-domain decimation).
template <int LOG2M, int LOG2MP, int LOG2MD>
__global__ __launch_bounds__((1 << LOG2M) / 16) void big_rows_kernel(BigDftArgs a)
{
  constexpr int M = 1 << LOG2M, MP = 1 << LOG2MP, MD = 1 << LOG2MD;
  constexpr int T = M / 16, TF = MP / 16, TD = MD / 16;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double2 *l2 = reinterpret_cast<double2 *>(lds);
  const int tid = threadIdx.x;
  const int k1 = blockIdx.x, item = blockIdx.y;
  const double2 *__restrict__ src = a.w1 + ((size_t)item * 16 + k1) * MP;

  c64 v[16];
  const bool fwd_active = tid < TF;
  if (fwd_active) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const double2 q = src[tid + s * TF];
      v[s] = {q.x, q.y};
    }
  }
  fft_regs<LOG2MP, -1, 0>(v, tid, fwd_active, a.d.tw_fwd, lds);

  if constexpr (LOG2MP < LOG2M) { // bin k1 + 16*k2 of the N-point spectrum = bin k1 + 16*(k2 mod MP) of the P-point one
    if (fwd_active) {
#pragma unroll
      for (int s = 0; s < 16; ++s) l2[tid + s * TF] = make_double2(v[s].x, v[s].y);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const double2 q = l2[(tid + s * T) & (MP - 1)];
      v[s] = {q.x, q.y};
    }
    __syncthreads();
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) { // x G[k1 + 16*k2]
    const double2 g = a.d.G[k1 + 16 * (tid + s * T)];
    v[s] = cmul(v[s], c64{g.x, g.y});
  }
  if constexpr (LOG2MD < LOG2M) {
    // frequency-domain decimation (dft_filter.h:157-188): bins below Nd/2 and the top Nd/2 of the N-point spectrum;
    // row-local: k2d < MD/2 -> k2 = k2d, else k2 = k2d + M - MD; the new Nyquist bin (row 0, k2d = MD/2) is the mean
    // of its two images
#pragma unroll
    for (int s = 0; s < 16; ++s) l2[tid + s * T] = make_double2(v[s].x, v[s].y);
    __syncthreads();
    if (tid < TD) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int k = tid + s * TD;
        const double2 q = l2[k < MD / 2 ? k : k + M - MD];
        v[s] = {q.x, q.y};
      }
      if (tid == 0 && k1 == 0) {
        const double2 lo = l2[MD / 2], hi = l2[M - MD / 2];
        v[8] = {0.5 * (lo.x + hi.x), 0.5 * (lo.y + hi.y)}; // slot 8 of thread 0 is k2d = 8*TD = MD/2
      }
    }
    __syncthreads();
  }
  const bool inv_active = tid < TD;
  fft_regs<LOG2MD, +1, 0>(v, tid, inv_active, a.d.tw_inv, lds);
  if (inv_active) {
    double2 *dst = a.w2 + ((size_t)item * 16 + k1) * MD;
    const int tstride = 1 << (a.log2n - LOG2MD - 4); // N / Nd
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int m2 = tid + s * TD;
      const c64 t = k1 ? cmul(v[s], big_tw(a.twN, (long long)m2 * k1 * tstride)) : v[s]; // e^{+2 pi i k1 m2 / Nd}
      dst[m2] = make_double2(t.x, t.y);
    }
  }
}

__global__ __launch_bounds__(256) void big_cols_inv_kernel(AnyView out, BigDftArgs a)
{
  const int m2 = blockIdx.x * 256 + threadIdx.x;
  const int item = blockIdx.y;
  const int Md = 1 << a.log2md;
  if (m2 >= Md) return;
  const int git = a.item0 + item, npairs = a.d.npairs;
  const int bl = git / npairs, pair = git - bl * npairs;
  const long long B = a.d.B0 + bl;
  const PairCh pc = pair_channels(pair, a.d.C, a.d.nchs, a.d.pps_magic);
  const int ca = pc.ca, cb = pc.cb;
  const bool hasb = pc.hasb;
  const ChanRef oa = chan_ref(out, ca), ob = chan_ref(out, hasb ? cb : ca);

  c64 v[16];
  const double2 *__restrict__ w2 = a.w2 + ((size_t)item * 16) * Md + m2;
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) {
    const double2 q = w2[(size_t)k1 * Md];
    v[k1] = {q.x, q.y};
  }
  Bfly<16, +1>::run(v); // v[m1] = y[m2 + Md*m1]
  if (a.d.M == 1) {
    const long long o0 = B * a.d.Vout;
#pragma unroll
    for (int m1 = 0; m1 < 16; ++m1) {
      const int n = m2 + m1 * Md;
      if (n < a.d.Vout) {
        fifo_put(oa, a.d.out_offset + o0 + n, v[m1].x);
        if (hasb) fifo_put(ob, a.d.out_offset + o0 + n, v[m1].y);
      }
    }
  } else { // time-domain decimation (dft_filter.h:148-154): keep filtered samples Y with Y % M == 0
    const long long Y0 = B * a.d.V;
#pragma unroll
    for (int m1 = 0; m1 < 16; ++m1) {
      const int n = m2 + m1 * Md;
      const long long Y = Y0 + n;
      if (n < a.d.V && Y % a.d.M == 0) {
        const long long o = Y / a.d.M;
        fifo_put(oa, a.d.out_offset + o, v[m1].x);
        if (hasb) fifo_put(ob, a.d.out_offset + o, v[m1].y);
      }
    }
  }
}

template <int LOG2M, int LOG2MP, int LOG2MD> static hipError_t launch_rows_t(const BigDftArgs &a, int nitems, hipStream_t st)
{
  constexpr int M = 1 << LOG2M;
  constexpr size_t lds_bytes = 8 * size_t(std::max(std::max(fft_lds_doubles(LOG2M), fft_lds_doubles(LOG2MP)), fft_lds_doubles(LOG2MD)));
  static DynLdsOnce attr;
  if (hipError_t e = attr.set(reinterpret_cast<const void *>(&big_rows_kernel<LOG2M, LOG2MP, LOG2MD>), int(lds_bytes)); e != hipSuccess) return e;
  hipLaunchKernelGGL((big_rows_kernel<LOG2M, LOG2MP, LOG2MD>), dim3(16, nitems), dim3(M / 16), lds_bytes, st, a);
  return hipGetLastError();
}

bool big_dft_supported(int log2n, int log2p, int log2nd)
{
  if (log2n < 15 || log2n > 17) return false;
  if (log2p == log2n) return log2nd >= log2n - 2 && log2nd <= log2n;
  return log2nd == log2n && log2p >= log2n - 2;
}

#define RSMP_ROWS_CASE(m, p, d) \
  if (lm == m && lp == p && ld == d) return launch_rows_t<m, p, d>(a, nitems, st);
#define RSMP_ROWS_SIZE(m, m1, m2) \
  RSMP_ROWS_CASE(m, m, m) RSMP_ROWS_CASE(m, m1, m) RSMP_ROWS_CASE(m, m2, m) RSMP_ROWS_CASE(m, m, m1) RSMP_ROWS_CASE(m, m, m2)

static hipError_t launch_rows(const BigDftArgs &a, int nitems, hipStream_t st)
{
  const int lm = a.log2n - 4, lp = a.log2mp, ld = a.log2md;
  RSMP_ROWS_SIZE(11, 10, 9)
  RSMP_ROWS_SIZE(12, 11, 10)
  RSMP_ROWS_SIZE(13, 12, 11)
  return hipErrorInvalidValue;
}

// `a.item0` / workspaces are filled in here per chunk: at most `ws_items` (block, pair) items are in flight at once.
hipError_t launch_dft_big(bool src_f32, bool dst_f32, const F32View &sf, const F64View &sd, const F32View &df, const F64View &dd,
                          BigDftArgs a, int ws_items, hipStream_t st)
{
  const AnyView in = make_view(src_f32, sf, sd), out = make_view(dst_f32, df, dd);
  a.d.npairs = pair_count(a.d.C, a.d.nchs);
  a.d.pps_magic = pair_magic(a.d.C, a.d.nchs);
  const int npairs = a.d.npairs;
  const long long total = (long long)a.d.nblocks * npairs;
  const int Mp = 1 << a.log2mp, Md = 1 << a.log2md;
  for (long long i0 = 0; i0 < total; i0 += ws_items) {
    const int n = int(std::min<long long>(ws_items, total - i0));
    a.item0 = int(i0);
    hipLaunchKernelGGL(big_cols_fwd_kernel, dim3((Mp + 255) / 256, n), dim3(256), 0, st, in, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if ((e = launch_rows(a, n, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(big_cols_inv_kernel, dim3((Md + 255) / 256, n), dim3(256), 0, st, out, a);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  return hipSuccess;
}

} // namespace rsmp
