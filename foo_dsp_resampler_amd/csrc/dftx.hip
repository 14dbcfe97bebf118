// x4 (x8) upsampling DFT stage on 2048-point transforms: dft_stage_fn's power-of-two-L branch
// (rate/dft_filter.h:86-104,118-156) for blocks of N = L * 2048 points, computed as its L polyphase components.
//
// The reference transforms the block's P = N/L inputs, repeats the P-point spectrum L times, multiplies by the filter's
// N-point spectrum G and inverse-transforms N points.  With the output index written m = L n + r and the bin index
// k = k1 + P k2 the inverse transform splits exactly (first step of a decimation in time):
//
//   y[L n + r] = sum_{k1 < P} e^{+2 pi i k1 n / P} X[k1] G_r[k1],   G_r[k1] = e^{+2 pi i k1 r / N} sum_{k2 < L} G[k1 + P k2] e^{+2 pi i k2 r / L}
//
// and G_r is nothing but DFT_P of the filter's r-th polyphase component (L * h_placed[L j + r]) / P, which the host
// builds directly from the taps (Engine: d_Gr_).  So a block costs one forward and L inverse transforms of P = 2048
// points instead of one of P and one of N points: 15 % fewer flops, and -- the point -- every transform has the shape
// the headline kernel's transforms have (256 threads, 4 waves, 3 workgroups per CU) instead of dft_kernel<13,11,13>'s
// 512 threads at 128 VGPRs.
//
// Work split inside the workgroup: every transform runs 8 points per thread on all 256 threads (fft8_regs), so a thread
// keeps its 8 spectrum values X[tid + 256 j] in registers for all L components and nothing has to be redistributed;
// ~100 VGPRs and 37 KB of LDS let four workgroups share a CU.  (Tried first: the two halves of the workgroup running
// two components side by side in the 16-points-per-thread layout -- fewer exchanges, but X[16] + v[16] + the half-round
// exchange's 8 spare values spill at 168 VGPRs.)  The results of two components (adjacent output frames) are staged
// through LDS so that a lane pair stores them into one 64-byte write request.
#include "fft_device.hpp"
#include "fifo_device.hpp"
#include "kernels.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>

// RSMP_DFTX_SKIP (knobs.hpp; -DRSMP_EXPERIMENTS builds only, WRONG results): 1 = no stores of the first component pair, 2 = of the last

namespace rsmp {

namespace {
constexpr int kXP = 11, kP = 1 << kXP; // points per transform
// LDS: [exchange area of fft8 (aliased by the output stage: 2P float2)] [twiddles w^1, w^2, w^4 of the two radix-8 passes
// with twiddles and w^1 of the final radix-4 pass: (3*8 + 3*64 + 512) entries]
constexpr int kXExch = std::max(fft8_lds_doubles(kXP), 2 * kP);
constexpr int kXTw = 3 * 8 + 3 * 64 + 512;
constexpr size_t kXLdsBytes = 8 * size_t(kXExch) + 16 * size_t(kXTw);

// fft8_pass (fft_device.hpp) for the 2048-point plan 8 x 8 x 8 x 4 with the twiddles read from LDS instead of the global
// table: nothing inside a transform touches vector memory, so no transform ever waits on the workgroup's own stores.
template <int R, int NS, int DIR, bool LAST>
__device__ __forceinline__ void fft8x_pass(c64 (&u)[8], int tid, const double2 *twl, double *lds)
{
  constexpr int T8 = kP / 8, NB = 8 / R;
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    c64 b[R];
#pragma unroll
    for (int r = 0; r < R; ++r) b[r] = u[t + NB * r];
    if (NS > 1) {
      const int k = (tid + t * T8) & (NS - 1);
      c64 w[8];
      const double2 q1 = twl[k];
      w[1] = {q1.x, q1.y};
      if (R == 8) {
        const double2 q2 = twl[NS + k], q4 = twl[2 * NS + k];
        w[2] = {q2.x, q2.y};
        w[4] = {q4.x, q4.y};
      } else
        w[2] = cmul(w[1], w[1]);
      w[3] = cmul(w[1], w[2]);
      if (R == 8) {
        w[5] = cmul(w[4], w[1]);
        w[6] = cmul(w[4], w[2]);
        w[7] = cmul(w[4], w[3]);
      }
#pragma unroll
      for (int r = 1; r < R; ++r) b[r] = DIR > 0 ? cmul(b[r], w[r]) : cmulc(b[r], w[r]);
    }
    Bfly<R, DIR>::run(b);
#pragma unroll
    for (int r = 0; r < R; ++r) u[t + NB * r] = b[r];
  }
  if (!LAST) {
    double2 *l2 = reinterpret_cast<double2 *>(lds);
    constexpr int PADR = NS == 1 ? 8 : 1;
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const int j = tid + t * T8, k = j & (NS - 1);
#pragma unroll
      for (int r = 0; r < R; ++r) l2[lds_phys<PADR>((j - k) * R + k + r * NS)] = make_double2(u[t + NB * r].x, u[t + NB * r].y);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const double2 q = l2[lds_phys<PADR>(tid + s * T8)];
      u[s] = {q.x, q.y};
    }
    __syncthreads();
  }
}
template <int DIR> __device__ __forceinline__ void fft8x(c64 (&u)[8], int tid, const double2 *twl, double *lds)
{
  fft8x_pass<8, 1, DIR, false>(u, tid, twl, lds);
  fft8x_pass<8, 8, DIR, false>(u, tid, twl, lds);
  fft8x_pass<8, 64, DIR, false>(u, tid, twl + 24, lds);
  fft8x_pass<4, 512, DIR, true>(u, tid, twl + 24 + 192, lds);
}
} // namespace

// OKIND = 1 / 2: every block of the launch reads one contiguous span and writes one contiguous span of float frames /
// of the planar fp64 rings (launch_dftx checks that on the host by calling pair_span itself, span_contiguous), and the
// kernel has no other path -- which is what lets the compiler count the stores between a load and its use;
// OKIND = 0 (generic): any block, element-wise fifo addressing where a span is split
// (ring wrap, a block half in the ring and half in the caller's buffer, odd channel count) -- same arithmetic, so which
// of the two a block gets changes no bit of its output.
#ifndef RSMP_DFTX_WG // workgroups per CU the register budget is sized for (experiments: 2 = 256 VGPRs, 4 = 128)
#define RSMP_DFTX_WG 3
#endif
template <int LL, int OKIND> __global__ __launch_bounds__(256, RSMP_DFTX_WG) void dftx_kernel(AnyView in, AnyView out, DftArgs a)
{
  constexpr bool GENERIC = OKIND == 0;
  constexpr int P = kP, T8 = P / 8;
  static_assert(T8 == 256 && (LL == 4 || LL == 8), "dftx_kernel: 2048-point pieces, x4 or x8");
  extern __shared__ __attribute__((aligned(16))) double lds[];

  const int tid = threadIdx.x;
  int bl, pair;
  if (!item_map(blockIdx.x, a.nblocks, a.npairs, a.hp, bl, pair)) return; // uniform
  const long long B = a.B0 + bl;
  const PairCh pc = pair_channels(pair, a.C, a.nchs, a.pps_magic);
  const int ca = pc.ca, cb = pc.cb;
  const bool hasb = pc.hasb;

  // ---- the block's P inputs (dft_filter.h:88: the frequency-domain branch reads N/L inputs from the block start)
  c64 x[8];
  {
    const long long base = B * a.q;
    const PairSpan sp = pair_span(in, pair, hasb, base, P, ca);
    if (sp.kind == 1) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float2 f = sp.p2[(tid + s * T8) * sp.fstride];
        x[s] = {(double)f.x, (double)f.y};
      }
    } else if (sp.kind == 2) { // (pb == pa when the pair has one channel: every load is unconditional)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const double xa = sp.pa[tid + s * T8], xb = sp.pb[tid + s * T8];
        x[s] = {xa, hasb ? xb : 0.0};
      }
    } else if constexpr (GENERIC) {
      const ChanRef ia = chan_ref(in, ca), ib = chan_ref(in, hasb ? cb : ca);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const long long e = base + tid + s * T8;
        x[s].x = fifo_get(ia, e);
        x[s].y = hasb ? fifo_get(ib, e) : 0.0;
      }
    } else
      __builtin_trap(); // the host's range check and pair_span disagree
  }
  // twiddles into LDS (global table layout [pass][r-1][k], fft8_regs): rows r = 1, 2, 4 of passes 1 and 2, row 1 of pass 3
  double2 *const twl = reinterpret_cast<double2 *>(lds + kXExch);
  for (int i = tid; i < kXTw; i += 256) {
    int src;
    if (i < 24) src = ((1 << (i >> 3)) - 1) * 8 + (i & 7);
    else if (i < 216) src = 56 + ((1 << ((i - 24) >> 6)) - 1) * 64 + ((i - 24) & 63);
    else src = 504 + (i - 216);
    twl[i] = a.tw_fwd8[src];
  }
  __syncthreads();
  fft8x<-1>(x, tid, twl, lds); // x[s] = X[tid + 256 s]

  // ---- the L components, staged and stored two at a time
  const long long o0 = B * a.Vout;
  const bool whole = o0 >= a.clip_lo && o0 + a.Vout <= a.clip_hi;
  const PairSpan so = whole ? pair_span(out, pair, hasb, a.out_offset + o0, a.Vout, ca) : PairSpan{0, nullptr, 1, nullptr, nullptr, hasb};
  if (!GENERIC && so.kind != OKIND) __builtin_trap(); // the host's range check and pair_span disagree
  const double2 *__restrict__ Gt = a.Gr + tid;
  float2 keep[8];
  // The next component's filter values are loaded BEFORE the current one's results are stored: loads and stores share
  // one in-order counter (vmcnt), so a load issued behind the stores could only be waited for together with their
  // acknowledgements, and the store drain would sit on the critical path of every workgroup (measured: 0.92 -> see DESIGN).
  double2 g[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) g[s] = Gt[s * T8];
#pragma unroll
  for (int r = 0; r < LL; ++r) { // unrolled: the wait counts above need straight-line code
    c64 v[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) v[s] = cmul(x[s], c64{g[s].x, g[s].y});
    __syncthreads(); // the previous component's stage reads are done
    fft8x<+1>(v, tid, twl, lds); // v[s] = y[LL * (tid + 256 s) + r]; ends behind a barrier
    {
      const int rn = r + 1 < LL ? r + 1 : r;
#pragma unroll
      for (int s = 0; s < 8; ++s) g[s] = Gt[rn * P + s * T8];
    }
    if ((GENERIC && so.kind == 1) || OKIND == 1) { // float frames, contiguous: the usual case
      // the even component's values wait in registers for the odd one; both go to the stage (element (n, r & 1) at
      // [2 n + (r & 1)], aliasing the exchange area while no transform runs) and leave it in linear order
      float2 *stg = reinterpret_cast<float2 *>(lds);
      if (!(r & 1)) {
#pragma unroll
        for (int s = 0; s < 8; ++s) keep[s] = make_float2((float)v[s].x, (float)v[s].y);
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s)
          *reinterpret_cast<float4 *>(stg + 2 * (tid + s * T8)) = make_float4(keep[s].x, keep[s].y, (float)v[s].x, (float)v[s].y);
        __syncthreads();
        // idx = tid + 256 j is element (n, c) = (idx >> 1, idx & 1): output m = LL n + (r - 1) + c = mb + 128 LL j
        const unsigned fbytes = 8u * (unsigned)so.fstride;
        const int mb = LL * (tid >> 1) + (tid & 1) + (r - 1);
        char *const ob8 = reinterpret_cast<char *>(so.p2) + (unsigned)mb * fbytes;
        // Branch-free: a slot behind the block's last output repeats the store of the same lane's slot 8 earlier (same
        // value, same address), so that the number of stores is fixed and the wait for the filter values prefetched
        // above can leave all of them outstanding (s_waitcnt vmcnt(16) instead of vmcnt(0)).
#pragma unroll
        for (int jb = 0; jb < 16; jb += 8) {
          float2 f[8];
          int jj[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            jj[j] = (jb + j >= 8 && mb + (jb + j) * (128 * LL) >= a.Vout) ? jb + j - 8 : jb + j;
            f[j] = stg[tid + jj[j] * 256];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (!(RSMP_DFTX_SKIP == 1 && r == 1) && !(RSMP_DFTX_SKIP == 2 && r == LL - 1))
              *reinterpret_cast<float2 *>(ob8 + (unsigned)(jj[j] * (128 * LL)) * fbytes) = f[j];
        }
      }
    } else if ((GENERIC && so.kind == 2) || OKIND == 2) { // planar fp64 rings, contiguous
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int m = LL * (tid + s * T8) + r;
        if (m < a.Vout) {
          so.pa[m] = v[s].x;
          if (hasb) so.pb[m] = v[s].y;
        }
      }
    } else if constexpr (GENERIC) {
      const ChanRef oa = chan_ref(out, ca), ob = chan_ref(out, hasb ? cb : ca);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int m = LL * (tid + s * T8) + r;
        const long long o = o0 + m;
        if (m < a.Vout && o >= a.clip_lo && o < a.clip_hi) {
          fifo_put(oa, a.out_offset + o, v[s].x);
          if (hasb) fifo_put(ob, a.out_offset + o, v[s].y);
        }
      }
    }
  }
}

// Do `len` samples starting at absolute index a0 lie contiguously for EVERY channel pair of the launch?  Asked of pair_span
// itself (the function the kernel uses), for the pairs that can differ in the answer: alignment depends on the pair only
// through its stream (stream stride) -- first pair, first pair of the second stream, last pair.  The lean instances have no
// other path and trap if the kernel's own pair_span ever disagrees with this.
static bool span_contiguous(const AnyView &v, long long a0, long long len, int C)
{
  if (v.is_f32 && ((v.f.nch & 1) || (C & 1))) return false;
  const int npairs = C / 2, hp = v.is_f32 ? v.f.nch / 2 : npairs;
  const int want = v.is_f32 ? 1 : 2;
  for (int pair : {0, hp < npairs ? hp : 0, npairs - 1})
    if (pair_span(v, pair, true, a0, len).kind != want) return false;
  return true;
}

template <int LL, int OKIND> static hipError_t launch_dftx_run(const AnyView &in, const AnyView &out, const DftArgs &a, hipStream_t st)
{
  static DynLdsOnce attr;
  if (hipError_t e = attr.set(reinterpret_cast<const void *>(&dftx_kernel<LL, OKIND>), int(kXLdsBytes)); e != hipSuccess) return e;
  dim3 grid(item_grid(a.nblocks, a.npairs, a.hp)), block(256);
  hipLaunchKernelGGL((dftx_kernel<LL, OKIND>), grid, block, kXLdsBytes, st, in, out, a);
  return hipGetLastError();
}

// runs of blocks with contiguous spans on both sides go to the lean instance, the others (a ring wrap, the block that
// straddles the previous push's tail and the caller's buffer) to the generic one
template <int LL> static hipError_t launch_dftx_t(const AnyView &in, const AnyView &out, const DftArgs &a, hipStream_t st)
{
  DftArgs b = a;
  b.hp = frame_pairs(in, out, a.C);
  b.npairs = pair_count(a.C, a.nchs);
  b.pps_magic = pair_magic(a.C, a.nchs);
  const bool clip_all = a.clip_lo <= a.B0 * (long long)a.Vout && (a.B0 + a.nblocks) * (long long)a.Vout <= a.clip_hi;
  auto fast = [&](int k) {
    const long long B = a.B0 + k;
    return clip_all && span_contiguous(in, B * a.q, kP, a.C) && span_contiguous(out, a.out_offset + B * a.Vout, a.Vout, a.C);
  };
  for (int k = 0; k < a.nblocks;) {
    const bool f = fast(k);
    int e = k + 1;
    while (e < a.nblocks && fast(e) == f) ++e;
    b.B0 = a.B0 + k;
    b.nblocks = e - k;
    const hipError_t rc = !f ? launch_dftx_run<LL, 0>(in, out, b, st)
                         : out.is_f32 ? launch_dftx_run<LL, 1>(in, out, b, st) : launch_dftx_run<LL, 2>(in, out, b, st);
    if (rc != hipSuccess) return rc;
    k = e;
  }
  return hipSuccess;
}

// x L in the frequency domain with blocks of L * 2048 points that keep their length
bool dftx_supported(int log2n, int log2p, int log2nd)
{
  return !knobs().no_dftx && log2p == kXP && log2nd == log2n && (log2n == 13 || log2n == 14);
}

hipError_t launch_dftx(int log2n, bool src_f32, bool dst_f32, const F32View &sf, const F64View &sd, const F32View &df,
                       const F64View &dd, const DftArgs &a, hipStream_t st, const char **kname)
{
  const AnyView in = make_view(src_f32, sf, sd), out = make_view(dst_f32, df, dd);
  if (log2n == 13) {
    if (kname) *kname = "rsmp::dftx_kernel<4>";
    return launch_dftx_t<4>(in, out, a, st);
  }
  if (log2n == 14) {
    if (kname) *kname = "rsmp::dftx_kernel<8>";
    return launch_dftx_t<8>(in, out, a, st);
  }
  return hipErrorInvalidValue;
}

} // namespace rsmp
