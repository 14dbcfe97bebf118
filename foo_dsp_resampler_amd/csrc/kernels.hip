// HIP kernels for the stage chain (gfx950 / MI355X).
//
//   dft_kernel  : overlap-save FFT-FIR stage (reference: rate/dft_filter.h:60-190), one workgroup per
//                 (block, channel PAIR): the two channels ride as real / imaginary part of one complex
//                 transform, so no real-FFT split pass is needed and stereo frames load as one complex.
//   poly_kernel : polyphase FIR stage, orders 0..3 (reference: rate/rate_filters_generic.h:272-504).
//   half_kernel : half-band decimate-by-2 (reference: rate/rate_filters_generic.h:80-249).
//
// All arithmetic is fp64 (the reference's "Best" path is `sox_sample_t = double`, rate/rate_base.h:63-67);
// float32 only at the caller-facing ends, converted with round-to-nearest-even like the C cast in
// rate/rate_base.h:559-563.
#include "kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <atomic>

#include "fft_device.hpp"
#include "fifo_device.hpp"

namespace rsmp {

// ---------------------------------------------------------------------------------------------
// DFT stage
// ---------------------------------------------------------------------------------------------
template <bool SPLIT> struct LdsCx {
  double *p;
  __device__ __forceinline__ void put(int idx, c64 v, int h) const
  {
    if (!SPLIT) reinterpret_cast<double2 *>(p)[idx] = make_double2(v.x, v.y);
    else p[idx] = h ? v.y : v.x;
  }
  __device__ __forceinline__ void get(int idx, c64 &v, int h) const
  {
    if (!SPLIT) {
      const double2 q = reinterpret_cast<const double2 *>(p)[idx];
      v = {q.x, q.y};
    } else if (h) v.y = p[idx];
    else v.x = p[idx];
  }
};

// SP ("stream pairs"): the channel pairs of a batch handle with an odd channel count per stream, which must not straddle
// streams (pair_channels).  Its own instance because the general pair -> channel map costs the 128-VGPR instances a few
// registers they do not have (dft_kernel<14, 13, 14>: 10 -> 14 spilled VGPRs, +4 % time); every other handle runs SP = false.
template <int LOG2N, int LOG2P, int LOG2ND, bool SP>
__global__ __launch_bounds__((1 << LOG2N) / 16, (LOG2N == 13 && LOG2ND == 13) ? 4 : (LOG2N == 12 && LOG2ND >= 11) ? 3 : 1) void dft_kernel(AnyView in, AnyView out, DftArgs a)
{
  constexpr int N = 1 << LOG2N, P = 1 << LOG2P, ND = 1 << LOG2ND;
  constexpr int T = N / 16, TF = P / 16, TD = ND / 16;
  constexpr bool SPLIT = LOG2N >= 14;
  // 8192-point blocks that keep their length: exchanges in two half rounds (70 KB instead of 139 KB of LDS), so two
  // workgroups share a CU
  // 16384-point blocks that keep their length: half rounds of 16-byte elements instead of real / imaginary rounds
  // of 8-byte ones (same 128 KB, ds_*_b128 moves 1 KB in 8.4 cycles where ds_*_b64 needs 12: +7 % on the 44.1k->192k chain)
  // 4096-point blocks (kept length, or halved in the frequency domain): half rounds as well, so that three workgroups
  // fit a CU instead of two
  constexpr int XMODE = (LOG2N == 14 && LOG2ND == 14) ? 2 : SPLIT ? 1 : (LOG2N == 13 && LOG2ND == 13) ? 2 : (LOG2N == 12 && LOG2ND >= 11) ? 2 : 0;
  constexpr int ROUNDS = SPLIT ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) double lds[];

  const int tid = threadIdx.x;
  int bl, pair;
  if (!item_map(blockIdx.x, a.nblocks, a.npairs, a.hp, bl, pair)) return; // uniform
  const long long B = a.B0 + bl;
  int ca = 2 * pair, cb = ca + 1;
  bool hasb = cb < a.C;
  if constexpr (SP) {
    const PairCh pc = pair_channels(pair, a.C, a.nchs, a.pps_magic);
    ca = pc.ca;
    cb = pc.cb;
    hasb = pc.hasb;
  }
  const ChanRef ia = chan_ref(in, ca), ib = chan_ref(in, hasb ? cb : ca);

  c64 v[16];
  const bool fwd_active = tid < TF;
  // x4 upsampling in the frequency domain on up to 8192-point blocks: the forward transform has a quarter of the
  // points; run it 8 points per thread on twice as many threads (fft8_regs) -- a thread tid < 2*TF then owns
  // Zp[tid + s*2*TF], and the inverse transform's thread t needs Zp[(t mod T) + T*j], j < 4: its own even slots for
  // t < 2*TF, the odd slots of thread t - 2*TF otherwise
  // (with x2 upsampling the same plan uses ALL threads, and a thread ends up with exactly the 8 values it needs)
  constexpr bool F8 = XMODE != 1 && (LOG2N - LOG2P == 2 || LOG2N - LOG2P == 1) && LOG2ND == LOG2N && LOG2P >= 6 && LOG2P <= 13;
  constexpr int T8 = P / 8; // threads of the 8-points-per-thread forward transform
  // Frequency-domain decimation by 2 (same block length in, half out): the kept bins k' = t + s'*T (s' < 8) of thread t
  // are its own slots 0-3 and 12-15 of the N-point spectrum, so no exchange is needed, and the ND-point inverse
  // transform runs 8 points per thread on ALL threads instead of 16 points on half of them.
  constexpr bool D8 = !SPLIT && LOG2P == LOG2N && LOG2ND == LOG2N - 1 && LOG2ND >= 6 && LOG2ND <= 12;
  c64 u8[8];
  if constexpr (F8) {
    if (tid < T8) {
      const long long base = B * a.q;
      const PairSpan sp = base + P <= a.in_limit ? pair_span(in, pair, hasb, base, P, ca) : PairSpan{0, nullptr, 1, nullptr, nullptr, hasb};
      if (sp.kind) {
        span_load<8>(sp, tid, T8, u8);
      } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const long long e = base + tid + s * T8;
          const bool have = e < a.in_limit;
          u8[s].x = have ? fifo_get(ia, e) : 0.0;
          u8[s].y = have && hasb ? fifo_get(ib, e) : 0.0;
        }
      }
    }
  } else
  if (fwd_active) {
    if (LOG2P < LOG2N || a.L == 1) {
      const long long base = B * a.q;
      const PairSpan sp = base + P <= a.in_limit ? pair_span(in, pair, hasb, base, P, ca) : PairSpan{0, nullptr, 1, nullptr, nullptr, hasb};
      if (sp.kind) {
        span_load<16>(sp, tid, TF, v);
      } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const long long e = base + tid + s * TF;
          const bool have = e < a.in_limit;
          v[s].x = have ? fifo_get(ia, e) : 0.0;
          v[s].y = have && hasb ? fifo_get(ib, e) : 0.0;
        }
      }
    } else { // time-domain zero stuffing (dft_filter.h:109-115) in absolute coordinates
      const long long U = B * a.V;
      const long long j0 = (U - a.c0 + a.L - 1) / a.L;
      const int remL = (int)(j0 * a.L + a.c0 - U);
      // slot d = remL + L q holds input j0 + q; everything else is a stuffed zero
      const int nin = (N - remL + a.L - 1) / a.L; // inputs the block touches
      const PairSpan sp = j0 + nin <= a.in_limit ? pair_span(in, pair, hasb, j0, nin, ca) : PairSpan{0, nullptr, 1, nullptr, nullptr, hasb};
      if (sp.kind) {
        // contiguous inputs: one unconditional (clamped) load per slot, all issued before the first is used, the quotient by a
        // multiply (d < 2^32 / L).  The element-wise path below costs a division and a memory round trip per slot.
        const unsigned magic = 0xffffffffu / (unsigned)a.L + 1u;
        int q[16];
        bool take[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int d = tid + s * TF - remL;
          const int qq = (int)__umulhi((unsigned)max(d, 0), magic);
          take[s] = d >= 0 && qq * a.L == d;
          q[s] = take[s] ? qq : 0;
        }
        if (sp.kind == 1) {
#pragma unroll
          for (int s = 0; s < 16; ++s) {
            const float2 f = sp.p2[q[s] * sp.fstride];
            v[s] = {(double)f.x, (double)f.y};
          }
        } else {
#pragma unroll
          for (int s = 0; s < 16; ++s) v[s] = {sp.pa[q[s]], sp.pb[q[s]]};
          if (!hasb) {
#pragma unroll
            for (int s = 0; s < 16; ++s) v[s].y = 0.0;
          }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s)
          if (!take[s]) v[s] = {0.0, 0.0};
      } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int d = tid + s * TF - remL;
        v[s] = {0.0, 0.0};
        if (d >= 0 && d % a.L == 0) {
          const long long e = j0 + d / a.L;
          if (e < a.in_limit) {
            v[s].x = fifo_get(ia, e);
            v[s].y = hasb ? fifo_get(ib, e) : 0.0;
          }
        }
      }
      }
    }
  }

  if constexpr (F8) {
    // (all threads take part in the barriers of fft8_regs; threads >= T8 carry zeros)
    if (tid >= T8) {
#pragma unroll
      for (int s = 0; s < 8; ++s) u8[s] = {0.0, 0.0};
    }
    fft8_regs_masked<LOG2P, -1>(u8, tid, tid < T8, a.tw_fwd8, lds);
    if constexpr (LOG2N - LOG2P == 1) { // T8 == T: Z[tid + s*T] = Zp[tid + (s & 7)*T] is already here
      __syncthreads(); // the inverse transform's exchange reuses the LDS the forward one just read
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const double2 g = a.G[tid + s * T];
        v[s] = cmul(u8[s & 7], c64{g.x, g.y});
      }
    } else {
      double2 *l2 = reinterpret_cast<double2 *>(lds);
      if (tid < T8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) l2[tid + j * T8] = make_double2(u8[2 * j + 1].x, u8[2 * j + 1].y);
      }
      __syncthreads();
      c64 z[4];
      if (tid < T8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) z[j] = u8[2 * j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double2 q = l2[tid - T8 + j * T8];
          z[j] = {q.x, q.y};
        }
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const double2 g = a.G[tid + s * T];
        v[s] = cmul(z[s & 3], c64{g.x, g.y});
      }
    }
  } else {
  fft_regs<LOG2P, -1, (XMODE == 2 && LOG2P < LOG2N) ? 0 : XMODE>(v, tid, fwd_active, a.tw_fwd, lds);
  }

  const LdsCx<SPLIT> L{lds};
  if constexpr (F8) {
  } else if constexpr (LOG2P == LOG2N && LOG2ND == LOG2N) {
    // same thread/slot layout on both sides: multiply in registers
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const double2 g = a.G[tid + s * T];
      v[s] = cmul(v[s], c64{g.x, g.y});
    }
  } else if constexpr (LOG2P < LOG2N) {
    // spectrum of the zero-stuffed block = periodic repetition of the P-point spectrum
    // (dft_filter.h:88-103), then the filter
#pragma unroll
    for (int h = 0; h < ROUNDS; ++h) {
      if (fwd_active) {
#pragma unroll
        for (int s = 0; s < 16; ++s) L.put(tid + s * TF, v[s], h);
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < 16; ++s) L.get((tid + s * T) & (P - 1), v[s], h);
      __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const double2 g = a.G[tid + s * T];
      v[s] = cmul(v[s], c64{g.x, g.y});
    }
  } else {
    // frequency-domain decimation by 2^m (dft_filter.h:157-188): keep the lowest and highest
    // ND/2 bins of the filtered spectrum; the new Nyquist bin is the mean of its two images.
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const double2 g = a.G[tid + s * T];
      v[s] = cmul(v[s], c64{g.x, g.y});
    }
    if constexpr (D8) {
      c64 d8[8];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        d8[s] = v[s];
        d8[s + 4] = v[s + 12];
      }
      if (tid == 0) d8[4] = {0.5 * (v[4].x + v[12].x), 0.5 * (v[4].y + v[12].y)}; // new Nyquist bin: mean of its two images
      fft8_regs<LOG2ND, +1>(d8, tid, a.tw_inv8, lds);
      const ChanRef oa = chan_ref(out, ca), ob = chan_ref(out, hasb ? cb : ca);
      const long long o0 = B * a.Vout;
      const PairSpan so = (o0 >= a.clip_lo && o0 + a.Vout <= a.clip_hi) ? pair_span(out, pair, hasb, a.out_offset + o0, a.Vout, ca)
                                                                       : PairSpan{0, nullptr, 1, nullptr, nullptr, hasb};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int n = tid + s * T;
        const long long o = o0 + n;
        if (so.kind) {
          if (n < a.Vout) so.put(n, d8[s].x, d8[s].y);
        } else if (n < a.Vout && o >= a.clip_lo && o < a.clip_hi) {
          fifo_put(oa, a.out_offset + o, d8[s].x);
          if (hasb) fifo_put(ob, a.out_offset + o, d8[s].y);
        }
      }
      return;
    }
    c64 nyq = {0.0, 0.0};
#pragma unroll
    for (int h = 0; h < ROUNDS; ++h) {
#pragma unroll
      for (int s = 0; s < 16; ++s) L.put(tid + s * T, v[s], h);
      __syncthreads();
      if (tid < TD) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int k = tid + s * TD;
          L.get(k < ND / 2 ? k : N - ND + k, v[s], h);
        }
        if (tid == 0) L.get(ND / 2, nyq, h);
      }
      __syncthreads();
    }
    if (tid == 0) v[8] = {0.5 * (v[8].x + nyq.x), 0.5 * (v[8].y + nyq.y)};
  }

  const bool inv_active = tid < TD;
  // twiddles of the next pass loaded ahead of each exchange where the registers allow it (not at the 16384-point instance's
  // 128-VGPR cap: 10 -> 36 spilled VGPRs)
  fft_regs<LOG2ND, +1, XMODE, (LOG2N <= 13 ? 8 : 0)>(v, tid, inv_active, a.tw_inv, lds);

  if (inv_active) {
    const ChanRef oa = chan_ref(out, ca), ob = chan_ref(out, hasb ? cb : ca);
    if (a.M == 1) {
      const long long o0 = B * a.Vout;
      const PairSpan so = (o0 >= a.clip_lo && o0 + a.Vout <= a.clip_hi) ? pair_span(out, pair, hasb, a.out_offset + o0, a.Vout, ca)
                                                                       : PairSpan{0, nullptr, 1, nullptr, nullptr, hasb};
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int n = tid + s * TD;
        const long long o = o0 + n;
        if (so.kind) {
          if (n < a.Vout) so.put(n, v[s].x, v[s].y);
        } else if (n < a.Vout && o >= a.clip_lo && o < a.clip_hi) {
          fifo_put(oa, a.out_offset + o, v[s].x);
          if (hasb) fifo_put(ob, a.out_offset + o, v[s].y);
        }
      }
    } else { // time-domain decimation (dft_filter.h:148-154): keep filtered samples Y with Y % M == 0
      if constexpr (LOG2N >= 14) { // (the 16384-point instances have no registers to spare: the plain form)
        const long long Y0 = B * a.V;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int n = tid + s * TD;
          const long long Y = Y0 + n;
          if (n < a.V && Y % a.M == 0) {
            const long long o = Y / a.M;
            if (o >= a.clip_lo && o < a.clip_hi) {
              fifo_put(oa, a.out_offset + o, v[s].x);
              if (hasb) fifo_put(ob, a.out_offset + o, v[s].y);
            }
          }
        }
      } else {
      // kept sample Y = Y0 + n is output Y / M: one 64-bit division per workgroup (Y0), per sample a multiply (n + r0 < 2^32 / M)
      const long long Y0 = B * a.V, ob0 = Y0 / a.M;
      const int r0 = (int)(Y0 - ob0 * a.M);
      const unsigned magic = 0xffffffffu / (unsigned)a.M + 1u;
      const long long o_first = ob0 + (r0 ? 1 : 0);                    // first output of the block
      const int cnt = (int)((Y0 + a.V - 1) / a.M - o_first) + 1;       // outputs of the block (V >= M)
      const PairSpan so = (o_first >= a.clip_lo && o_first + cnt <= a.clip_hi) ? pair_span(out, pair, hasb, a.out_offset + o_first, cnt, ca)
                                                                         : PairSpan{0, nullptr, 1, nullptr, nullptr, hasb};
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int n = tid + s * TD;
        const unsigned t = (unsigned)(n + r0);
        const int qq = (int)__umulhi(t, magic);
        if (n < a.V && (unsigned)qq * (unsigned)a.M == t) {
          const long long o = ob0 + qq;
          if (so.kind) so.put((int)(o - o_first), v[s].x, v[s].y);
          else if (o >= a.clip_lo && o < a.clip_hi) {
            fifo_put(oa, a.out_offset + o, v[s].x);
            if (hasb) fifo_put(ob, a.out_offset + o, v[s].y);
          }
        }
      }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Polyphase stage.  One workgroup = one tile of consecutive outputs of one channel; the input window
// of the tile is staged in LDS as fp64, coefficients come from the (L2-resident) table.
// ---------------------------------------------------------------------------------------------
template <int ORDER> __global__ __launch_bounds__(256) void poly_kernel(AnyView in, AnyView out, PolyArgs a)
{
  extern __shared__ __attribute__((aligned(16))) double win[];
  const int tid = threadIdx.x, c = blockIdx.y;
  const long long i0 = (long long)blockIdx.x * a.tile;
  const int cnt = (int)min((long long)a.tile, a.count - i0);
  const ChanRef src = chan_ref(in, c), dst = chan_ref(out, c);

  const long long A0 = a.at + i0 * a.step, A1 = a.at + (i0 + cnt - 1) * a.step;
  const long long q0 = ORDER == 0 ? A0 / a.L : (A0 >> 32);
  const long long q1 = ORDER == 0 ? A1 / a.L : (A1 >> 32);
  const int wlen = (int)(q1 - q0) + a.n;
  for (int ib = tid; ib < wlen; ib += 256 * 4) { // 4 loads in flight per thread (a plain loop waits for each before its LDS store)
    double t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = fifo_get(src, a.rd + q0 + min(ib + 256 * j, wlen - 1));
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (ib + 256 * j < wlen) win[ib + 256 * j] = t[j];
  }
  // rational stage: the whole coefficient table rides in LDS behind the window when it fits (a.tab_lds)
  const double *__restrict__ tab = a.tab;
  if (ORDER == 0 && a.tab_lds) {
    double *t = win + a.win;
    for (int i = tid; i < a.L * a.n; i += 256) t[i] = a.tab[i];
    tab = t;
  }
  __syncthreads();

  if (ORDER == 0) {
    // clock of output i0 + u relative to the tile's first window: t = t0 + u * step, all in 32 bits; a thread
    // walks u = tid, tid + 256, ... with an incremental (q, phase) update, one division per thread
    const unsigned L = (unsigned)a.L, step = (unsigned)a.step;
    const unsigned t0 = (unsigned)(A0 - q0 * a.L) + (unsigned)tid * step;
    unsigned q = t0 / L, ph = t0 - q * L;
    const unsigned dq = (256u * step) / L, dph = 256u * step - dq * L;
    for (int u = tid; u < cnt; u += 256) {
      const double *cf = tab + ph * a.n;
      const double *x = win + q;
      double sum = 0.0;
#pragma unroll 4
      for (int j = 0; j < a.n; ++j) sum = fma(cf[j], x[j], sum);
      fifo_put(dst, a.out_abs + i0 + u, sum);
      q += dq;
      ph += dph;
      if (ph >= L) { ph -= L; ++q; }
    }
  } else {
    for (int u = tid; u < cnt; u += 256) {
      const long long A = a.at + (i0 + u) * a.step;
      const long long q = A >> 32;
      const unsigned frac = (unsigned)A;
      const int ph = (int)(frac >> (32 - a.phase_bits));
      const double t = (double)(unsigned)(frac << a.phase_bits) * (1.0 / 4294967296.0);
      const double *__restrict__ cf = a.tab + (long long)ph * a.n * (ORDER + 1);
      const double *x = win + (q - q0);
      double sum = 0.0;
#pragma unroll 4
      for (int j = 0; j < a.n; ++j) { // coefficient loads of four taps in flight
        double w = cf[j * (ORDER + 1)];
#pragma unroll
        for (int o = 1; o <= ORDER; ++o) w = fma(w, t, cf[j * (ORDER + 1) + o]);
        sum = fma(w, x[j], sum);
      }
      fifo_put(dst, a.out_abs + i0 + u, sum);
    }
  }
}

// Interpolated polyphase stage (orders 1-3), cooperative variant.  An output's coefficients are one contiguous row of
// n*(ORDER+1) doubles (768 B at Best), a different row for every output; with one output per lane each 8-byte load
// instruction touches 64 cache lines.  Here 8 lanes share an output: lane l takes taps [l*n/8, (l+1)*n/8), i.e. a
// contiguous n/8*(ORDER+1)-double piece of the row (16-byte loads, 8 lanes = one run of the row), and the 8 partial sums
// are combined with three DPP/shuffle steps -- north_star's "per-phase tap products reduced across the wavefront".
// (Summation order differs from the reference's sequential loop by that tree: results agree to fp64 rounding.)
template <int ORDER> __global__ __launch_bounds__(256) void poly_coop_kernel(AnyView in, AnyView out, PolyArgs a)
{
  extern __shared__ __attribute__((aligned(16))) double win[];
  const int tid = threadIdx.x, c = blockIdx.y;
  const long long i0 = (long long)blockIdx.x * a.tile;
  const int cnt = (int)min((long long)a.tile, a.count - i0);
  const ChanRef src = chan_ref(in, c), dst = chan_ref(out, c);
  const long long A0 = a.at + i0 * a.step, A1 = a.at + (i0 + cnt - 1) * a.step;
  const long long q0 = A0 >> 32, q1 = A1 >> 32;
  const int wlen = (int)(q1 - q0) + a.n;
  for (int ib = tid; ib < wlen; ib += 256 * 4) { // 4 loads in flight per thread (a plain loop waits for each before its LDS store)
    double t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = fifo_get(src, a.rd + q0 + min(ib + 256 * j, wlen - 1));
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (ib + 256 * j < wlen) win[ib + 256 * j] = t[j];
  }
  __syncthreads();

  const int sub = tid & 7, tpl = a.n >> 3; // lane within its output's group of 8; taps per lane
  for (int u0 = 0; u0 < cnt; u0 += 32) {  // 32 outputs per pass of the workgroup
    const int u = u0 + (tid >> 3);
    const bool live = u < cnt;
    const long long A = a.at + (i0 + (live ? u : 0)) * a.step;
    const unsigned frac = (unsigned)A;
    const int ph = (int)(frac >> (32 - a.phase_bits));
    const double t = (double)(unsigned)(frac << a.phase_bits) * (1.0 / 4294967296.0);
    const double *cf = a.tab + ((long long)ph * a.n + sub * tpl) * (ORDER + 1);
    const double *x = win + ((A >> 32) - q0) + sub * tpl;
    double sum = 0.0;
    for (int j = 0; j < tpl; ++j) {
      double w;
      if constexpr (ORDER == 3) {
        const double2 c01 = *reinterpret_cast<const double2 *>(cf + 4 * j), c23 = *reinterpret_cast<const double2 *>(cf + 4 * j + 2);
        w = fma(fma(fma(c01.x, t, c01.y), t, c23.x), t, c23.y);
      } else if constexpr (ORDER == 1) {
        const double2 c01 = *reinterpret_cast<const double2 *>(cf + 2 * j);
        w = fma(c01.x, t, c01.y);
      } else {
        w = cf[j * (ORDER + 1)];
#pragma unroll
        for (int o = 1; o <= ORDER; ++o) w = fma(w, t, cf[j * (ORDER + 1) + o]);
      }
      sum = fma(w, x[j], sum);
    }
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    sum += __shfl_xor(sum, 4);
    if (live && sub == 0) fifo_put(dst, a.out_abs + i0 + u, sum);
  }
}

// Interpolated polyphase stage (orders 1-3), SHARED-ROW variant.  All channels of a handle run on one clock
// (rate_base.h:533-540: every channel gets the same rate_t parameters), so output i has the same phase and the same
// interpolation fraction in every channel: the interpolated coefficient row w_i[j] = Horner(coefs[phase_i][j], x_i)
// (rate_filters_generic.h:416-424) depends on i only.  A workgroup computes the rows of a tile of kPolyiTile outputs ONCE
// (into LDS) and applies them to kPolyiCh channels whose windows are staged next to them: the table reads (768 B per row at
// Best, a different row per output) and the Horner work are paid once per kPolyiCh channels instead of once per channel.
// Taps are summed in the reference's order (no cross-lane reduction).
constexpr int kPolyiTile = 128, kPolyiCh = 16;
template <int ORDER> __global__ __launch_bounds__(256) void polyi_kernel(AnyView in, AnyView out, PolyArgs a)
{
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int tid = threadIdx.x, n = a.n;
  const long long i0 = (long long)blockIdx.x * kPolyiTile;
  const int cnt = (int)min((long long)kPolyiTile, a.count - i0);
  const long long A0 = a.at + i0 * a.step, A1 = a.at + (i0 + cnt - 1) * a.step;
  const long long q0 = A0 >> 32, q1 = A1 >> 32;
  const int wlen = (int)(q1 - q0) + n;
  const int wstride = a.win;          // doubles per channel window, odd (bank spread across channels)
  double *wrow = sh;                  // [kPolyiTile][n]
  double *win = sh + kPolyiTile * n;  // [kPolyiCh][wstride]
  // (no loop divides by a run-time value: a thread is (row, column) of every tile it touches, and works out a channel's
  // addressing once per channel, not once per element)
  // (... and the loads of a phase are issued in batches before anything waits for them: one L2 round trip per batch, not
  // per element -- the row phase used to be 16 dependent table reads per thread, half of the kernel's time)
  // The rows depend on the output index only: computed ONCE per workgroup, then applied to one group of kPolyiCh channels
  // after the other (blockIdx.y, blockIdx.y + gridDim.y, ...: the launcher keeps just enough groups side by side to fill
  // the GPU).  With a group per workgroup, as first written, the 16 workgroups of a tile each read the same 98 KB of table
  // rows and ran the same Horner steps.
  { // interpolated rows: 32 lanes along a row (n <= 32), 8 outputs per pass, 8 passes' table reads in flight
    const int j = min(tid & 31, n - 1);
    const bool jok = (tid & 31) < n;
    constexpr int RB = 8;
    for (int ub = tid >> 5; ub < cnt; ub += 8 * RB) {
      double2 c01[RB], c23[RB];
      double tt[RB];
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int u = min(ub + 8 * i, cnt - 1);
        const long long A = a.at + (i0 + u) * a.step;
        const unsigned frac = (unsigned)A;
        const int ph = (int)(frac >> (32 - a.phase_bits));
        tt[i] = (double)(unsigned)(frac << a.phase_bits) * (1.0 / 4294967296.0);
        const double *__restrict__ cf = a.tab + ((long long)ph * n + j) * (ORDER + 1);
        if constexpr (ORDER == 3) {
          c01[i] = *reinterpret_cast<const double2 *>(cf);
          c23[i] = *reinterpret_cast<const double2 *>(cf + 2);
        } else if constexpr (ORDER == 1) {
          c01[i] = *reinterpret_cast<const double2 *>(cf);
          c23[i] = make_double2(0.0, 0.0);
        } else { // ORDER == 2: c0, c1 | c2
          c01[i] = make_double2(cf[0], cf[1]);
          c23[i] = make_double2(cf[2], 0.0);
        }
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int u = ub + 8 * i;
        const double t = tt[i];
        double w;
        if constexpr (ORDER == 3) w = fma(fma(fma(c01[i].x, t, c01[i].y), t, c23[i].x), t, c23[i].y);
        else if constexpr (ORDER == 1) w = fma(c01[i].x, t, c01[i].y);
        else w = fma(fma(c01[i].x, t, c01[i].y), t, c23[i].x);
        if (u < cnt && jok) wrow[u * n + j] = w;
      }
    }
  }
  for (int c0 = blockIdx.y * kPolyiCh; c0 < a.C; c0 += gridDim.y * kPolyiCh) {
  const int nc = min(kPolyiCh, a.C - c0);
  { // channel windows: 64 lanes along a window, 4 channels per pass, up to 4 loads in flight per thread
    const int k0 = tid & 63;
    for (int cl = tid >> 6; cl < nc; cl += 4) {
      const ChanRef src = chan_ref(in, c0 + cl);
      for (int kb = k0; kb < wlen; kb += 256) {
        double t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = fifo_get(src, a.rd + q0 + min(kb + 64 * i, wlen - 1));
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (kb + 64 * i < wlen) win[cl * wstride + kb + 64 * i] = t[i];
      }
    }
  }
  __syncthreads(); // rows (first pass) and windows are in place
  { // outputs: neighbouring lanes = neighbouring channels of one output (rows broadcast), 16 outputs per pass.
    // (Four channels per thread -- a row element read once per four multiply-adds -- was measured 8 % slower.)
    const int cl = tid & (kPolyiCh - 1);
    if (cl < nc) {
      const ChanRef dst = chan_ref(out, c0 + cl);
      const double *xw = win + cl * wstride;
      for (int u = tid >> 4; u < cnt; u += 256 / kPolyiCh) {
        const long long A = a.at + (i0 + u) * a.step;
        const double *x = xw + (int)((A >> 32) - q0);
        const double *w = wrow + u * n;
        double sum = 0.0;
#pragma unroll 8
        for (int j = 0; j < n; ++j) sum = fma(w[j], x[j], sum);
        fifo_put(dst, a.out_abs + i0 + u, sum);
      }
    }
  }
  __syncthreads(); // the next group's windows overwrite these
  }
}

// ---------------------------------------------------------------------------------------------
// Half-band stage: y[i] = .5 x[c] + sum_k coef[k] (x[c-(2k+1)] + x[c+(2k+1)]),  c = rd + pre + 2 i
// One workgroup = a tile of kHalfTile outputs of one channel.  Its 2*tile + 4*ncoef input samples are staged in LDS
// de-interleaved by parity (the centre taps live on one parity, all other taps on the other), so that consecutive
// lanes read consecutive 8-byte elements; sums are formed in the order of the reference's loop.
// ---------------------------------------------------------------------------------------------
constexpr int kHalfTile = 2048;
// NC = number of coefficient pairs (8..13, rate_filters_generic.h:31-70).  A thread produces 4 consecutive outputs: their
// 2*NC + 3 off-centre inputs are fetched once (16-byte LDS reads) and reused from registers -- 8.5 LDS reads per output
// instead of 2*NC + 1.
template <int NC> __global__ __launch_bounds__(256) void half_kernel(AnyView in, AnyView out, HalfArgs a)
{
  __shared__ __attribute__((aligned(16))) double plane[2][kHalfTile + 32]; // [parity relative to the window start][index / 2]
  const int tid = threadIdx.x, c = blockIdx.y;
  const ChanRef src = chan_ref(in, c), dst = chan_ref(out, c);
  const long long i0 = (long long)blockIdx.x * kHalfTile;
  const int cnt = (int)min((long long)kHalfTile, a.count - i0);
  constexpr int reach = 2 * NC - 1;                        // farthest tap from the centre
  const long long w0 = a.rd + a.pre + 2 * i0 - reach - 1;  // window start: centre of output 0 sits at w0 + reach + 1 (even)
  const int wlen = 2 * cnt + 2 * reach + 1;
  // window into the two parity planes, 6 loads in flight per thread and 3 batches.  With the window contiguous in one buffer
  // (chan_span) the loads are plain and unconditional; fifo_get's ring / caller-buffer test per element would put every
  // load behind a branch and a wait of its own (17 memory round trips in a row).
  const ChanSpan cs = chan_span(in, c, w0, wlen);
  for (int ib = tid; ib < 2 * (kHalfTile + 32); ib += 256 * 6) {
    double t[6];
    if (cs.kind == 1) {
#pragma unroll
      for (int j = 0; j < 6; ++j) t[j] = (double)cs.p32[min(ib + 256 * j, wlen - 1) * cs.stride32];
    } else if (cs.kind == 2) {
#pragma unroll
      for (int j = 0; j < 6; ++j) t[j] = cs.p64[min(ib + 256 * j, wlen - 1)];
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j) t[j] = ib + 256 * j < wlen ? fifo_get(src, w0 + ib + 256 * j) : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int i = ib + 256 * j;
      if (i < 2 * (kHalfTile + 32)) plane[i & 1][i >> 1] = i < wlen ? t[j] : 0.0;
    }
  }
  __syncthreads();
  // centre of output u: even plane [u + NC]; its taps -+(2k+1): odd plane [u + NC - 1 - k] and [u + NC + k]
  for (int u4 = 4 * tid; u4 < cnt; u4 += 4 * 256) {
    double o[2 * NC + 4]; // odd plane [u4 .. u4 + 2 NC + 3]
#pragma unroll
    for (int j = 0; j < NC + 2; ++j) {
      const double2 q = *reinterpret_cast<const double2 *>(&plane[1][u4 + 2 * j]);
      o[2 * j] = q.x;
      o[2 * j + 1] = q.y;
    }
    double ctr[4];
    if (NC & 1) { // even-plane index u4 + NC is odd: 16-byte aligned pairs start one element earlier
      const double2 q0 = *reinterpret_cast<const double2 *>(&plane[0][u4 + NC - 1]), q1 = *reinterpret_cast<const double2 *>(&plane[0][u4 + NC + 1]),
                    q2 = *reinterpret_cast<const double2 *>(&plane[0][u4 + NC + 3]);
      ctr[0] = q0.y; ctr[1] = q1.x; ctr[2] = q1.y; ctr[3] = q2.x;
    } else {
      const double2 q0 = *reinterpret_cast<const double2 *>(&plane[0][u4 + NC]), q1 = *reinterpret_cast<const double2 *>(&plane[0][u4 + NC + 2]);
      ctr[0] = q0.x; ctr[1] = q0.y; ctr[2] = q1.x; ctr[3] = q1.y;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      double sum = ctr[m] * 0.5;
#pragma unroll
      for (int k = 0; k < NC; ++k) sum += (o[m + NC - 1 - k] + o[m + NC + k]) * a.coef[k];
      if (u4 + m < cnt) fifo_put(dst, a.out_abs + i0 + u4 + m, sum);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// generic fifo-to-fifo copy of an absolute index range (small: ring growth, input carry, device pull)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_kernel(AnyView in, AnyView out, long long a0, long long a1)
{
  const int c = blockIdx.y;
  const ChanRef src = chan_ref(in, c), dst = chan_ref(out, c);
  for (long long a = a0 + (long long)blockIdx.x * 256 + threadIdx.x; a < a1; a += (long long)gridDim.x * 256)
    fifo_put(dst, a, fifo_get(src, a));
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <int LOG2N, int LOG2P, int LOG2ND, bool SP>
static hipError_t launch_dft_t(const AnyView &in, const AnyView &out, const DftArgs &a, hipStream_t st)
{
  constexpr int N = 1 << LOG2N;
  constexpr size_t lds_fwd8 = (LOG2ND == LOG2N && LOG2P < LOG2N && LOG2P >= 6 && LOG2P <= 13) ? 8 * size_t(fft8_lds_doubles(LOG2P)) : 0;
  constexpr size_t lds_d8 = (LOG2P == LOG2N && LOG2ND == LOG2N - 1 && LOG2ND >= 6 && LOG2ND <= 12) ? 8 * size_t(fft8_lds_doubles(LOG2ND)) : 0;
  constexpr size_t lds_bytes = std::max(std::max(lds_fwd8, lds_d8), LOG2N >= 14 ? 8 * size_t(N)
                               : (LOG2N == 12 && LOG2ND >= 11)
                                   ? std::max(8 * size_t(fft_lds_doubles_halves(12)), (LOG2P < LOG2N && LOG2P > 13) ? 8 * size_t(fft_lds_doubles(LOG2P)) : 0)
                               : (LOG2N == 13 && LOG2ND == 13)
                                   ? std::max(8 * size_t(fft_lds_doubles_halves(13)), LOG2P < LOG2N ? 8 * size_t(fft_lds_doubles(LOG2P)) : 0)
                                   : 8 * size_t(fft_lds_doubles(LOG2N)));
  static DynLdsOnce attr;
  if (hipError_t e = attr.set(reinterpret_cast<const void *>(&dft_kernel<LOG2N, LOG2P, LOG2ND, SP>), int(lds_bytes)); e != hipSuccess) return e;
  DftArgs b = a;
  b.hp = frame_pairs(in, out, a.C);
  b.npairs = pair_count(a.C, a.nchs);
  b.pps_magic = pair_magic(a.C, a.nchs);
  dim3 grid(item_grid(a.nblocks, b.npairs, b.hp)), block(N / 16);
  hipLaunchKernelGGL((dft_kernel<LOG2N, LOG2P, LOG2ND, SP>), grid, block, lds_bytes, st, in, out, b);
  return hipGetLastError();
}

#define RSMP_DFT_CASE(n, p, d)                                         \
  if (log2n == n && log2p == p && log2nd == d) {                       \
    if (kname) *kname = a.nchs > 0 ? "rsmp::dft_kernel<" #n ", " #p ", " #d ", true>" : "rsmp::dft_kernel<" #n ", " #p ", " #d ", false>"; \
    return a.nchs > 0 ? launch_dft_t<n, p, d, true>(in, out, a, st) : launch_dft_t<n, p, d, false>(in, out, a, st); \
  }
#define RSMP_DFT_SIZE(n, n1, n2) \
  RSMP_DFT_CASE(n, n, n)          \
  RSMP_DFT_CASE(n, n1, n)         \
  RSMP_DFT_CASE(n, n2, n)         \
  RSMP_DFT_CASE(n, n, n1)         \
  RSMP_DFT_CASE(n, n, n2)

bool dft_shape_supported(int log2n, int log2p, int log2nd)
{
  if (log2n < 11 || log2n > 14) return false;
  if (log2p == log2n) return log2nd >= log2n - 2 && log2nd <= log2n;
  return log2nd == log2n && log2p >= log2n - 2;
}

hipError_t launch_dft(int log2n, int log2p, int log2nd, bool src_f32, bool dst_f32, const F32View &sf, const F64View &sd,
                      const F32View &df, const F64View &dd, const DftArgs &a, hipStream_t st, const char **kname)
{
  const AnyView in = make_view(src_f32, sf, sd), out = make_view(dst_f32, df, dd);
  RSMP_DFT_SIZE(11, 10, 9)
  RSMP_DFT_SIZE(12, 11, 10)
  RSMP_DFT_SIZE(13, 12, 11)
  RSMP_DFT_SIZE(14, 13, 12)
  return hipErrorInvalidValue;
}

hipError_t launch_poly(int order, bool src_f32, bool dst_f32, const F32View &sf, const F64View &sd, const F32View &df,
                       const F64View &dd, const PolyArgs &a, hipStream_t st, const char **kname)
{
  static const char *const names[2][4] = {{"rsmp::poly_kernel<0>", "rsmp::poly_kernel<1>", "rsmp::poly_kernel<2>", "rsmp::poly_kernel<3>"},
                                          {"", "rsmp::poly_coop_kernel<1>", "rsmp::poly_coop_kernel<2>", "rsmp::poly_coop_kernel<3>"}};
  if (kname && order >= 0 && order <= 3) *kname = names[(order >= 1 && a.coop) ? 1 : 0][order];
  const AnyView in = make_view(src_f32, sf, sd), out = make_view(dst_f32, df, dd);
  const long long tiles = (a.count + a.tile - 1) / a.tile;
  dim3 grid((unsigned)tiles, a.C), block(256);
  const size_t lds_bytes = sizeof(double) * (size_t(a.win) + (order == 0 && a.tab_lds ? size_t(a.L) * a.n : 0));
  if (order >= 1 && a.shared_rows) { // rows shared by the channels of a workgroup (polyi_kernel)
    static const char *const pn[4] = {"", "rsmp::polyi_kernel<1>", "rsmp::polyi_kernel<2>", "rsmp::polyi_kernel<3>"};
    if (kname) *kname = pn[order];
    const long long ptiles = (a.count + kPolyiTile - 1) / kPolyiTile;
    // channel groups side by side: as few as fill the GPU (~4 workgroups per CU slot), the rest are walked inside the kernel
    const long long ngroups = (a.C + kPolyiCh - 1) / kPolyiCh;
    const long long gy = std::max<long long>(1, std::min<long long>(ngroups, (3072 + ptiles - 1) / ptiles));
    dim3 pgrid((unsigned)ptiles, (unsigned)gy);
    const size_t pl = sizeof(double) * (size_t(kPolyiTile) * a.n + size_t(kPolyiCh) * a.win);
    // (idempotent; the size only grows with the stage's window, so raising the limit to 150 KB once per instance is enough)
    static DynLdsOnce attr[4];
    auto set_attr = [&](const void *fn) { return attr[order].set(fn, 150 * 1024); };
    hipError_t e = hipSuccess;
    switch (order) {
      case 1: e = set_attr(reinterpret_cast<const void *>(&polyi_kernel<1>)); if (e == hipSuccess) hipLaunchKernelGGL(polyi_kernel<1>, pgrid, block, pl, st, in, out, a); break;
      case 2: e = set_attr(reinterpret_cast<const void *>(&polyi_kernel<2>)); if (e == hipSuccess) hipLaunchKernelGGL(polyi_kernel<2>, pgrid, block, pl, st, in, out, a); break;
      case 3: e = set_attr(reinterpret_cast<const void *>(&polyi_kernel<3>)); if (e == hipSuccess) hipLaunchKernelGGL(polyi_kernel<3>, pgrid, block, pl, st, in, out, a); break;
      default: return hipErrorInvalidValue;
    }
    return e != hipSuccess ? e : hipGetLastError();
  }
  if (order >= 1 && a.coop) {
    switch (order) {
      case 1: hipLaunchKernelGGL(poly_coop_kernel<1>, grid, block, lds_bytes, st, in, out, a); break;
      case 2: hipLaunchKernelGGL(poly_coop_kernel<2>, grid, block, lds_bytes, st, in, out, a); break;
      case 3: hipLaunchKernelGGL(poly_coop_kernel<3>, grid, block, lds_bytes, st, in, out, a); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (order) {
    case 0: hipLaunchKernelGGL(poly_kernel<0>, grid, block, lds_bytes, st, in, out, a); break;
    case 1: hipLaunchKernelGGL(poly_kernel<1>, grid, block, lds_bytes, st, in, out, a); break;
    case 2: hipLaunchKernelGGL(poly_kernel<2>, grid, block, lds_bytes, st, in, out, a); break;
    case 3: hipLaunchKernelGGL(poly_kernel<3>, grid, block, lds_bytes, st, in, out, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// whole float frames [a0, a1) of every stream, where both sides hold them contiguously: 16 bytes per thread when everything
// is 16-byte aligned (the element-wise copy_kernel moves a push's 8 MB carry of the 44.1k->192k chain in 28 us, this in ~4)
__global__ __launch_bounds__(256) void copy_frames_kernel(const float *src, long long src_stride, float *dst, long long dst_stride,
                                                          long long nfloats, int vec4)
{
  const float *s = src + (long long)blockIdx.y * src_stride;
  float *d = dst + (long long)blockIdx.y * dst_stride;
  const long long step = (long long)gridDim.x * 256;
  if (vec4) {
    const long long n4 = nfloats >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += step)
      reinterpret_cast<float4 *>(d)[i] = reinterpret_cast<const float4 *>(s)[i];
    for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < nfloats; i += step) d[i] = s[i];
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nfloats; i += step) d[i] = s[i];
  }
}

// base pointer of frames [a0, a1) of stream 0 when every stream holds them contiguously in one buffer of the view, else null
static const float *frames_base(const F32View &v, long long a0, long long a1, long long &stream_stride)
{
  if (v.ext && a0 >= v.ext_begin && a1 <= v.ext_end) {
    stream_stride = v.ext_stream_stride;
    return v.ext + (a0 - v.ext_begin) * v.nch;
  }
  if ((!v.ext || a1 <= v.ext_begin || a0 >= v.ext_end) && a0 >= 0 && (a0 & v.ring_mask) + (a1 - a0) <= v.ring_mask + 1) {
    stream_stride = v.ring_stream_stride;
    return v.ring + (a0 & v.ring_mask) * v.nch;
  }
  return nullptr;
}

static hipError_t copy_range(bool f32, const F32View &sf, const F64View &sd, const F32View &df, const F64View &dd, long long a0,
                             long long a1, int C, hipStream_t st, int depth)
{
  if (a1 <= a0) return hipSuccess;
  if (f32 && depth < 3 && sf.nch == df.nch && sf.nch > 0 && C % sf.nch == 0) { // (at most 8 pieces, then element-wise)
    long long ss = 0, ds = 0;
    const float *sp = frames_base(sf, a0, a1, ss);
    float *dp = const_cast<float *>(frames_base(df, a0, a1, ds));
    if (sp && dp) {
      const long long nfloats = (a1 - a0) * sf.nch;
      const int vec4 = ((reinterpret_cast<unsigned long long>(sp) | reinterpret_cast<unsigned long long>(dp) |
                         (unsigned long long)(ss * 4) | (unsigned long long)(ds * 4)) & 15) == 0;
      const long long blocks = std::min<long long>(std::max<long long>(1, (nfloats / (vec4 ? 4 : 1) + 255) / 256), 1024);
      hipLaunchKernelGGL(copy_frames_kernel, dim3((unsigned)blocks, C / sf.nch), dim3(256), 0, st, sp, ss, dp, ds, nfloats, vec4);
      return hipGetLastError();
    }
    // not contiguous on one side: cut the range where a view changes buffers (its external buffer's ends, a ring wrap)
    // and copy the pieces (a ring's capacity is just above what it must hold, so the carry of a push wraps every few pushes)
    long long cut = a1;
    for (const F32View *v : {&sf, &df}) {
      long long dummy = 0;
      if (frames_base(*v, a0, a1, dummy)) continue; // this side is in one piece already
      if (v->ext) {
        if (v->ext_begin > a0 && v->ext_begin < cut) cut = v->ext_begin;
        if (v->ext_end > a0 && v->ext_end < cut) cut = v->ext_end;
      }
      if (a0 >= 0) {
        const long long wrap = (a0 | v->ring_mask) + 1; // first index behind a0 that maps to ring slot 0
        if (wrap < cut) cut = wrap;
      }
    }
    if (cut < a1) {
      const hipError_t e = copy_range(f32, sf, sd, df, dd, a0, cut, C, st, depth + 1);
      return e != hipSuccess ? e : copy_range(f32, sf, sd, df, dd, cut, a1, C, st, depth + 1);
    }
  }
  const AnyView in = make_view(f32, sf, sd), out = make_view(f32, df, dd);
  long long blocks = (a1 - a0 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  dim3 grid((unsigned)blocks, C), block(256);
  hipLaunchKernelGGL(copy_kernel, grid, block, 0, st, in, out, a0, a1);
  return hipGetLastError();
}

hipError_t launch_copy(bool f32, const F32View &sf, const F64View &sd, const F32View &df, const F64View &dd, long long a0,
                       long long a1, int C, hipStream_t st)
{
  return copy_range(f32, sf, sd, df, dd, a0, a1, C, st, 0);
}

hipError_t launch_half(bool src_f32, bool dst_f32, const F32View &sf, const F64View &sd, const F32View &df,
                       const F64View &dd, const HalfArgs &a, hipStream_t st, const char **kname)
{
  static const char *const names[6] = {"rsmp::half_kernel<8>", "rsmp::half_kernel<9>", "rsmp::half_kernel<10>",
                                       "rsmp::half_kernel<11>", "rsmp::half_kernel<12>", "rsmp::half_kernel<13>"};
  if (kname && a.ncoef >= 8 && a.ncoef <= 13) *kname = names[a.ncoef - 8];
  const AnyView in = make_view(src_f32, sf, sd), out = make_view(dst_f32, df, dd);
  const long long tiles = (a.count + kHalfTile - 1) / kHalfTile;
  dim3 grid((unsigned)tiles, a.C), block(256);
  switch (a.ncoef) {
    case 8: hipLaunchKernelGGL(half_kernel<8>, grid, block, 0, st, in, out, a); break;
    case 9: hipLaunchKernelGGL(half_kernel<9>, grid, block, 0, st, in, out, a); break;
    case 10: hipLaunchKernelGGL(half_kernel<10>, grid, block, 0, st, in, out, a); break;
    case 11: hipLaunchKernelGGL(half_kernel<11>, grid, block, 0, st, in, out, a); break;
    case 12: hipLaunchKernelGGL(half_kernel<12>, grid, block, 0, st, in, out, a); break;
    case 13: hipLaunchKernelGGL(half_kernel<13>, grid, block, 0, st, in, out, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

} // namespace rsmp
