// Rational polyphase stage (vpoly0, rate/rate_filters_generic.h:272-305) on the fp64 matrix pipe, as a stage
// of its own: used where the stage cannot ride inside fused_kernel (its producer is an FFT-FIR stage with
// N >= 8192, or another kind of stage).  Same formulation as the polyphase part of fused.hip -- 4 consecutive
// output residues x 4 taps times 4 taps x 4 periods per v_mfma_f64_4x4x4 block, A operands (coefficients) from
// the pre-arranged table, B operands one ds_read_b128 of (channel A, channel B) sample pairs -- but the samples
// come from the stage's input fifo: a workgroup stages Vt + n + 3 of them for one channel pair in LDS and
// produces every output whose window STARTS in its tile [B*Vt, (B+1)*Vt), so there are no seams.
// Tiles are fixed in absolute sample coordinates, so results do not depend on how a stream was pushed.
#include "fifo_device.hpp"
#include "kernels.hpp"

#include <type_traits>

namespace rsmp {

constexpr int kPmPad = 32; // zeroed guard samples in front of and behind the staged window

template <int KS> __global__ __launch_bounds__(256, KS <= 8 ? 4 : 3) void polymf_kernel(AnyView in, AnyView out, PolyMfArgs a)
{
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double2 *smp = reinterpret_cast<double2 *>(lds) + kPmPad; // smp[n] = sample b0 + n of (channel A, channel B)
  const int tid = threadIdx.x;
  const int npairs = (a.C + 1) >> 1;
  const int w = blockIdx.x;
  const int bl = w / npairs, pair = w - bl * npairs;
  const int ca = 2 * pair, cb = ca + 1;
  const bool hasb = cb < a.C;
  const FusedBlock fb = a.blk[bl];
  if (fb.cnt <= 0) return; // uniform
  const long long b0 = (a.B0 + bl) * (long long)a.Vt;
  const int pl = a.polyL, step = a.step, at0 = (int)a.at0;
  const int W = a.Vt + a.n + 4; // samples any stored output of this tile can touch

  { // stage the window; positions the producer has not written yet read as zero (only unstored outputs see them)
    const PairSpan sp = b0 + W <= a.in_limit ? pair_span(in, pair, hasb, b0, W) : PairSpan{0, nullptr, 1, nullptr, nullptr, hasb};
    if (sp.kind) {
      for (int i = tid; i < W; i += 256) {
        double x, y;
        sp.get(i, x, y);
        smp[i] = make_double2(x, y);
      }
    } else {
      const ChanRef ia = chan_ref(in, ca), ib = chan_ref(in, hasb ? cb : ca);
      for (int i = tid; i < W; i += 256) {
        const long long e = b0 + i;
        const bool have = e < a.in_limit;
        smp[i] = make_double2(have ? fifo_get(ia, e) : 0.0, have && hasb ? fifo_get(ib, e) : 0.0);
      }
    }
    if (tid < kPmPad) {
      smp[tid - kPmPad] = make_double2(0.0, 0.0);
      smp[W + tid] = make_double2(0.0, 0.0);
    }
  }
  __syncthreads();

  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 4, bq = (lane >> 2) & 3, jq = lane & 3;
  const int irel_hi = fb.irel_lo + fb.cnt;

  // float32 destination with both channels of the pair side by side in a frame: one 8-byte store per lane
  bool ofast = false;
  float *obase = nullptr;
  int ofs = 2;
  {
    const long long o0 = a.out_offset + fb.i_lo, o1 = o0 + fb.cnt;
    if (out.is_f32 && hasb && !(out.f.nch & 1)) {
      const int hp = out.f.nch >> 1, strm = pair / hp, pin = pair - strm * hp;
      ofs = out.f.nch;
      if (out.f.ext && o0 >= out.f.ext_begin && o1 <= out.f.ext_end) {
        obase = out.f.ext + strm * out.f.ext_stream_stride + (o0 - out.f.ext_begin) * out.f.nch + 2 * pin;
        ofast = true;
      } else if ((!out.f.ext || o0 >= out.f.ext_end || o1 <= out.f.ext_begin) &&
                 (o0 & out.f.ring_mask) + (o1 - o0) <= out.f.ring_mask + 1) {
        obase = out.f.ring + strm * out.f.ring_stream_stride + (o0 & out.f.ring_mask) * out.f.nch + 2 * pin;
        ofast = true;
      }
      ofast = ofast && (reinterpret_cast<unsigned long long>(obase) & 7) == 0;
    }
  }

  auto run = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const ChanRef oa = chan_ref(out, ca), ob = chan_ref(out, hasb ? cb : ca);
    char *const obytes = reinterpret_cast<char *>(obase);
    const int frame_bytes = ofs * 4, period4_bytes = 4 * pl * frame_bytes;
    constexpr int NW = 4, MAXCS = 4;
    const int li_lo = -kPmPad, li_hi = W + kPmPad - 4 * KS;
    const int ncs = (fb.K + 3) >> 2, half0 = (ncs + 1) >> 1;

    double cn_[KS];
    {
      const double *cp = a.cfm + (wave >> 1) * (KS * 64);
#pragma unroll
      for (int s = 0; s < KS; ++s) cn_[s] = cp[s * 64 + lane];
    }
    // stores of an item are issued at the start of the next one (see fused.hip: one in-order vmcnt)
    double pA[MAXCS], pB[MAXCS];
    int pend_n = 0, pend_ib = 0, pend_hi = 0, pend_allv = 0, pend_off = 0;
    auto flush = [&]() {
#pragma unroll
      for (int u = 0; u < MAXCS; ++u) {
        if (u < pend_n) {
          const int ib = pend_ib + u * 4 * pl;
          if (((pend_allv >> u) & 1) || (ib >= fb.irel_lo && ib < pend_hi)) {
            const int orel = ib - fb.irel_lo;
            if (FAST) {
              *reinterpret_cast<float2 *>(obytes + (pend_off + u * period4_bytes)) = make_float2((float)pA[u], (float)pB[u]);
            } else {
              const long long oabs = a.out_offset + fb.i_lo + orel;
              fifo_put(oa, oabs, pA[u]);
              if (hasb) fifo_put(ob, oabs, pB[u]);
            }
          }
        }
      }
      pend_n = 0;
    };
    for (int it = wave; it < 2 * a.NGRP; it += NW) { // (16-residue group, half of the column steps)
      const int g = it >> 1, second = (it + (it >> 2)) & 1;
      int cs0 = second ? half0 : 0, cs1 = second ? ncs : half0;
      while (cs0 < cs1 && (4 * cs0 + 3) * pl + 16 * g + 15 < fb.irel_lo) ++cs0;
      while (cs1 > cs0 && 4 * (cs1 - 1) * pl + 16 * g >= irel_hi) --cs1;
      double ca_[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) ca_[s] = cn_[s];
      flush();
      {
        const int nx = it + NW < 2 * a.NGRP ? it + NW : wave;
        const double *cp = a.cfm + (nx >> 1) * (KS * 64);
#pragma unroll
        for (int s = 0; s < KS; ++s) cn_[s] = cp[s * 64 + lane];
      }
      int rb = 16 * g + 4 * bq;
      if (rb >= pl) rb = 0;
      const int qb = (at0 + rb * step) / pl + fb.base_li + hi + jq * step;
      const int step4 = 4 * step;
      double2 x0[KS], x1[KS];
      auto fill = [&](double2 (&x)[KS], int cs) {
        const int li = max(li_lo, min(li_hi, qb + cs * step4));
        const double2 *xp = smp + li;
#pragma unroll
        for (int s = 0; s < KS; ++s) x[s] = xp[4 * s];
      };
      auto column_step = [&](const double2 (&x)[KS], double &accA, double &accB) {
        accA = 0.0;
        accB = 0.0;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          accA = __builtin_amdgcn_mfma_f64_4x4x4f64(ca_[s], x[s].x, accA, 0, 0, 0);
          accB = __builtin_amdgcn_mfma_f64_4x4x4f64(ca_[s], x[s].y, accB, 0, 0, 0);
        }
      };
      if (cs0 < cs1) fill(x0, cs0);
#pragma unroll
      for (int u = 0; u < MAXCS; ++u) {
        if (cs0 + u < cs1) {
          if (cs0 + u + 1 < cs1) fill((u & 1) ? x0 : x1, cs0 + u + 1);
          column_step((u & 1) ? x1 : x0, pA[u], pB[u]);
        }
      }
      const int rD = 16 * g + 4 * bq + hi, k0 = 4 * cs0;
      pend_n = cs1 - cs0;
      pend_ib = (k0 + jq) * pl + rD;
      pend_off = (pend_ib - fb.irel_lo) * frame_bytes;
      pend_hi = rD < pl ? min(irel_hi, fb.K * pl) : -1;
      pend_allv = 0;
      if (16 * g + 15 < pl) {
#pragma unroll
        for (int u = 0; u < MAXCS; ++u)
          if (k0 + 4 * u + 3 < fb.K && (k0 + 4 * u) * pl + 16 * g >= fb.irel_lo && (k0 + 4 * u + 3) * pl + 16 * g + 15 < irel_hi)
            pend_allv |= 1 << u;
      }
    }
    flush();
  };
  if (ofast) run(std::true_type{});
  else run(std::false_type{});
}

bool polymf_supported(int ksteps) { return ksteps >= 7 && ksteps <= 9; }

hipError_t launch_polymf(int ksteps, bool src_f32, bool dst_f32, const F32View &sf, const F64View &sd, const F32View &df,
                         const F64View &dd, const PolyMfArgs &a, hipStream_t st, const char **kname)
{
  if (kname) *kname = ksteps == 7 ? "rsmp::polymf_kernel<7>" : ksteps == 8 ? "rsmp::polymf_kernel<8>" : "rsmp::polymf_kernel<9>";
  const AnyView in = make_view(src_f32, sf, sd), out = make_view(dst_f32, df, dd);
  const size_t lds_bytes = size_t(kPmPad + a.Vt + a.n + 4 + kPmPad) * 16;
  dim3 grid(a.nblocks * ((a.C + 1) / 2)), block(256);
  if (ksteps == 7) hipLaunchKernelGGL(polymf_kernel<7>, grid, block, lds_bytes, st, in, out, a);
  else if (ksteps == 8) hipLaunchKernelGGL(polymf_kernel<8>, grid, block, lds_bytes, st, in, out, a);
  else if (ksteps == 9) hipLaunchKernelGGL(polymf_kernel<9>, grid, block, lds_bytes, st, in, out, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

} // namespace rsmp
