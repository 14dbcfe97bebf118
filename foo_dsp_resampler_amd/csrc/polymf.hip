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
  const int npairs = a.npairs;
  const int w = blockIdx.x;
  const int bl = w / npairs, pair = w - bl * npairs;
  const PairCh pc = pair_channels(pair, a.C, a.nchs, a.pps_magic);
  const int ca = pc.ca, cb = pc.cb;
  const bool hasb = pc.hasb;
  const FusedBlock fb = a.blk[bl];
  if (fb.cnt <= 0) return; // uniform
  const long long b0 = (a.B0 + bl) * (long long)a.Vt;
  const int pl = a.polyL, step = a.step;
  const int W = a.Vt + a.n + 4; // samples any stored output of this tile can touch

  { // stage the window; positions the producer has not written yet read as zero (only unstored outputs see them)
    const PairSpan sp = b0 + W <= a.in_limit ? pair_span(in, pair, hasb, b0, W, ca) : PairSpan{0, nullptr, 1, nullptr, nullptr, hasb};
    if (sp.kind) {
      // batches of 5 loads per thread in flight (W <= 2048 + 36 + 4: two batches): as a plain loop the compiler waits for
      // every load before the LDS store behind it -- nine memory round trips in a row, most of a workgroup's lifetime
      for (int ib = tid; ib < W; ib += 256 * 5) {
        double x[5], y[5];
        if (sp.kind == 1) {
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const float2 f = sp.p2[min(ib + 256 * j, W - 1) * sp.fstride];
            x[j] = (double)f.x;
            y[j] = (double)f.y;
          }
        } else { // (pb == pa when the pair has one channel: every load is unconditional)
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const int i = min(ib + 256 * j, W - 1);
            x[j] = sp.pa[i];
            y[j] = sp.pb[i];
          }
          if (!hasb) {
#pragma unroll
            for (int j = 0; j < 5; ++j) y[j] = 0.0;
          }
        }
#pragma unroll
        for (int j = 0; j < 5; ++j)
          if (ib + 256 * j < W) smp[ib + 256 * j] = make_double2(x[j], y[j]);
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

  // Tile walk as in fused_fast.hip: tiles (16-residue group, column step of 4 periods) in group-major order, a contiguous
  // range per wave; coefficient tile double-buffered a group ahead, window start of a group from the host table, every
  // store issued straight after its tile under a per-lane range test, next tile's samples always prefetched.
  auto run = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const ChanRef oa = chan_ref(out, ca), ob = chan_ref(out, hasb ? cb : ca);
    char *const obytes = reinterpret_cast<char *>(obase);
    const int frame_bytes = ofs * 4;
    const int li_lo = -kPmPad, li_hi = W + kPmPad - 4 * KS;
    const int ncs = (fb.K + 3) >> 2, nt = a.NGRP * ncs;
    const int t0 = (nt * wave) >> 2, t1 = (nt * (wave + 1)) >> 2;
    if (t0 >= t1) return;
    int g = t0 / ncs, c = t0 - g * ncs;
    const int g_last = (t1 - 1) / ncs;
    const int rloc = 4 * bq + hi;
    const int lane_li = fb.base_li + hi + jq * step;
    const int lane_ib = jq * pl + rloc - fb.irel_lo; // output index relative to i_lo = lane_ib + 16 g + c * 4 * pl
    const int cnt = min(irel_hi, fb.K * pl) - fb.irel_lo;
    const int step4 = 4 * step, pl4 = 4 * pl;
    const double *const cfm_lane = a.cfm + lane;
    const int *const qtab_lane = a.qtab + bq;

    double cc[KS], cn[KS];
    int qc, qn = 0;
    auto load_tile = [&](int gg, double (&c_)[KS], int &q_) {
      const double *cp = cfm_lane + gg * (KS * 64);
#pragma unroll
      for (int s = 0; s < KS; ++s) c_[s] = cp[s * 64];
      q_ = qtab_lane[gg * 4];
    };
    load_tile(g, cc, qc);
    if (g < g_last) load_tile(g + 1, cn, qn);
    double2 x0[KS], x1[KS];
    auto fill = [&](double2 (&x)[KS], int q, int cstep) {
      const int li = max(li_lo, min(li_hi, lane_li + q + cstep * step4));
      const double2 *xp = smp + li;
#pragma unroll
      for (int s = 0; s < KS; ++s) x[s] = xp[4 * s];
    };
    fill(x0, qc, c);
    int left = t1 - t0;
    auto tile = [&](const double2 (&xc)[KS], double2 (&xn)[KS]) {
      int cnext = c + 1, gnext = g;
      if (cnext == ncs) {
        cnext = 0;
        gnext = g + 1;
      }
      const bool switch_group = gnext != g && left > 1;
      fill(xn, switch_group ? qn : qc, left > 1 ? cnext : c);
      double accA = 0.0, accB = 0.0;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        accA = __builtin_amdgcn_mfma_f64_4x4x4f64(cc[s], xc[s].x, accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f64_4x4x4f64(cc[s], xc[s].y, accB, 0, 0, 0);
      }
      const int ib = lane_ib + 16 * g + c * pl4;
      if (ib >= 0 && ib < cnt && 16 * g + rloc < pl) {
        if (FAST) {
          *reinterpret_cast<float2 *>(obytes + (unsigned)(ib * frame_bytes)) = make_float2((float)accA, (float)accB);
        } else {
          const long long oabs = a.out_offset + fb.i_lo + ib;
          fifo_put(oa, oabs, accA);
          if (hasb) fifo_put(ob, oabs, accB);
        }
      }
      if (switch_group) {
#pragma unroll
        for (int s = 0; s < KS; ++s) cc[s] = cn[s];
        qc = qn;
        if (gnext < g_last) load_tile(gnext + 1, cn, qn);
      }
      g = gnext;
      c = cnext;
      --left;
    };
    while (true) {
      tile(x0, x1);
      if (left == 0) break;
      tile(x1, x0);
      if (left == 0) break;
    }
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
  PolyMfArgs b = a;
  b.npairs = pair_count(a.C, a.nchs);
  b.pps_magic = pair_magic(a.C, a.nchs);
  dim3 grid(a.nblocks * b.npairs), block(256);
  if (ksteps == 7) hipLaunchKernelGGL(polymf_kernel<7>, grid, block, lds_bytes, st, in, out, b);
  else if (ksteps == 8) hipLaunchKernelGGL(polymf_kernel<8>, grid, block, lds_bytes, st, in, out, b);
  else if (ksteps == 9) hipLaunchKernelGGL(polymf_kernel<9>, grid, block, lds_bytes, st, in, out, b);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

} // namespace rsmp
