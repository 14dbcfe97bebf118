// Stage planner. See plan.hpp.  Decision logic follows rate/rate_base.h:247-423 of the reference
// (what chain it builds for a given ratio / quality); the code is organised around a RatioSplit
// value instead of the reference's macro-driven stage indexing.
#include "plan.hpp"

#include "design.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <sstream>

namespace rsmp {

namespace {

const double kTwo32 = 4294967296.0;

double to_dB(double x) { return std::log10(x) * 20; }
double to_3dB(double a) { return (1.6e-6 * a - 7.5e-4) * a + .646; } // TO_3dB, rate_base.h:240
bool pow2_ge2(int x) { return x >= 2 && !(x & (x - 1)); }            // sox_i.h:28

// ---- half-band tables: literal constants of rate/rate_filters_generic.h:31-70 ----
const double kHb8[] = {0.3115465451887802, -0.08734497241282892, 0.03681452335604365, -0.01518925831569441,
                       0.005454118437408876, -0.001564400922162005, 0.0003181701445034203, -3.48001341225749e-5};
const double kHb9[] = {0.3122703613711853, -0.08922155288172305, 0.03913974805854332, -0.01725059723447163,
                       0.006858970092378141, -0.002304518467568703, 0.0006096426006051062, -0.0001132393923815236,
                       1.119795386287666e-5};
const double kHb10[] = {0.3128545521327376, -0.09075671986104322, 0.04109637155154835, -0.01906629512749895,
                        0.008184039342054333, -0.0030766775017262, 0.0009639607022414314, -0.0002358552746579827,
                        4.025184282444155e-5, -3.629779111541012e-6};
const double kHb11[] = {0.3133358837508807, -0.09203588680609488, 0.04276515428384758, -0.02067356614745591,
                        0.00942253142371517, -0.003856330993895144, 0.001363470684892284, -0.0003987400965541919,
                        9.058629923971627e-5, -1.428553070915318e-5, 1.183455238783835e-6};
const double kHb12[] = {0.3137392991811407, -0.0931182192961332, 0.0442050575271454, -0.02210391200618091,
                        0.01057473015666001, -0.00462766983973885, 0.001793630226239453, -0.0005961819959665878,
                        0.0001631475979359577, -3.45557865639653e-5, 5.06188341942088e-6, -3.877010943315563e-7};
const double kHb13[] = {0.3140822554324578, -0.0940458550886253, 0.04545990399121566, -0.02338339450796002,
                        0.01164429409071052, -0.005380686021429845, 0.002242915773871009, -0.000822047600000082,
                        0.0002572510962395222, -6.607320708956279e-5, 1.309926399120154e-5, -1.790719575255006e-6,
                        1.27504961098836e-7};
struct HalfChoice { int n; const double *c; float att; };
// attenuation each half-band achieves; float in the reference (rate_filters_generic.h:255-262)
const HalfChoice kHalf[] = {{8, kHb8, 136.51f}, {9, kHb9, 152.32f}, {10, kHb10, 168.07f},
                            {11, kHb11, 183.78f}, {12, kHb12, 199.44f}, {13, kHb13, 212.75f}};

// Phase-bit budgets for the interpolated polyphase variants (rate_filters_generic.h:724-746):
// bits[0], bits[1] belong to the 1st / 2nd interpolated alternative; order = interpolation kernel
// (vpoly1/2/3).  Rows 12-13 are the fixed U100 filters, only reachable for mode <= 1.
struct InterpRow { float bits1; int fn1; float bits2; int fn2; };
const InterpRow kInterp[19] = {
    {7.2f, 1, 5.0f, 2},  {9.4f, 1, 6.7f, 2},  {12.4f, 1, 7.8f, 2}, {13.6f, 1, 9.3f, 2}, {10.5f, 2, 8.4f, 3}, {11.85f, 2, 9.0f, 3},
    {8.0f, 1, 5.3f, 2},  {8.6f, 1, 5.7f, 2},  {10.6f, 1, 6.75f, 2}, {12.6f, 1, 8.6f, 2}, {9.6f, 2, 7.6f, 3},  {11.4f, 2, 8.65f, 3},
    {0, 0, 0, 0},        {0, 0, 0, 0},
    {9, 1, 6, 2},        {11, 1, 7, 2},       {13, 1, 8, 2},       {10, 2, 8, 3},       {12, 2, 9, 3}};

struct RatioSplit {
  int shift = 0;            // number of leading half-band /2 stages
  int preL = 1, preM = 1;   // DFT stage ahead of the arbitrary-ratio stage
  double arbM = 1;          // arbitrary stage: step (numerator when rational)
  int arbL = 1;             //                   phases (denominator when rational)
  int postL = 1, postM = 1; // DFT stage after it
  bool upsample = false, rational = false;
  int mode = 0;
  bool has_pre() const { return preM * preL != 1; }
  bool has_arb() const { return arbM * arbL != 1; }
  bool has_post() const { return postM * postL != 1; }
};

// rate_base.h:283-310
RatioSplit split_ratio(double factor, int mode, int interpolator, int max_coefs_size, bool small_int_opt)
{
  RatioSplit r;
  r.mode = mode;
  r.arbM = factor;
  bool again = true;
  while (again) {
    again = false;
    const int maxL = interpolator > 0 ? 1 : r.mode ? 2048 : int(std::ceil(max_coefs_size * 1000. / (44 * sizeof(double))));
    double eps = 0;
    r.upsample = r.arbM < 1;
    r.shift = 0;
    for (int i = int(r.arbM * .5); i >>= 1;) {
      r.arbM *= .5;
      ++r.shift;
    }
    r.preM = r.upsample || (r.arbM > 1.5 && r.arbM < 2);
    r.postM = 1 + (r.arbM > 1 && r.preM);
    r.arbM /= r.postM;
    r.preL = 1 + (!r.preM && r.arbM < 2) + (r.upsample && r.mode);
    r.arbM *= r.preL;
    const double frac = r.arbM - int(r.arbM);
    if (frac != 0) eps = std::fabs(std::floor(frac * kTwo32 + .5) / (frac * kTwo32) - 1);
    r.rational = frac == 0;
    for (int i = 1; i <= maxL && !r.rational; ++i) {
      const double d = frac * i;
      const int near = int(d + .5);
      r.rational = std::fabs(near / d - 1) <= eps;
      if (r.rational) {
        if (near == i) { // the ratio is an integer after all
          r.arbM = std::ceil(r.arbM);
          const int extra = r.arbM > 3;
          r.shift += extra;
          r.arbM /= 1 + extra;
        } else {
          r.arbM = i * int(r.arbM) + near;
          r.arbL = i;
        }
      }
    }
    int L = r.preL * r.arbL, M = int(r.arbM * r.postM);
    const int odd = (L | M) & 1;
    L >>= !odd;
    M >>= !odd;
    double d;
    if (small_int_opt && r.postL == 1 && (d = r.preL * r.arbL / r.arbM) > 4 && d != 5) {
      r.postL = 4;
      for (int i = int(d / 16); i >>= 1;) r.postL <<= 1;
      r.arbM = r.arbM * r.postL / r.arbL / r.preL;
      r.arbL = 1;
      again = true;
    } else if (r.rational && (std::max(L, M) < 3 + 2 * small_int_opt || L * M < 6 * small_int_opt)) {
      r.preL = L;
      r.preM = M;
      r.arbM = 1;
      r.arbL = r.postM = 1;
    }
    if (!r.mode && (!r.rational || again)) {
      ++r.mode;
      again = true;
    }
  }
  return r;
}

// rate/prepare_coefs.h:20-46, generic layout [phase][tap][order+1], highest power first
std::vector<double> polyphase_table(const std::vector<double> &taps, int n, int phases, int order)
{
  const int o1 = order + 1;
  std::vector<double> tab(size_t(n) * phases * o1, 0.0);
  double fm1 = taps[0], f1 = 0, f2 = 0;
  for (int i = n - 1; i >= 0; --i)
    for (int j = phases - 1; j >= 0; --j) {
      const double f0 = fm1;
      const int pos = i * phases + j - 1;
      fm1 = pos > 0 ? taps[pos - 1] : 0;
      double b = 0, c = 0, d = 0;
      if (order == 1) b = f1 - f0;
      else if (order == 2) { c = .5 * (f2 + f0) - f1; b = f1 - c - f0; }
      else if (order == 3) { c = .5 * (f1 + fm1) - f0; d = (1 / 6.) * (f2 - f1 + fm1 - f0 - 4 * c); b = f1 - f0 - d - c; }
      double *slot = &tab[(size_t(j) * n + (n - 1 - i)) * o1];
      slot[order] = f0;
      if (order > 0) slot[order - 1] = b;
      if (order > 1) slot[order - 2] = c;
      if (order > 2) slot[order - 3] = d;
      f2 = f1;
      f1 = f0;
    }
  return tab;
}

// rate_base.h:156-192
void plan_dft_stage(ChainPlan &plan, int which, double Fp, double Fs, double Fn, double att, double phase,
                    StageSpec &st, int L, int M)
{
  DftFilter &f = plan.dft[which];
  if (!f.num_taps) {
    int num_taps = 0;
    const int k = phase == 50 && pow2_ge2(L) && Fn == L ? L << 1 : 4;
    f.taps = design_lowpass(Fp, Fs, Fn, att, num_taps, -k);
    plan.trace.push_back({Fp, Fs, Fn, att, -k, num_taps});
    if (phase != 50) {
      to_phase(f.taps, f.post_peak, phase);
      num_taps = int(f.taps.size());
    } else
      f.post_peak = num_taps / 2;
    f.num_taps = num_taps;
    f.N = dft_block_length(num_taps);
  }
  st.kind = StageKind::Dft;
  st.filt = which;
  st.preload = f.post_peak / L;
  st.remL0 = f.post_peak % L;
  st.L = L;
  st.step = (std::abs(3 - M) == 1 && Fs == 1) ? -M / 2 : M;
}

} // namespace

const double *half_band_coefs(int n)
{
  for (const auto &h : kHalf)
    if (h.n == n) return h.c;
  return nullptr;
}

int make_plan(const Config &cfg, ChainPlan &plan)
{
  plan = ChainPlan();
  plan.cfg = cfg;
  if (!cfg.in_rate || !cfg.out_rate) return 6;
  const double factor = double(cfg.in_rate) / double(cfg.out_rate);
  if (factor > 5644.8 || factor < 1.0 / 5644.8) return 6; // rate_base.h:528
  if (!(cfg.phase >= 0 && cfg.phase <= 100)) return 6;     // asserted at rate_base.h:278

  // convert_settings, rate_base.h:674-704
  const bool best = cfg.quality == 0;
  const int rolloff = best ? 0 : 1; // rolloff_none / rolloff_small
  const double bits = 16 + 4 * std::max((best ? 6 : 4) - 3, 0);
  const double aa_pc = cfg.allow_aliasing ? cfg.bandwidth : 100;
  const double bw_pc = 100 - (100 - cfg.bandwidth) / to_3dB(bits * to_dB(2.));
  if (!(bw_pc >= 53 && bw_pc <= 100) || !(aa_pc >= 85 && aa_pc <= 100)) return 6; // rate_base.h:279-280
  const int interpolator = -1, max_coefs_size = 400;
  const bool small_int_opt = true, maintain_3dB = true;

  plan.factor = factor;
  plan.isamp_max = 1048576; // rate_base.h:531
  if (factor < 1) plan.isamp_max = size_t(double(plan.isamp_max) * factor);

  double att = (bits + 1) * to_dB(2.), att_arb = att;
  const double tbw0 = 1 - bw_pc / 100, Fs_a = 2 - aa_pc / 100;
  int mode = rolloff > 1 ? (factor > 1 || bw_pc > (67 + 5 / 8.)) : int(std::ceil(2 + (bits - 17) / 4));

  RatioSplit r = split_ratio(factor, mode, interpolator, max_coefs_size, small_int_opt);
  mode = r.mode;

  int remaining = r.shift + r.has_pre() + r.has_arb() + r.has_post();
  plan.stages.assign(size_t(remaining), StageSpec());
  if (remaining > 1) { // attenuation budget, rate_base.h:317-321
    if (r.has_arb()) {
      att += to_dB(2.);
      att_arb = att;
      --remaining;
    }
    att += to_dB(double(remaining));
  }

  int pick = 0;
  while (pick + 1 < 6 && att > kHalf[pick].att) ++pick;
  size_t si = 0;
  for (int i = 0; i < r.shift; ++i, ++si) { // rate_base.h:324-328
    StageSpec &s = plan.stages[si];
    s.kind = StageKind::Half;
    s.hb_n = kHalf[pick].n;
    s.hb = kHalf[pick].c;
    s.pre_post = 4 * s.hb_n;
    s.preload = s.pre = s.pre_post >> 1;
  }

  double tighten = 1;
  if (r.has_pre()) { // rate_base.h:330-341
    if (maintain_3dB && r.has_post()) {
      const double tbw3 = tbw0 * to_3dB(att);
      double x = ((2.1429e-4 - 5.2083e-7 * att) * att - .015863) * att + 3.95;
      x = att * std::pow((tbw0 - tbw3) / (r.postM / (factor * r.postL) - 1 + tbw0), x);
      if (x > .035) tighten = ((4.3074e-3 - 3.9121e-4 * x) * x - .040009) * x + 1.0014;
    }
    plan_dft_stage(plan, 0, 1 - tbw0 * tighten, Fs_a, r.preM ? double(std::max(r.preL, r.preM)) : r.arbM / r.arbL, att,
                   cfg.phase, plan.stages[si], r.preL, std::max(r.preM, 1));
    ++si;
  }

  if (r.has_arb()) { // rate_base.h:350-410
    StageSpec &a = plan.stages[si++];
    const int row = 6 * (int(r.upsample) + !!r.preM) + mode - !r.upsample;
    if (row < 0 || row >= 19 || row == 12 || row == 13) return 2; // not reachable through RR_config
    const double mult = r.upsample ? 1 : r.arbL / r.arbM;
    double x = .5;
    double Fn = 1;
    if (!r.upsample && r.preM) Fn = x = r.arbM / r.arbL;
    double Fp = !r.preM ? mult : mode ? .5 : 1;
    const double Fs = 2 - Fp;
    Fp *= 1 - tbw0;
    if (rolloff > 1 && mode) Fp = !r.preM ? mult * .5 - .125 : mult * .05 + .1;
    else if (rolloff == 1) Fp = Fs - (Fs - .148 * x - Fp * .852) * (.00813 * bits + .973);

    int alt = (interpolator < 0 ? !r.rational : std::max(interpolator, int(!r.rational))) - 1;
    int order = 0, num_coefs = 0, phase_bits = 0, phases = 0;
    double at = 0;
    for (;;) {
      ++alt;
      const double budget = alt == 0 ? 0.0 : alt == 1 ? double(kInterp[row].bits1) : double(kInterp[row].bits2);
      if (alt) { // interpolated variants run on the real-valued ratio
        r.arbM /= r.arbL;
        r.arbL = 1;
        r.rational = false;
      }
      phase_bits = int(std::ceil(budget + std::log(mult) / std::log(2.)));
      phases = !r.rational ? (1 << phase_bits) : r.arbL;
      { // taps per phase from a sizing run
        const int phases0 = std::max(phases, 19);
        int n0 = 0;
        design_lowpass(Fp, Fs, -Fn, att_arb, n0, phases0);
        num_coefs = n0 / phases0 + 1;
        num_coefs += num_coefs & !r.preM;
      }
      if ((num_coefs & 1) && r.rational && (r.arbL & 1)) {
        phases <<= 1;
        r.arbL <<= 1;
        r.arbM *= 2;
      }
      at = r.arbL * .5 * (num_coefs & 1);
      order = alt + (alt && mode > 4);
      const int bytes = num_coefs * phases * (order + 1) * int(sizeof(double));
      const bool has_next = alt == 0 ? kInterp[row].fn1 != 0 : alt == 1 ? kInterp[row].fn2 != 0 : false;
      if (!(interpolator < 0 && alt < 2 && has_next && bytes / 1000 > max_coefs_size)) break;
    }
    if (plan.poly_table.empty()) {
      int num_taps = num_coefs * phases - 1;
      std::vector<double> proto = design_lowpass(Fp, Fs, Fn, att_arb, num_taps, phases);
      plan.trace.push_back({Fp, Fs, Fn, att_arb, phases, num_taps});
      plan.poly_table = polyphase_table(proto, num_coefs, phases, order);
    }
    a.kind = StageKind::Poly;
    a.order = order;
    a.pre_post = num_coefs - 1;
    a.preload = (num_coefs - 1) >> 1;
    a.n = num_coefs;
    a.phase_bits = phase_bits;
    a.L = r.arbL;
    a.at0 = int64_t(at * kTwo32 + .5);
    a.step64 = int64_t(r.arbM * kTwo32 + .5);
    a.out_in_ratio = kTwo32 * r.arbL / double(a.step64);
  }

  if (r.has_post()) { // rate_base.h:412-415
    const double Fp = 1 - (1 - (1 - tbw0) * (r.upsample ? factor * r.postL / r.postM : 1)) * tighten;
    plan_dft_stage(plan, 1, Fp, Fs_a, double(std::max(r.postL, r.postM)), att, cfg.phase, plan.stages[si], r.postL, r.postM);
    ++si;
  }
  return 0;
}

std::string ChainPlan::describe() const
{
  std::ostringstream o;
  char buf[256];
  o << "{\"factor\": ";
  std::snprintf(buf, sizeof buf, "%.17g", factor);
  o << buf << ", \"isamp_max\": " << isamp_max << ", \"stages\": [";
  for (size_t i = 0; i < stages.size(); ++i) {
    const StageSpec &s = stages[i];
    if (i) o << ", ";
    if (s.kind == StageKind::Dft) {
      const DftFilter &f = dft[s.filt];
      o << "{\"kind\": \"dft\", \"L\": " << s.L << ", \"step_int\": " << s.step << ", \"num_taps\": " << f.num_taps
        << ", \"dft_length\": " << f.N << ", \"post_peak\": " << f.post_peak << ", \"preload\": " << s.preload
        << ", \"remL\": " << s.remL0 << ", \"filt\": " << s.filt << "}";
    } else if (s.kind == StageKind::Poly) {
      o << "{\"kind\": \"poly\", \"L\": " << s.L << ", \"step_int\": " << (s.step64 >> 32) << ", \"step\": " << s.step64
        << ", \"at\": " << s.at0 << ", \"n\": " << s.n << ", \"interp_order\": " << s.order
        << ", \"phase_bits\": " << s.phase_bits << ", \"preload\": " << s.preload << ", \"pre_post\": " << s.pre_post << "}";
    } else {
      o << "{\"kind\": \"half\", \"n\": " << s.hb_n << ", \"pre\": " << s.pre << ", \"pre_post\": " << s.pre_post
        << ", \"preload\": " << s.preload << "}";
    }
  }
  o << "], \"design_calls\": [";
  for (size_t i = 0; i < trace.size(); ++i) {
    const DesignCall &c = trace[i];
    std::snprintf(buf, sizeof buf, "%s{\"Fp\": %.17g, \"Fs\": %.17g, \"Fn\": %.17g, \"att\": %.17g, \"k\": %d, \"num_taps\": %d}",
                  i ? ", " : "", c.Fp, c.Fs, c.Fn, c.att, c.k, c.num_taps);
    o << buf;
  }
  o << "]}";
  return o.str();
}

} // namespace rsmp
