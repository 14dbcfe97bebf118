// Host-side filter design (fp64). See design.hpp.
#include "design.hpp"

#include <algorithm>
#include <cmath>
#include <complex>

namespace rsmp {

namespace {
const double kPi = 3.14159265358979323846;

unsigned bit_reverse(unsigned v, int bits)
{
  unsigned r = 0;
  for (int b = 0; b < bits; ++b) r |= ((v >> b) & 1u) << (bits - 1 - b);
  return r;
}
} // namespace

void fft_inplace(std::vector<cplx> &a, int sign)
{
  const size_t n = a.size();
  int bits = 0;
  while ((size_t(1) << bits) < n) ++bits;
  for (size_t i = 0; i < n; ++i) {
    size_t r = bit_reverse(unsigned(i), bits);
    if (r > i) std::swap(a[i], a[r]);
  }
  std::vector<cplx> w(n / 2 ? n / 2 : 1);
  for (size_t k = 0; k < n / 2; ++k) {
    double th = 2 * kPi * double(k) / double(n);
    w[k] = cplx(std::cos(th), sign * std::sin(th));
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const size_t half = len / 2, stride = n / len;
    for (size_t base = 0; base < n; base += len)
      for (size_t k = 0; k < half; ++k) {
        cplx t = a[base + k + half] * w[k * stride];
        cplx u = a[base + k];
        a[base + k] = u + t;
        a[base + k + half] = u - t;
      }
  }
}

double bessel_i0(double x)
{
  // power series, summed until it stops changing (effects_i_dsp.c:46-55)
  double acc = 1, term = 1;
  const double hx = x / 2;
  for (int i = 1;; ++i) {
    const double y = hx / i;
    const double before = acc;
    term *= y * y;
    acc += term;
    if (acc == before) break;
  }
  return acc;
}

int dft_block_length(int num_taps)
{
  int len = 8;
  for (int n = num_taps; n > 2; n >>= 1) len <<= 1;
  if (len < 65536) len *= 2;
  return std::min(std::max(len, 2048), 131072);
}

double kaiser_beta(double att, double tr_bw)
{
  if (att >= 60) {
    // cubic fits in `att`, one row per octave of transition width (effects_i_dsp.c:85-96)
    static const double fit[10][4] = {
        {-6.784957e-10, 1.02856e-05, 0.1087556, -0.8988365 + .001},
        {-6.897885e-10, 1.027433e-05, 0.10876, -0.8994658 + .002},
        {-1.000683e-09, 1.030092e-05, 0.1087677, -0.9007898 + .003},
        {-3.654474e-10, 1.040631e-05, 0.1087085, -0.8977766 + .006},
        {8.106988e-09, 6.983091e-06, 0.1091387, -0.9172048 + .015},
        {9.519571e-09, 7.272678e-06, 0.1090068, -0.9140768 + .025},
        {-5.626821e-09, 1.342186e-05, 0.1083999, -0.9065452 + .05},
        {-9.965946e-08, 5.073548e-05, 0.1040967, -0.7672778 + .085},
        {1.604808e-07, -5.856462e-05, 0.1185998, -1.34824 + .1},
        {-1.511964e-07, 6.363034e-05, 0.1064627, -0.9876665 + .18},
    };
    const double octave = std::log(tr_bw / .0005) / std::log(2.);
    const int lo = std::min(std::max(int(octave), 0), 9);
    const int hi = std::min(std::max(1 + int(octave), 0), 9);
    auto eval = [att](const double *c) { return ((c[0] * att + c[1]) * att + c[2]) * att + c[3]; };
    const double b_lo = eval(fit[lo]), b_hi = eval(fit[hi]);
    return b_lo + (b_hi - b_lo) * (octave - int(octave));
  }
  if (att > 50) return .1102 * (att - 8.7);
  if (att > 20.96) return .58417 * std::pow(att - 20.96, .4) + .07886 * (att - 20.96);
  return 0;
}

std::vector<double> design_lowpass(double Fp, double Fs, double Fn, double att, int &num_taps, int k, double beta)
{
  const bool estimate_len = num_taps == 0;
  const int phases = std::max(k, 1), modulo = std::max(-k, 1);
  const double rho = phases == 1 ? .5 : att < 120 ? .63 : .75;

  const double fn = std::fabs(Fn);
  Fp /= fn;
  Fs /= fn;
  double tr_bw = .5 * (Fs - Fp); // 6 dB point to stop-band edge
  tr_bw /= phases;
  Fs /= phases;
  if (!(tr_bw <= .5 * Fs)) tr_bw = .5 * Fs;
  const double Fc = Fs - tr_bw;

  // lsx_kaiser_params (effects_i_dsp.c:129-135)
  if (beta < 0) beta = kaiser_beta(att, tr_bw * .5 / Fc);
  const double len_factor = att < 60 ? (att - 7.95) / (2.285 * kPi * 2)
                                     : ((.0007528358 - 1.577737e-05 * beta) * beta + .6248022) * beta + .06186902;
  if (estimate_len) {
    num_taps = int(std::ceil(len_factor / tr_bw + 1));
    if (phases > 1) {
      int per_phase = num_taps / phases + 1;
      per_phase = (per_phase + 3) & ~3;
      num_taps = per_phase * phases - 1;
    } else
      num_taps = (num_taps + modulo - 2) / modulo * modulo + 1;
  }
  if (Fn < 0) return {};

  // lsx_make_lpf (effects_i_dsp.c:110-127), scale = phases, no DC normalisation
  std::vector<double> h(num_taps);
  const int m = num_taps - 1;
  const double gain = double(phases) / bessel_i0(beta), inv_half = 1 / (.5 * m + rho);
  for (int i = 0; i <= m / 2; ++i) {
    const double z = i - .5 * m, x = z * kPi, y = z * inv_half;
    double v = x ? std::sin(Fc * x) / x : Fc;
    v *= bessel_i0(beta * std::sqrt(1 - y * y)) * gain;
    h[i] = v;
    if (m - i != i) h[m - i] = v;
  }
  return h;
}

// lsx_fir_to_phase (effects_i_dsp.c:181-278) in EXTENDED precision: the first transform in binary128 (fft_q below), everything
// behind it on x87 long double (64-bit significand), one rounding to double at the end.
//
// The construction takes the logarithm of the filter's spectrum; in the stop band (-180 dB for the Best filters) that
// amplifies the rounding of whatever FFT produced the spectrum by ~1e9, so in plain fp64 the designed taps are an accident of
// one FFT's rounding, reproducible to ~1e-7 of the peak only (round 2 measured 1.1e-6 ... 4.2e-6 relative RMS at the output
// between this library and an independent fp64 implementation of the same steps; the reference's own Ooura transform would
// give a third answer).  In the precisions used here two implementations with transforms of different structure agree to a
// few fp64 ulps of the peak tap (tests/test_host_plan.py), for 553-tap and 5000-tap filters alike.  Against the reference itself
// the tolerance stays what its fp64 Ooura arithmetic makes it: inherent ~1e-6 at the output, unpinned.
namespace {
typedef long double ld;
typedef std::complex<ld> cld;
const ld kPiL = 3.14159265358979323846264338327950288L;

// iterative radix-2 transform on long double, twiddles from a table of cosl / sinl (one entry per distinct angle)
void fft_ld(std::vector<cld> &a, int sign)
{
  const size_t n = a.size();
  int bits = 0;
  while ((size_t(1) << bits) < n) ++bits;
  for (size_t i = 0; i < n; ++i) {
    const size_t r = bit_reverse(unsigned(i), bits);
    if (r > i) std::swap(a[i], a[r]);
  }
  std::vector<cld> w(n / 2 ? n / 2 : 1);
  for (size_t k = 0; k < n / 2; ++k) {
    const ld th = 2 * kPiL * ld(k) / ld(n);
    w[k] = cld(cosl(th), sign * sinl(th));
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const size_t half = len / 2, stride = n / len;
    for (size_t base = 0; base < n; base += len)
      for (size_t k = 0; k < half; ++k) {
        const cld t = a[base + k + half] * w[k * stride], u = a[base + k];
        a[base + k] = u + t;
        a[base + k + half] = u - t;
      }
  }
}
} // namespace

// The FIRST transform of the construction (taps -> spectrum) in binary128 (__float128, software arithmetic: a one-off of
// 35 ms for the 553-tap filters, ~1 s for a 5000-tap one).  It is the only ill-conditioned step: the stop-band bins are
// 180-200 dB below the pass band, so a transform that is accurate to 1e-19 of the LARGEST bin (long double) leaves them a
// relative error of 1e-9, which the logarithm behind it turns into the error of everything that follows -- with long double
// alone two implementations agreed to 1e-10 of the peak tap for the 553-tap filters but only to 8.5e-8 for a 4981-tap one
// (192k -> 11.025k at a 99 % passband, phase 10: 1.3e-7 relative RMS at the output, over the 1e-7 bar).  With the spectrum
// exact to ~1e-33 the rest (log, two transforms, exp, one more transform) is well conditioned and stays in long double.
namespace {
typedef __float128 q128;
struct cq { q128 re, im; };
inline cq qmul(cq a, cq b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// exp(i * sign * 2 pi / n) in binary128: pi as the sum of two long doubles, sine and cosine by their power series (|x| <= 0.1)
cq unit_root_q(size_t n, int sign)
{
  const q128 pi = (q128)3.14159265358979323851280895940618620443274267017841339111328125L + (q128)(-5.016557612668332023557327080330757013833665769218359371379101e-20L);
  const q128 x = 2 * pi / (q128)(long double)n, x2 = x * x;
  q128 c = 1, s = x, tc = 1, ts = x;
  for (int k = 1; k < 20; ++k) {
    tc = -tc * x2 / (q128)((2 * k - 1) * (2 * k));
    ts = -ts * x2 / (q128)((2 * k) * (2 * k + 1));
    c += tc;
    s += ts;
  }
  return {c, sign * s};
}

// iterative radix-2 decimation in time, binary128; twiddles w^k by doubling (w^(2m) = (w^m)^2, w^(2m+1) = w^(2m) w)
void fft_q(std::vector<cq> &a, int sign)
{
  const size_t n = a.size();
  int bits = 0;
  while ((size_t(1) << bits) < n) ++bits;
  for (size_t i = 0; i < n; ++i) {
    const size_t r = bit_reverse(unsigned(i), bits);
    if (r > i) std::swap(a[i], a[r]);
  }
  std::vector<cq> w(n / 2 ? n / 2 : 1);
  w[0] = {1, 0};
  if (n >= 4) {
    const cq w1 = unit_root_q(n, sign);
    w[1] = w1;
    for (size_t k = 2; k < n / 2; ++k) w[k] = (k & 1) ? qmul(w[k - 1], w1) : qmul(w[k / 2], w[k / 2]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const size_t half = len / 2, stride = n / len;
    for (size_t base = 0; base < n; base += len)
      for (size_t k = 0; k < half; ++k) {
        const cq t = qmul(a[base + k + half], w[k * stride]), u = a[base + k];
        a[base + k] = {u.re + t.re, u.im + t.im};
        a[base + k + half] = {u.re - t.re, u.im - t.im};
      }
  }
}
} // namespace

void to_phase(std::vector<double> &h, int &post_len, double phase)
{
  const ld blend = ld((phase > 50 ? 100 - phase : phase) / 50); // 0: minimum phase ... 1: linear
  int len = int(h.size());
  int W = 32;
  for (int i = len; i > 1; i >>= 1) W <<= 1;
  const int half = W / 2;

  std::vector<cld> buf(W);
  { // taps -> spectrum in binary128 (see fft_q), handed on as long double
    std::vector<cq> sq(W, cq{0, 0});
    for (int i = 0; i < len; ++i) sq[i].re = (q128)h[i];
    fft_q(sq, +1);
    for (int i = 0; i < W; ++i) buf[i] = cld((ld)sq[i].re, (ld)sq[i].im);
  }
  buf[0] = cld(buf[0].real(), 0);       // the reference's packed real transform has no imaginary
  buf[half] = cld(buf[half].real(), 0); // part at DC / Nyquist

  // count phase wraps and take the log magnitude, bins 0..W/2 (effects_i_dsp.c:206-224)
  std::vector<ld> wraps(half + 1), logmag(half + 1);
  ld prev2 = 0, cum2 = 0, prev1 = 0, cum1 = 0;
  for (int k = 0; k <= half; ++k) {
    const ld re = buf[k].real(), im = buf[k].imag();
    ld angle = atan2l(im, re);
    ld span = 2 * kPiL, delta = angle - prev2;
    ld adj = span * ld((delta < -span * .7L) - (delta > span * .7L));
    prev2 = angle;
    cum2 += adj;
    angle += cum2;
    span = kPiL;
    delta = angle - prev1;
    adj = span * ld((delta < -span * .7L) - (delta > span * .7L));
    prev1 = angle;
    cum1 += fabsl(adj);
    wraps[k] = cum1;
    const ld mag = sqrtl(re * re + im * im);
    logmag[k] = mag != 0 ? logl(mag) : -26;
  }

  // real cepstrum, folded onto its causal half
  for (int k = 0; k <= half; ++k) buf[k] = cld(logmag[k], 0);
  for (int k = 1; k < half; ++k) buf[W - k] = cld(logmag[k], 0);
  fft_ld(buf, -1);
  std::vector<ld> cep(W);
  for (int i = 0; i < W; ++i) cep[i] = buf[i].real() / W;
  for (int i = 1; i < half; ++i) {
    cep[i] *= 2;
    cep[i + half] = 0;
  }
  for (int i = 0; i < W; ++i) buf[i] = cld(cep[i], 0);
  fft_ld(buf, +1); // real part: log magnitude, imaginary part: minimum phase

  // blend the phase toward linear and rebuild the (Hermitian) spectrum (effects_i_dsp.c:236-246)
  std::vector<cld> spec(W);
  spec[0] = cld(expl(buf[0].real()), 0);
  spec[half] = cld(expl(buf[half].real()), 0);
  for (int k = 1; k < half; ++k) {
    const ld ph = blend * (2 * ld(k)) / W * wraps[half] + (1 - blend) * (buf[k].imag() + wraps[k]) - wraps[k];
    const ld mag = expl(buf[k].real());
    spec[k] = cld(mag * cosl(ph), mag * sinl(ph));
    spec[W - k] = std::conj(spec[k]);
  }
  fft_ld(spec, -1);
  std::vector<double> imp(W);
  for (int i = 0; i < W; ++i) imp[i] = double(spec[i].real() / W); // the one rounding to fp64

  // locate the impulse peak (effects_i_dsp.c:251-260)
  int peak = 0;
  double run = 0, best = 0;
  const int search = int(double(wraps[half]) / kPi + .5);
  for (int i = 0; i <= search; ++i) {
    run += imp[i];
    if (std::fabs(run) > std::fabs(best)) {
      best = run;
      peak = i;
    }
  }
  while (peak && std::fabs(imp[peak - 1]) > std::fabs(imp[peak]) && imp[peak - 1] * imp[peak] > 0) --peak;

  const double bl = double(blend);
  int begin;
  if (bl == 0)
    begin = 0;
  else if (bl == 1)
    begin = peak - len / 2;
  else {
    begin = int((.997 - (2 - bl) * .22) * len + .5);
    int end = int((.997 + (0 - bl) * .22) * len + .5);
    begin = peak - (begin & ~3);
    end = peak + 1 + ((end + 3) & ~3);
    len = end - begin;
  }
  h.assign(len, 0.0);
  for (int i = 0; i < len; ++i) h[i] = imp[(begin + (phase > 50 ? len - 1 - i : i) + W) & (W - 1)];
  post_len = phase > 50 ? peak - begin : begin + len - (peak + 1);
}

} // namespace rsmp
