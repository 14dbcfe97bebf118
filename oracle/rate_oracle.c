/* rate_oracle.c -- see rate_oracle.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED at the sample
 * level (pinned by SURVEY.md [probe] facts only).  Plain C99, single-threaded per handle.
 *
 * Every function cites the reference file:line (under /root/reference/) whose behaviour it
 * restates.  The FFT is NOT Ooura's fft4g: only its packing, sign and scaling conventions are
 * kept (fft-double/fft4g_dbl.c:26-62), so results agree with the reference to fp64 rounding, not
 * bit for bit.
 */
#include "rate_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define TWO32 4294967296.0 /* MULT32, rate/rate_base.h:120 */

static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }
static double dmin(double a, double b) { return a <= b ? a : b; } /* util.h:52 min() */
static int is_pow2_ge2(int x) { return !(x < 2 || (x & (x - 1))); } /* sox_i.h:28: 1 is NOT a power of 2 here */
static double to_dB(double x) { return log10(x) * 20; }              /* util.h:78 */

void orc_free(void *p) { free(p); }

/* =========================================================================================
 * FFT.  Conventions of lsx_rdft_generic (fft-double/fft4g_dbl.c:26-62), own algorithm:
 * radix-2 decimation-in-time on n/2 complex points + even/odd split.
 * ========================================================================================= */
typedef struct {
  int m;       /* complex length */
  double *cs;  /* cos,sin(2 pi k / m), k < m/2 */
  int *rev;    /* bit reversal */
  double *rcs; /* cos,sin(2 pi k / (2m)), k <= m/2 : split twiddles for the real transform */
} fft_tab;

static fft_tab g_tab[32];
static pthread_mutex_t g_tab_lock = PTHREAD_MUTEX_INITIALIZER;

static int ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }

static const fft_tab *fft_table(int m)
{
  int lg = ilog2(m), k;
  fft_tab *t = &g_tab[lg];
  if (__atomic_load_n(&t->m, __ATOMIC_ACQUIRE) == m) return t;
  pthread_mutex_lock(&g_tab_lock);
  if (t->m != m) {
    double *cs = (double *)malloc(sizeof(double) * (size_t)imax(m, 2));
    double *rcs = (double *)malloc(sizeof(double) * (size_t)(m + 2));
    int *rev = (int *)malloc(sizeof(int) * (size_t)m);
    for (k = 0; k < m / 2; ++k) {
      cs[2 * k] = cos(2 * M_PI * k / m);
      cs[2 * k + 1] = sin(2 * M_PI * k / m);
    }
    for (k = 0; k <= m / 2; ++k) {
      rcs[2 * k] = cos(M_PI * k / m);
      rcs[2 * k + 1] = sin(M_PI * k / m);
    }
    for (k = 0; k < m; ++k) {
      int r = 0, b;
      for (b = 0; b < lg; ++b) r |= ((k >> b) & 1) << (lg - 1 - b);
      rev[k] = r;
    }
    t->cs = cs; t->rcs = rcs; t->rev = rev;
    __atomic_store_n(&t->m, m, __ATOMIC_RELEASE);
  }
  pthread_mutex_unlock(&g_tab_lock);
  return t;
}

/* in-place complex transform, z[2k],z[2k+1] = re,im; Y[k] = sum z[j] exp(sign 2 pi i jk/m).
 * Decimation in time: bit reversal, one radix-2 stage when log2(m) is odd, then radix-4 stages
 * (two radix-2 levels fused: 3 twiddle products per 4 points instead of 4). */
static void cfft(int m, int sign, double *z)
{
  const fft_tab *t = fft_table(m);
  int i, k, len;
  for (i = 0; i < m; ++i) {
    int r = t->rev[i];
    if (r > i) {
      double a = z[2 * i], b = z[2 * i + 1];
      z[2 * i] = z[2 * r]; z[2 * i + 1] = z[2 * r + 1];
      z[2 * r] = a; z[2 * r + 1] = b;
    }
  }
  len = 1;
  if (ilog2(m) & 1) { /* odd number of levels: peel one radix-2 stage (twiddle 1) */
    for (i = 0; i < m; i += 2) {
      double ar = z[2 * i], ai = z[2 * i + 1], br = z[2 * i + 2], bi = z[2 * i + 3];
      z[2 * i] = ar + br; z[2 * i + 1] = ai + bi;
      z[2 * i + 2] = ar - br; z[2 * i + 3] = ai - bi;
    }
    len = 2;
  }
  for (; len < m; len <<= 2) { /* combine four sub-transforms of length len into one of 4 len */
    const int q = len, span = 4 * len, ts1 = m / (2 * len), ts2 = m / (4 * len);
    for (i = 0; i < m; i += span) {
      double *p0 = z + 2 * i, *p1 = p0 + 2 * q, *p2 = p1 + 2 * q, *p3 = p2 + 2 * q;
      for (k = 0; k < q; ++k) {
        /* level 1 (length 2q) uses w1 = W_{2q}^k on both halves, level 2 (length 4q) uses
         * w2 = W_{4q}^k and W_{4q}^{k+q} = sign*i*w2 */
        const double w1r = t->cs[2 * k * ts1], w1i = sign * t->cs[2 * k * ts1 + 1];
        const double w2r = t->cs[2 * k * ts2], w2i = sign * t->cs[2 * k * ts2 + 1];
        const double a0r = p0[2 * k], a0i = p0[2 * k + 1];
        const double b1r = p1[2 * k] * w1r - p1[2 * k + 1] * w1i, b1i = p1[2 * k] * w1i + p1[2 * k + 1] * w1r;
        const double a2r = p2[2 * k], a2i = p2[2 * k + 1];
        const double b3r = p3[2 * k] * w1r - p3[2 * k + 1] * w1i, b3i = p3[2 * k] * w1i + p3[2 * k + 1] * w1r;
        /* level 1 */
        const double e0r = a0r + b1r, e0i = a0i + b1i, e1r = a0r - b1r, e1i = a0i - b1i; /* first pair: k, k+q  */
        const double f0r = a2r + b3r, f0i = a2i + b3i, f1r = a2r - b3r, f1i = a2i - b3i; /* second pair        */
        /* level 2: out[k] = e0 + w2 f0, out[k+2q] = e0 - w2 f0, out[k+q] = e1 + (sign i w2) f1, out[k+3q] = e1 - ... */
        const double g0r = f0r * w2r - f0i * w2i, g0i = f0r * w2i + f0i * w2r;
        const double h1r = f1r * w2r - f1i * w2i, h1i = f1r * w2i + f1i * w2r;
        const double g1r = -sign * h1i, g1i = sign * h1r; /* (sign i) * h1 */
        p0[2 * k] = e0r + g0r; p0[2 * k + 1] = e0i + g0i;
        p2[2 * k] = e0r - g0r; p2[2 * k + 1] = e0i - g0i;
        p1[2 * k] = e1r + g1r; p1[2 * k + 1] = e1i + g1i;
        p3[2 * k] = e1r - g1r; p3[2 * k + 1] = e1i - g1i;
      }
    }
  }
}

void orc_rdft(int n, int isgn, double *a)
{
  int m = n >> 1, k;
  const fft_tab *t = fft_table(m);
  if (isgn >= 0) {
    double x0;
    cfft(m, +1, a);
    for (k = 1; k < m - k; ++k) {
      double zr = a[2 * k], zi = a[2 * k + 1], yr = a[2 * (m - k)], yi = a[2 * (m - k) + 1];
      double er = .5 * (zr + yr), ei = .5 * (zi - yi);   /* even-sample spectrum */
      double orr = .5 * (zi + yi), oi = -.5 * (zr - yr); /* odd-sample spectrum */
      double wr = t->rcs[2 * k], wi = t->rcs[2 * k + 1]; /* exp(+2 pi i k/n) */
      double pr = wr * orr - wi * oi, pi = wr * oi + wi * orr;
      a[2 * k] = er + pr; a[2 * k + 1] = ei + pi;                 /* X[k]   = E + w O       */
      a[2 * (m - k)] = er - pr; a[2 * (m - k) + 1] = -(ei - pi);  /* X[m-k] = conj(E - w O) */
    }
    /* k == m/2 is unchanged (w = i) */
    x0 = a[0] - a[1];
    a[0] += a[1];
    a[1] = x0;
  } else {
    double e0 = .5 * (a[0] + a[1]), o0 = .5 * (a[0] - a[1]);
    a[0] = e0; a[1] = o0;
    for (k = 1; k < m - k; ++k) {
      double xr = a[2 * k], xi = a[2 * k + 1], yr = a[2 * (m - k)], yi = a[2 * (m - k) + 1];
      double er = .5 * (xr + yr), ei = .5 * (xi - yi);
      double dr = .5 * (xr - yr), di = .5 * (xi + yi);   /* w O */
      double wr = t->rcs[2 * k], wi = -t->rcs[2 * k + 1]; /* exp(-2 pi i k/n) */
      double orr = wr * dr - wi * di, oi = wr * di + wi * dr;
      a[2 * k] = er - oi; a[2 * k + 1] = ei + orr;               /* Z[k]   = E + i O */
      a[2 * (m - k)] = er + oi; a[2 * (m - k) + 1] = -ei + orr;  /* Z[m-k] = conj(E) + i conj(O) */
    }
    cfft(m, -1, a);
  }
}

/* =========================================================================================
 * Filter design (rate/effects_i_dsp.c)
 * ========================================================================================= */
double orc_bessel_I0(double x) /* :46-55 */
{
  double term = 1, sum = 1, prev, half = x / 2;
  int i = 1;
  do {
    double y = half / i++;
    prev = sum;
    term *= y * y;
    sum += term;
  } while (sum != prev);
  return sum;
}

int orc_dft_length(int num_taps) /* :64-73 */
{
  int len = 8, n = num_taps;
  while (n > 2) { len <<= 1; n >>= 1; }
  if (len < 65536) len *= 2;
  if (len < 2048) len = 2048;
  if (len > 131072) len = 131072;
  return len;
}

double orc_kaiser_beta(double att, double tr_bw) /* :83-108 */
{
  if (att >= 60) {
    static const double poly[][4] = {
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
    const int rows = (int)(sizeof(poly) / sizeof(poly[0]));
    double realm = log(tr_bw / .0005) / log(2.);
    int r0 = imin(imax((int)realm, 0), rows - 1);
    int r1 = imin(imax(1 + (int)realm, 0), rows - 1);
    const double *c0 = poly[r0], *c1 = poly[r1];
    double b0 = ((c0[0] * att + c0[1]) * att + c0[2]) * att + c0[3];
    double b1 = ((c1[0] * att + c1[1]) * att + c1[2]) * att + c1[3];
    return b0 + (b1 - b0) * (realm - (int)realm);
  }
  if (att > 50) return .1102 * (att - 8.7);
  if (att > 20.96) return .58417 * pow(att - 20.96, .4) + .07886 * (att - 20.96);
  return 0;
}

/* :110-127 (dc_norm is always false on this path) */
static double *make_lpf(int num_taps, double Fc, double beta, double rho, double scale)
{
  int i, m = num_taps - 1;
  double *h = (double *)malloc(sizeof(double) * (size_t)num_taps);
  double mult = scale / orc_bessel_I0(beta), mult1 = 1 / (.5 * m + rho);
  for (i = 0; i <= m / 2; ++i) {
    double z = i - .5 * m, x = z * M_PI, y = z * mult1;
    h[i] = x ? sin(Fc * x) / x : Fc;
    h[i] *= orc_bessel_I0(beta * sqrt(1 - y * y)) * mult;
    if (m - i != i) h[m - i] = h[i];
  }
  return h;
}

/* :129-135 */
static void kaiser_params(double att, double Fc, double tr_bw, double *beta, int *num_taps)
{
  *beta = *beta < 0 ? orc_kaiser_beta(att, tr_bw * .5 / Fc) : *beta;
  att = att < 60 ? (att - 7.95) / (2.285 * M_PI * 2)
                 : ((.0007528358 - 1.577737e-05 * *beta) * *beta + .6248022) * *beta + .06186902;
  if (!*num_taps) *num_taps = (int)ceil(att / tr_bw + 1);
}

static double g_last_beta; /* recorded for the design trace only */

/* :137-171, CREATE_4X_NUMTAPS variant (sox_i.h:17) */
double *orc_design_lpf(double Fp, double Fs, double Fn, double att, int *num_taps, int k, double beta)
{
  int n = *num_taps, phases = imax(k, 1), modulo = imax(-k, 1);
  double tr_bw, Fc, rho = phases == 1 ? .5 : att < 120 ? .63 : .75;

  Fp /= fabs(Fn); Fs /= fabs(Fn);
  tr_bw = .5 * (Fs - Fp);
  tr_bw /= phases; Fs /= phases;
  tr_bw = dmin(tr_bw, .5 * Fs);
  Fc = Fs - tr_bw;
  kaiser_params(att, Fc, tr_bw, &beta, num_taps);
  g_last_beta = beta;
  if (!n) {
    if (phases > 1) {
      int per_phase = *num_taps / phases + 1;
      per_phase = (per_phase + 3) & ~3;
      *num_taps = per_phase * phases - 1;
    } else
      *num_taps = (*num_taps + modulo - 2) / modulo * modulo + 1;
  }
  return Fn < 0 ? NULL : make_lpf(*num_taps, Fc, beta, rho, (double)phases);
}

static double safe_log(double x) { return x ? log(x) : -26; } /* :173-179 */

/* ---- extended-precision form of the same construction (the default) --------------------------------------
 * In fp64 the function below is reproducible to ~1e-7 of the filter's peak only: it takes the log of a -180 dB stop band, which
 * amplifies the rounding of whichever FFT produced the spectrum by ~1e9, so two correct fp64 implementations (this oracle's,
 * the product's, the reference's Ooura transform) design three slightly different filters.  Here every transform, log, exp,
 * sin / cos and atan2 runs on long double (x87, 64-bit significand) and the taps are rounded to double once at the end; the
 * product's design.cpp does the same with code of its own, and the two then agree to a few ulps of the peak
 * (tests/test_host_plan.py::test_phase_tables_agree_in_extended_precision).  orc_set_phase_arith(1) selects the fp64 statement
 * that follows the reference line by line (orc_fir_to_phase_ref64): the gap between the two IS the reference's inherent
 * irreproducibility, ~1e-6 relative RMS at the output -- "parity unpinned" for phase != 50, see DESIGN.md. */
typedef long double orc_ld;
static const orc_ld ORC_PIL = 3.14159265358979323846264338327950288L;

/* In-place decimation in FREQUENCY (Gentleman-Sande butterflies, bit reversal at the end) on split re / im arrays, twiddles
 * straight from cosl / sinl; n a power of two.  (The product's design.cpp uses decimation in time from a twiddle table: the two
 * round differently, which is the point -- what they agree on does not depend on one transform's rounding.) */
static void fft_ld(orc_ld *re, orc_ld *im, int n, int sign)
{
  int len, base, k, i, j;
  for (len = n; len >= 2; len >>= 1) {
    const int half = len / 2;
    for (k = 0; k < half; ++k) {
      const orc_ld th = 2 * ORC_PIL * (orc_ld)k / (orc_ld)len, c = cosl(th), s = sign * sinl(th);
      for (base = 0; base < n; base += len) {
        const int a = base + k, b = a + half;
        const orc_ld dr = re[a] - re[b], di = im[a] - im[b];
        re[a] += re[b]; im[a] += im[b];
        re[b] = dr * c - di * s; im[b] = dr * s + di * c;
      }
    }
  }
  for (i = 0, j = 0; i < n; ++i) { /* bit reversal */
    int bit;
    if (i < j) { orc_ld t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    for (bit = n >> 1; bit && (j & bit); bit >>= 1) j ^= bit;
    j |= bit;
  }
}

/* The first transform (taps -> spectrum) in binary128: it alone is ill-conditioned (stop-band bins 180-200 dB below the
 * largest one; see the product's design.cpp, which does the same with a decimation-in-time transform and a doubled twiddle
 * table -- here: decimation in frequency, twiddles by repeated multiplication within a stage and re-seeded from the series
 * every 64 steps). */
typedef __float128 orc_q;
static void unit_root_q(int n, int k, int sign, orc_q *c_out, orc_q *s_out)
{ /* exp(i sign 2 pi k / n): argument reduced to |x| <= pi/4 by the octant symmetries, then the power series */
  const orc_q pi = (orc_q)3.14159265358979323851280895940618620443274267017841339111328125L +
                   (orc_q)(-5.016557612668332023557327080330757013833665769218359371379101e-20L);
  /* angle = 2 pi k / n with 0 <= k < n/2, n a power of two >= 8: octant o = floor(8k/n), remainder r in [0, n/8) */
  const int o = (int)((8LL * k) / n), r = k - (int)((long long)o * n / 8);
  int kk = (o & 1) ? n / 8 - r : r; /* odd octants run backwards from the next multiple of pi/4 */
  const orc_q x = 2 * pi * (orc_q)kk / (orc_q)n, x2 = x * x;
  orc_q c = 1, sn = x, tc = 1, ts = x, cc, ss;
  int t;
  for (t = 1; t < 24; ++t) {
    tc = -tc * x2 / (orc_q)((2 * t - 1) * (2 * t));
    ts = -ts * x2 / (orc_q)((2 * t) * (2 * t + 1));
    c += tc; sn += ts;
  }
  /* (cos, sin) of the full angle from (c, sn) of the reduced one: angle = o*pi/4 + x (even o) or (o+1)*pi/4 - x (odd o) */
  switch (o) {
    case 0: cc = c; ss = sn; break;           /* x */
    case 1: cc = sn; ss = c; break;           /* pi/2 - x */
    case 2: cc = -sn; ss = c; break;          /* pi/2 + x */
    default: cc = -c; ss = sn; break;         /* pi - x */
  }
  *c_out = cc; *s_out = sign * ss;
}
static void fft_q(orc_q *re, orc_q *im, int n, int sign)
{
  int len, base, k, i, j;
  for (len = n; len >= 2; len >>= 1) {
    const int half = len / 2;
    for (k = 0; k < half; ++k) {
      orc_q c, s;
      if (len >= 8) unit_root_q(len, k, sign, &c, &s);
      else if (len == 4) { c = k ? 0 : 1; s = k ? sign : 0; }
      else { c = 1; s = 0; }
      for (base = 0; base < n; base += len) {
        const int a = base + k, b = a + half;
        const orc_q dr = re[a] - re[b], di = im[a] - im[b];
        re[a] += re[b]; im[a] += im[b];
        re[b] = dr * c - di * s; im[b] = dr * s + di * c;
      }
    }
  }
  for (i = 0, j = 0; i < n; ++i) {
    int bit;
    if (i < j) { orc_q t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    for (bit = n >> 1; bit && (j & bit); bit >>= 1) j ^= bit;
    j |= bit;
  }
}

static int g_phase_ref64 = 0;
void orc_set_phase_arith(int ref64) { g_phase_ref64 = ref64; }
void orc_fir_to_phase_ref64(double **h, int *len, int *post_len, double phase);

/* effects_i_dsp.c:181-278, same steps and constants, long double inside */
void orc_fir_to_phase(double **h, int *len, int *post_len, double phase)
{
  const double phase1 = (phase > 50 ? 100 - phase : phase) / 50;
  const orc_ld p1 = phase1;
  int i, wlen, half, begin, end, peak = 0;
  orc_ld *re, *im, *wraps;
  double *imp, imp_sum = 0, peak_imp_sum = 0;
  orc_ld prev2 = 0, cum2 = 0, prev1 = 0, cum1 = 0;

  if (g_phase_ref64) { orc_fir_to_phase_ref64(h, len, post_len, phase); return; }
  for (i = *len, wlen = 2 * 2 * 8; i > 1; wlen <<= 1, i >>= 1) {}
  half = wlen / 2;
  re = (orc_ld *)calloc((size_t)wlen, sizeof(orc_ld));
  im = (orc_ld *)calloc((size_t)wlen, sizeof(orc_ld));
  wraps = (orc_ld *)malloc(sizeof(orc_ld) * (size_t)(half + 1));
  imp = (double *)malloc(sizeof(double) * (size_t)wlen);

  { /* taps -> spectrum in binary128, handed on as long double */
    orc_q *qr = (orc_q *)calloc((size_t)wlen, sizeof(orc_q)), *qi = (orc_q *)calloc((size_t)wlen, sizeof(orc_q));
    for (i = 0; i < *len; ++i) qr[i] = (orc_q)(*h)[i];
    fft_q(qr, qi, wlen, +1);
    for (i = 0; i < wlen; ++i) { re[i] = (orc_ld)qr[i]; im[i] = (orc_ld)qi[i]; }
    free(qr); free(qi);
  }
  im[0] = 0; im[half] = 0; /* the packed real transform of the reference carries no imaginary part at DC / Nyquist */

  for (i = 0; i <= half; ++i) { /* :206-224 */
    orc_ld angle = atan2l(im[i], re[i]);
    orc_ld detect = 2 * ORC_PIL;
    orc_ld delta = angle - prev2;
    orc_ld adjust = detect * (orc_ld)((delta < -detect * .7L) - (delta > detect * .7L));
    orc_ld mag;
    prev2 = angle;
    cum2 += adjust;
    angle += cum2;
    detect = ORC_PIL;
    delta = angle - prev1;
    adjust = detect * (orc_ld)((delta < -detect * .7L) - (delta > detect * .7L));
    prev1 = angle;
    cum1 += fabsl(adjust);
    wraps[i] = cum1;
    mag = sqrtl(re[i] * re[i] + im[i] * im[i]);
    re[i] = mag != 0 ? logl(mag) : -26;
    im[i] = 0;
  }
  for (i = 1; i < half; ++i) { re[wlen - i] = re[i]; im[wlen - i] = 0; } /* Hermitian image of a real, even log spectrum */
  fft_ld(re, im, wlen, -1);
  for (i = 0; i < wlen; ++i) { re[i] /= wlen; im[i] = 0; } /* real cepstrum */
  for (i = 1; i < half; ++i) { re[i] *= 2; re[i + half] = 0; } /* :231-234 */
  fft_ld(re, im, wlen, +1); /* re: log magnitude, im: minimum phase */

  { /* :236-246 */
    const orc_ld top = wraps[half];
    orc_ld m0 = expl(re[0]), mh = expl(re[half]);
    for (i = 1; i < half; ++i) {
      const orc_ld ph = p1 * (orc_ld)(2 * i) / wlen * top + (1 - p1) * (im[i] + wraps[i]) - wraps[i];
      const orc_ld mag = expl(re[i]);
      re[i] = mag * cosl(ph); im[i] = mag * sinl(ph);
      re[wlen - i] = re[i]; im[wlen - i] = -im[i];
    }
    re[0] = m0; im[0] = 0; re[half] = mh; im[half] = 0;
  }
  fft_ld(re, im, wlen, -1);
  for (i = 0; i < wlen; ++i) imp[i] = (double)(re[i] / wlen); /* the one rounding to fp64 */

  for (i = 0; i <= (int)((double)wraps[half] / M_PI + .5); ++i) { /* :251-260 */
    imp_sum += imp[i];
    if (fabs(imp_sum) > fabs(peak_imp_sum)) { peak_imp_sum = imp_sum; peak = i; }
  }
  while (peak && fabs(imp[peak - 1]) > fabs(imp[peak]) && imp[peak - 1] * imp[peak] > 0) --peak;

  if (!phase1)
    begin = 0;
  else if (phase1 == 1)
    begin = peak - *len / 2;
  else {
    begin = (int)((.997 - (2 - phase1) * .22) * *len + .5);
    end = (int)((.997 + (0 - phase1) * .22) * *len + .5);
    begin = peak - (begin & ~3);
    end = peak + 1 + ((end + 3) & ~3);
    *len = end - begin;
    *h = (double *)realloc(*h, sizeof(double) * (size_t)*len);
  }
  for (i = 0; i < *len; ++i)
    (*h)[i] = imp[(begin + (phase > 50 ? *len - 1 - i : i) + wlen) & (wlen - 1)];
  *post_len = phase > 50 ? peak - begin : begin + *len - (peak + 1);
  free(re); free(im); free(wraps); free(imp);
}

/* :181-278 (cepstral minimum-phase construction, then interpolation toward linear phase), fp64 as in the reference */
void orc_fir_to_phase_ref64(double **h, int *len, int *post_len, double phase)
{
  double *wraps, *w, phase1 = (phase > 50 ? 100 - phase : phase) / 50;
  int i, wlen, begin, end, peak = 0;
  double imp_sum = 0, peak_imp_sum = 0;
  double prev2 = 0, cum2 = 0, prev1 = 0, cum1 = 0;

  for (i = *len, wlen = 2 * 2 * 8; i > 1; wlen <<= 1, i >>= 1) {}
  w = (double *)calloc((size_t)wlen + 2, sizeof(double));
  wraps = (double *)malloc(sizeof(double) * (((size_t)wlen + 2) / 2));

  memcpy(w, *h, sizeof(double) * (size_t)*len);
  orc_rdft(wlen, +1, w);
  w[wlen] = w[1]; w[wlen + 1] = w[1] = 0; /* LSX_UNPACK, fft4g.h:46 */

  for (i = 0; i <= wlen; i += 2) {
    double angle = atan2(w[i + 1], w[i]);
    double detect = 2 * M_PI;
    double delta = angle - prev2;
    double adjust = detect * ((delta < -detect * .7) - (delta > detect * .7));
    prev2 = angle;
    cum2 += adjust;
    angle += cum2;
    detect = M_PI;
    delta = angle - prev1;
    adjust = detect * ((delta < -detect * .7) - (delta > detect * .7));
    prev1 = angle;
    cum1 += fabs(adjust);
    wraps[i >> 1] = cum1;
    w[i] = safe_log(sqrt(w[i] * w[i] + w[i + 1] * w[i + 1]));
    w[i + 1] = 0;
  }
  w[1] = w[wlen]; /* LSX_PACK */
  orc_rdft(wlen, -1, w);
  for (i = 0; i < wlen; ++i) w[i] *= 2. / wlen;

  for (i = 1; i < wlen / 2; ++i) { /* keep the causal part of the cepstrum */
    w[i] *= 2;
    w[i + wlen / 2] = 0;
  }
  orc_rdft(wlen, +1, w);

  for (i = 2; i < wlen; i += 2)
    w[i + 1] = phase1 * i / wlen * wraps[wlen >> 1] + (1 - phase1) * (w[i + 1] + wraps[i >> 1]) - wraps[i >> 1];

  w[0] = exp(w[0]); w[1] = exp(w[1]);
  for (i = 2; i < wlen; i += 2) {
    double x = exp(w[i]);
    w[i] = x * cos(w[i + 1]);
    w[i + 1] = x * sin(w[i + 1]);
  }
  orc_rdft(wlen, -1, w);
  for (i = 0; i < wlen; ++i) w[i] *= 2. / wlen;

  for (i = 0; i <= (int)(wraps[wlen >> 1] / M_PI + .5); ++i) {
    imp_sum += w[i];
    if (fabs(imp_sum) > fabs(peak_imp_sum)) { peak_imp_sum = imp_sum; peak = i; }
  }
  while (peak && fabs(w[peak - 1]) > fabs(w[peak]) && w[peak - 1] * w[peak] > 0) --peak;

  if (!phase1)
    begin = 0;
  else if (phase1 == 1)
    begin = peak - *len / 2;
  else {
    begin = (int)((.997 - (2 - phase1) * .22) * *len + .5);
    end = (int)((.997 + (0 - phase1) * .22) * *len + .5);
    begin = peak - (begin & ~3);
    end = peak + 1 + ((end + 3) & ~3);
    *len = end - begin;
    *h = (double *)realloc(*h, sizeof(double) * (size_t)*len);
  }
  for (i = 0; i < *len; ++i)
    (*h)[i] = w[(begin + (phase > 50 ? *len - 1 - i : i) + wlen) & (wlen - 1)];
  *post_len = phase > 50 ? peak - begin : begin + *len - (peak + 1);

  free(wraps);
  free(w);
}

/* =========================================================================================
 * Half-band decimator coefficient tables (literal constants, rate/rate_filters_generic.h:31-70)
 * and selection thresholds (:255-262; stored as float there).
 * ========================================================================================= */
static const double hb8[] = {0.3115465451887802, -0.08734497241282892, 0.03681452335604365,
  -0.01518925831569441, 0.005454118437408876, -0.001564400922162005, 0.0003181701445034203,
  -3.48001341225749e-5};
static const double hb9[] = {0.3122703613711853, -0.08922155288172305, 0.03913974805854332,
  -0.01725059723447163, 0.006858970092378141, -0.002304518467568703, 0.0006096426006051062,
  -0.0001132393923815236, 1.119795386287666e-5};
static const double hb10[] = {0.3128545521327376, -0.09075671986104322, 0.04109637155154835,
  -0.01906629512749895, 0.008184039342054333, -0.0030766775017262, 0.0009639607022414314,
  -0.0002358552746579827, 4.025184282444155e-5, -3.629779111541012e-6};
static const double hb11[] = {0.3133358837508807, -0.09203588680609488, 0.04276515428384758,
  -0.02067356614745591, 0.00942253142371517, -0.003856330993895144, 0.001363470684892284,
  -0.0003987400965541919, 9.058629923971627e-5, -1.428553070915318e-5, 1.183455238783835e-6};
static const double hb12[] = {0.3137392991811407, -0.0931182192961332, 0.0442050575271454,
  -0.02210391200618091, 0.01057473015666001, -0.00462766983973885, 0.001793630226239453,
  -0.0005961819959665878, 0.0001631475979359577, -3.45557865639653e-5, 5.06188341942088e-6,
  -3.877010943315563e-7};
static const double hb13[] = {0.3140822554324578, -0.0940458550886253, 0.04545990399121566,
  -0.02338339450796002, 0.01164429409071052, -0.005380686021429845, 0.002242915773871009,
  -0.000822047600000082, 0.0002572510962395222, -6.607320708956279e-5, 1.309926399120154e-5,
  -1.790719575255006e-6, 1.27504961098836e-7};
static const struct { int num; const double *c; float att; } half_firs[] = {
  {8, hb8, 136.51f}, {9, hb9, 152.32f}, {10, hb10, 168.07f},
  {11, hb11, 183.78f}, {12, hb12, 199.44f}, {13, hb13, 212.75f}};
#define N_HALF_FIRS 6

/* interpolated-polyphase selection table (rate/rate_filters_generic.h:724-746): per row, the
 * `scalar` (phase-bit budget, stored as float there) of interp[1], interp[2] and the interpolation
 * function index (1,2,3 = vpoly1,2,3; 0 = none).  Rows 12 and 13 (U100_0 / u100_*) need mode <= 1
 * and are unreachable through this API (mode is 3 or 5). */
static const struct { float s1; int f1; float s2; int f2; } poly_rows[] = {
  {7.2f, 1, 5.0f, 2}, {9.4f, 1, 6.7f, 2}, {12.4f, 1, 7.8f, 2}, {13.6f, 1, 9.3f, 2}, {10.5f, 2, 8.4f, 3}, {11.85f, 2, 9.0f, 3},
  {8.0f, 1, 5.3f, 2}, {8.6f, 1, 5.7f, 2}, {10.6f, 1, 6.75f, 2}, {12.6f, 1, 8.6f, 2}, {9.6f, 2, 7.6f, 3}, {11.4f, 2, 8.65f, 3},
  {0, 0, 0, 0}, {0, 0, 0, 0},
  {9, 1, 6, 2}, {11, 1, 7, 2}, {13, 1, 8, 2}, {10, 2, 8, 3}, {12, 2, 9, 3}};

/* =========================================================================================
 * FIFO of doubles (semantics of rate/fifo.h:58-201: append at the end, consume from the front,
 * trim from the end; storage strategy is our own).
 * ========================================================================================= */
typedef struct { double *buf; size_t cap, head, tail; } dfifo;

static void fifo_init(dfifo *f) { f->cap = 4096; f->buf = (double *)malloc(sizeof(double) * f->cap); f->head = f->tail = 0; }
static void fifo_free(dfifo *f) { free(f->buf); f->buf = NULL; }
static int fifo_count(const dfifo *f) { return (int)(f->tail - f->head); }
static double *fifo_front(dfifo *f) { return f->buf + f->head; }
static double *fifo_append(dfifo *f, size_t n)
{
  double *p;
  if (f->head == f->tail) f->head = f->tail = 0;
  if (f->tail + n > f->cap) {
    size_t live = f->tail - f->head;
    if (f->head) { memmove(f->buf, f->buf + f->head, sizeof(double) * live); f->head = 0; f->tail = live; }
    if (f->tail + n > f->cap) {
      while (f->tail + n > f->cap) f->cap *= 2;
      f->buf = (double *)realloc(f->buf, sizeof(double) * f->cap);
    }
  }
  p = f->buf + f->tail;
  f->tail += n;
  return p;
}
static void fifo_drop_front(dfifo *f, size_t n) { if (n <= f->tail - f->head) f->head += n; } /* fifo.h:165-175 */
static void fifo_trim_back(dfifo *f, size_t n) { f->tail -= n; }
static void fifo_keep(dfifo *f, size_t n) { f->tail = f->head + n; }

/* =========================================================================================
 * Engine types (rate/rate_base.h:83-128,224-231)
 * ========================================================================================= */
typedef struct { int dft_length, num_taps, post_peak; double *taps; double *spec; } dft_filt;
typedef struct { double *poly; int poly_len; dft_filt dft[2]; } shared_t;

typedef struct {
  int kind;
  dfifo in;
  int pre, pre_post, preload;
  double out_in_ratio;
  int dft_idx, L, remL, remM, step_int;
  int n, phase_bits, order;
  int64_t at, step;
  const double *hb; int hb_n;
  int remL0;
} stage_t;

typedef struct {
  double factor;
  size_t samples_in, samples_out;
  int num_stages;
  stage_t *st; /* num_stages + 1 entries; the last one only carries the output fifo */
  size_t in_rate, out_rate;
} chan_t;

#define MAX_TRACE 16
struct orc_handle {
  int nch;
  chan_t *ch;
  shared_t sh;
  size_t isamp_max;
  orc_design_call trace[MAX_TRACE];
  int ntrace;
};

static double *traced_design(orc_handle *h, double Fp, double Fs, double Fn, double att, int *num_taps, int k, double beta)
{
  double *r = orc_design_lpf(Fp, Fs, Fn, att, num_taps, k, beta);
  if (h->ntrace < MAX_TRACE) {
    orc_design_call *c = &h->trace[h->ntrace++];
    c->Fp = Fp; c->Fs = Fs; c->Fn = Fn; c->att = att; c->k = k; c->num_taps = *num_taps; c->beta = g_last_beta;
  }
  return r;
}

/* rate/prepare_coefs.h:20-46 (generic part; multiplier is always 1 here) */
static double *make_poly_table(const double *taps, int n, int phases, int order, int *out_len)
{
  int i, j, len = n * phases * (order + 1);
  double *tab = (double *)calloc((size_t)len, sizeof(double));
  double fm1 = taps[0], f1 = 0, f2 = 0;
  for (i = n - 1; i >= 0; --i)
    for (j = phases - 1; j >= 0; --j) {
      double f0 = fm1, b = 0, c = 0, d = 0;
      int pos = i * phases + j - 1;
      double *slot = tab + (size_t)(n * (order + 1)) * j + (size_t)(order + 1) * (n - 1 - i);
      fm1 = pos > 0 ? taps[pos - 1] : 0;
      switch (order) {
        case 1: b = f1 - f0; break;
        case 2: b = f1 - (.5 * (f2 + f0) - f1) - f0; c = .5 * (f2 + f0) - f1; break;
        case 3: c = .5 * (f1 + fm1) - f0; d = (1 / 6.) * (f2 - f1 + fm1 - f0 - 4 * c); b = f1 - f0 - d - c; break;
        default: break;
      }
      slot[order] = f0;
      if (order > 0) slot[order - 1] = b;
      if (order > 1) slot[order - 2] = c;
      if (order > 2) slot[order - 3] = d;
      f2 = f1; f1 = f0;
    }
  *out_len = len;
  return tab;
}

/* rate/rate_base.h:156-192 */
static void dft_stage_setup(orc_handle *h, int which, double Fp, double Fs, double Fn, double att,
                            double phase, stage_t *s, int L, int M)
{
  dft_filt *f = &h->sh.dft[which];
  if (!f->num_taps) {
    int num_taps = 0, N, i;
    int k = phase == 50 && is_pow2_ge2(L) && Fn == L ? L << 1 : 4;
    double *taps = traced_design(h, Fp, Fs, Fn, att, &num_taps, -k, -1.);
    if (phase != 50) orc_fir_to_phase(&taps, &num_taps, &f->post_peak, phase);
    else f->post_peak = num_taps / 2;
    N = orc_dft_length(num_taps);
    f->spec = (double *)calloc((size_t)N, sizeof(double));
    for (i = 0; i < num_taps; ++i)
      f->spec[(i + N - num_taps + 1) & (N - 1)] = taps[i] / N * 2 * L;
    f->taps = taps;
    f->num_taps = num_taps;
    f->dft_length = N;
    orc_rdft(N, +1, f->spec);
  }
  s->kind = ORC_STAGE_DFT;
  s->preload = f->post_peak / L;
  s->remL = s->remL0 = f->post_peak % L;
  s->L = L;
  s->step_int = abs(3 - M) == 1 && Fs == 1 ? -M / 2 : M;
  s->dft_idx = which;
}

/* rate/rate_base.h:247-423.  Decides the chain (half-bands, pre dft, arbitrary-ratio poly, post
 * dft), designs the shared filters on first use and preloads the fifos. */
static void chan_setup(orc_handle *h, chan_t *p, double factor, double bits, double phase, double bw_pc,
                       double aa_pc, int rolloff /*0 none,1 small,2 medium*/, int maintain_3dB,
                       int interpolator, int max_coefs_size, int no_small_int_opt)
{
  double att = (bits + 1) * to_dB(2.), attArb = att;
  double tbw0 = 1 - bw_pc / 100, Fs_a = 2 - aa_pc / 100;
  double arbM = factor, tbw_tighten = 1;
  int n = 0, i, preL = 1, preM = 1, shift = 0, arbL = 1, postL = 1, postM = 1;
  int upsample = 0, rational = 0, iOpt = !no_small_int_opt;
  int mode = rolloff > 1 ? (factor > 1 || bw_pc > (67 + 5 / 8.)) : (int)ceil(2 + (bits - 17) / 4);
  int have_pre, have_arb, have_post;
  stage_t *s;

  p->factor = factor;
  while (!n++) { /* :283-310 */
    int try_i, L, M, x, maxL = interpolator > 0 ? 1 : mode ? 2048 : (int)ceil(max_coefs_size * 1000. / (44 * sizeof(double)));
    double d, epsilon = 0, frac;
    upsample = arbM < 1;
    for (i = (int)(arbM * .5), shift = 0; i >>= 1; arbM *= .5, ++shift) {}
    preM = upsample || (arbM > 1.5 && arbM < 2);
    postM = 1 + (arbM > 1 && preM); arbM /= postM;
    preL = 1 + (!preM && arbM < 2) + (upsample && mode); arbM *= preL;
    if ((frac = arbM - (int)arbM) != 0)
      epsilon = fabs(floor(frac * TWO32 + .5) / (frac * TWO32) - 1);
    for (i = 1, rational = !frac; i <= maxL && !rational; ++i) {
      d = frac * i; try_i = (int)(d + .5);
      if ((rational = fabs(try_i / d - 1) <= epsilon)) {
        if (try_i == i) { arbM = ceil(arbM); x = arbM > 3; shift += x; arbM /= 1 + x; }
        else { arbM = i * (int)arbM + try_i; arbL = i; }
      }
    }
    L = preL * arbL; M = (int)(arbM * postM); x = (L | M) & 1; L >>= !x; M >>= !x;
    if (iOpt && postL == 1 && (d = preL * arbL / arbM) > 4 && d != 5) {
      for (postL = 4, i = (int)(d / 16); i >>= 1; postL <<= 1) {}
      arbM = arbM * postL / arbL / preL; arbL = 1; n = 0;
    } else if (rational && (imax(L, M) < 3 + 2 * iOpt || L * M < 6 * iOpt)) {
      preL = L; preM = M; arbM = arbL = postM = 1;
    }
    if (!mode && (!rational || !n)) { ++mode; n = 0; }
  }
  have_pre = preM * preL != 1;
  have_arb = arbM * arbL != 1;
  have_post = postM * postL != 1;

  p->num_stages = shift + have_pre + have_arb + have_post;
  p->st = (stage_t *)calloc((size_t)p->num_stages + 1, sizeof(stage_t));

  if ((n = p->num_stages) > 1) { /* attenuation budget :317-321 */
    if (have_arb) { att += to_dB(2.); attArb = att; --n; }
    att += to_dB((double)n);
  }

  for (n = 0; n + 1 < N_HALF_FIRS && att > half_firs[n].att; ++n) {}
  for (i = 0, s = p->st; i < shift; ++i, ++s) { /* :324-328 */
    s->kind = ORC_STAGE_HALF;
    s->hb = half_firs[n].c; s->hb_n = half_firs[n].num;
    s->pre_post = 4 * half_firs[n].num;
    s->preload = s->pre = s->pre_post >> 1;
  }

  if (have_pre) { /* :330-341 */
    if (maintain_3dB && have_post) {
      double a = att;
      double tbw3 = tbw0 * ((1.6e-6 * a - 7.5e-4) * a + .646); /* TO_3dB, :240 */
      double x = ((2.1429e-4 - 5.2083e-7 * att) * att - .015863) * att + 3.95;
      x = att * pow((tbw0 - tbw3) / (postM / (factor * postL) - 1 + tbw0), x);
      if (x > .035) tbw_tighten = ((4.3074e-3 - 3.9121e-4 * x) * x - .040009) * x + 1.0014;
    }
    dft_stage_setup(h, 0, 1 - tbw0 * tbw_tighten, Fs_a, preM ? imax(preL, preM) : arbM / arbL, att,
                    phase, &p->st[shift], preL, imax(preM, 1));
  }

  if (have_arb) { /* :350-410 */
    stage_t *a = &p->st[shift + have_pre];
    int row = 6 * (upsample + !!preM) + mode - !upsample;
    int order = 0, num_coefs = 0, phase_bits = 0, phases = 0, coefs_size, has_next;
    double x = .5, at = 0, Fp, Fs, Fn, mult = upsample ? 1 : arbL / arbM;

    Fn = !upsample && preM ? (x = arbM / arbL) : 1;
    Fp = !preM ? mult : mode ? .5 : 1;
    Fs = 2 - Fp;
    Fp *= 1 - tbw0;
    if (rolloff > 1 && mode) Fp = !preM ? mult * .5 - .125 : mult * .05 + .1;
    else if (rolloff == 1) Fp = Fs - (Fs - .148 * x - Fp * .852) * (.00813 * bits + .973);

    i = (interpolator < 0 ? !rational : imax(interpolator, !rational)) - 1;
    do {
      double scalar;
      ++i;
      scalar = i == 0 ? 0 : i == 1 ? poly_rows[row].s1 : poly_rows[row].s2;
      if (i) { arbM /= arbL; arbL = 1; rational = 0; }
      phase_bits = (int)ceil(scalar + log(mult) / log(2.));
      phases = !rational ? (1 << phase_bits) : arbL;
      {
        int phases0 = imax(phases, 19), n0 = 0;
        traced_design(h, Fp, Fs, -Fn, attArb, &n0, phases0, -1.);
        num_coefs = n0 / phases0 + 1; num_coefs += num_coefs & !preM;
      }
      if ((num_coefs & 1) && rational && (arbL & 1)) { phases <<= 1; arbL <<= 1; arbM *= 2; }
      at = arbL * .5 * (num_coefs & 1);
      order = i + (i && mode > 4);
      coefs_size = num_coefs * phases * (order + 1) * (int)sizeof(double);
      has_next = i == 0 ? poly_rows[row].f1 != 0 : i == 1 ? poly_rows[row].f2 != 0 : 0;
    } while (interpolator < 0 && i < 2 && has_next && coefs_size / 1000 > max_coefs_size);

    if (!h->sh.poly) {
      int num_taps = num_coefs * phases - 1;
      double *taps = traced_design(h, Fp, Fs, Fn, attArb, &num_taps, phases, -1.);
      h->sh.poly = make_poly_table(taps, num_coefs, phases, order, &h->sh.poly_len);
      free(taps);
    }
    a->kind = ORC_STAGE_POLY;
    a->order = order;
    a->pre_post = num_coefs - 1;
    a->preload = (num_coefs - 1) >> 1;
    a->n = num_coefs;
    a->phase_bits = phase_bits;
    a->L = arbL;
    a->at = (int64_t)(at * TWO32 + .5);
    a->step = (int64_t)(arbM * TWO32 + .5);
    a->out_in_ratio = TWO32 * arbL / a->step;
  }

  if (have_post) /* :412-415 */
    dft_stage_setup(h, 1, 1 - (1 - (1 - tbw0) * (upsample ? factor * postL / postM : 1)) * tbw_tighten,
                    Fs_a, (double)imax(postL, postM), att, phase,
                    &p->st[shift + have_pre + have_arb], postL, postM);

  for (i = 0, s = p->st; i <= p->num_stages; ++i, ++s) { /* :417-422 */
    fifo_init(&s->in);
    if (i < p->num_stages && s->preload) memset(fifo_append(&s->in, (size_t)s->preload), 0, sizeof(double) * (size_t)s->preload);
  }
}

/* =========================================================================================
 * Stage kernels
 * ========================================================================================= */
static int stage_avail(stage_t *s) { return imax(0, fifo_count(&s->in) - s->pre_post); } /* rate_base.h:130 */

/* rate/rate_filters_generic.h:80-249 (h8..h13 share one body here) */
static void run_half(stage_t *s, dfifo *out)
{
  const double *x = fifo_front(&s->in) + s->pre;
  int i, k, num_out = (stage_avail(s) + 1) / 2;
  double *y = fifo_append(out, (size_t)num_out);
  for (i = 0; i < num_out; ++i, x += 2) {
    double sum = x[0] * .5;
    for (k = 0; k < s->hb_n; ++k) sum += (x[-(2 * k + 1)] + x[2 * k + 1]) * s->hb[k];
    y[i] = sum;
  }
  fifo_drop_front(&s->in, (size_t)(2 * num_out));
}

/* rate/dft_filter.h:60-190 */
static void run_dft(const shared_t *sh, stage_t *s, dfifo *out)
{
  const dft_filt *f = &sh->dft[s->dft_idx];
  const double *H = f->spec;
  const int N = f->dft_length, ov = f->num_taps - 1;
  int i, j, avail = imax(0, fifo_count(&s->in));

  while (s->remL + s->L * avail >= N) {
    int span = N - ov - s->remL + s->L - 1;
    int take = span / s->L, take_rem = span % s->L;
    const double *x = fifo_front(&s->in);
    double *y;
    fifo_drop_front(&s->in, (size_t)take);
    avail -= take;
    y = fifo_append(out, (size_t)N);

    if (is_pow2_ge2(s->L)) { /* spectrum replication = zero stuffing, :86-104 */
      int P = N / s->L, w;
      memcpy(y, x, sizeof(double) * (size_t)P);
      orc_rdft(P, +1, y);
      for (i = P + 2; i < 2 * P; i += 2) { y[i] = y[2 * P - i]; y[i + 1] = -y[2 * P - i + 1]; }
      y[P] = y[1]; y[P + 1] = 0; y[1] = y[0];
      for (w = 2 * P, i = 2 * P; i < N; i += w, w <<= 1) { memcpy(y + i, y, sizeof(double) * (size_t)w); y[i + 1] = 0; }
    } else {
      if (s->L == 1) memcpy(y, x, sizeof(double) * (size_t)N);
      else { /* time-domain zero stuffing, :109-115 */
        memset(y, 0, sizeof(double) * (size_t)N);
        for (j = 0, i = s->remL; i < N; ++j, i += s->L) y[i] = x[j];
        s->remL = s->L - 1 - take_rem;
      }
      orc_rdft(N, +1, y);
    }
    y[0] *= H[0];
    if (s->step_int > 0) {
      y[1] *= H[1];
      for (i = 2; i < N; i += 2) {
        double t = y[i];
        y[i] = H[i] * t - H[i + 1] * y[i + 1];
        y[i + 1] = H[i + 1] * t + H[i] * y[i + 1];
      }
      orc_rdft(N, -1, y);
      if (s->step_int != 1) { /* time-domain decimation, :148-154 */
        for (j = 0, i = s->remM; i < N - ov; ++j, i += s->step_int) y[j] = y[i];
        s->remM = i - (N - ov);
        fifo_trim_back(out, (size_t)(N - j));
      } else fifo_trim_back(out, (size_t)ov);
    } else { /* frequency-domain decimation by 2^m, :157-188 */
      int m = -s->step_int, Nd = N >> m;
      for (i = 2; i < Nd; i += 2) {
        double t = y[i];
        y[i] = H[i] * t - H[i + 1] * y[i + 1];
        y[i + 1] = H[i + 1] * t + H[i] * y[i + 1];
      }
      y[1] = H[i] * y[i] - H[i + 1] * y[i + 1];
      orc_rdft(Nd, -1, y);
      fifo_trim_back(out, (size_t)((((1 << m) - 1) * N + ov) >> m));
    }
  }
}

/* rate/rate_filters_generic.h:272-305 (vpoly0) and :311-504 (vpoly1..3, fixed-point clock branch;
 * the hi_prec_clock branch is dead: rate_base.h:703) */
static void run_poly(const shared_t *sh, stage_t *s, dfifo *out)
{
  const double *x = fifo_front(&s->in) + s->pre;
  int i, j, num_in = stage_avail(s), max_out = 1 + (int)(num_in * s->out_in_ratio);
  double *y = fifo_append(out, (size_t)max_out);
  const double *tab = sh->poly;
  const int n = s->n;

  if (s->order == 0) {
    int at = (int)(s->at >> 32), step = (int)(s->step >> 32);
    for (i = 0; at < num_in * s->L; ++i, at += step) {
      const double *xs = x + at / s->L;
      const double *c = tab + (size_t)n * (at % s->L);
      double sum = 0;
      for (j = 0; j < n; ++j) sum += c[j] * xs[j];
      y[i] = sum;
    }
    fifo_trim_back(out, (size_t)(max_out - i));
    fifo_drop_front(&s->in, (size_t)(at / s->L));
    s->at = (int64_t)(at % s->L) << 32;
  } else {
    const int o1 = s->order + 1;
    int64_t at = s->at;
    for (i = 0; (int)(at >> 32) < num_in; ++i, at += s->step) {
      const double *xs = x + (int)(at >> 32);
      uint32_t frac = (uint32_t)at;
      int ph = (int)(frac >> (32 - s->phase_bits));
      const double *c = tab + (size_t)n * ph * o1;
      double t = (double)(uint32_t)(frac << s->phase_bits) * (1 / TWO32);
      double sum = 0;
      for (j = 0; j < n; ++j, c += o1) {
        double v = c[0];
        int q;
        for (q = 1; q < o1; ++q) v = v * t + c[q];
        sum += v * xs[j];
      }
      y[i] = sum;
    }
    fifo_trim_back(out, (size_t)(max_out - i));
    fifo_drop_front(&s->in, (size_t)(at >> 32));
    s->at = at & 0xffffffffLL;
  }
}

/* rate/rate_base.h:425-432 */
static void chan_process(const shared_t *sh, chan_t *p)
{
  int i;
  for (i = 0; i < p->num_stages; ++i) {
    stage_t *s = &p->st[i];
    dfifo *out = &p->st[i + 1].in;
    if (s->kind == ORC_STAGE_HALF) run_half(s, out);
    else if (s->kind == ORC_STAGE_DFT) run_dft(sh, s, out);
    else run_poly(sh, s, out);
  }
}

/* rate/rate_base.h:434-443 */
static double *chan_input(chan_t *p, size_t n)
{
  p->samples_in += n;
  while (p->samples_in > p->in_rate && p->samples_out > p->out_rate) {
    p->samples_in -= p->in_rate;
    p->samples_out -= p->out_rate;
  }
  return fifo_append(&p->st[0].in, n);
}

/* rate/rate_base.h:445-450 */
static const double *chan_output(chan_t *p, size_t *n)
{
  dfifo *f = &p->st[p->num_stages].in;
  const double *r = fifo_front(f);
  size_t have = (size_t)fifo_count(f);
  if (*n > have) *n = have;
  p->samples_out += *n;
  fifo_drop_front(f, *n);
  return r;
}

/* rate/rate_base.h:454-468 */
static void chan_flush(const shared_t *sh, chan_t *p)
{
  dfifo *f = &p->st[p->num_stages].in;
  size_t target = (size_t)(p->samples_in / p->factor + .5);
  if (target > p->samples_out) {
    size_t remaining = target - p->samples_out;
    while ((size_t)fifo_count(f) < remaining) {
      memset(chan_input(p, 1024), 0, sizeof(double) * 1024);
      chan_process(sh, p);
    }
    fifo_keep(f, remaining);
    p->samples_out = p->samples_in = 0;
  }
}

/* =========================================================================================
 * Public wrapper (rate/rate_base.h:497-741, rate/rate_uni.c:27-111)
 * ========================================================================================= */
int orc_open(const orc_config *cfg, int nchannels, orc_handle **out)
{
  orc_handle *h;
  double factor, bits, bw0, aa, rej;
  int i, quality, rolloff;
  if (!out) return ORC_INVPARAM;
  *out = NULL;
  if (!cfg || nchannels < 1) return ORC_INVPARAM;
  factor = (double)cfg->in_rate / (double)cfg->out_rate;
  if (factor > 5644.8 || factor < 1.0 / 5644.8) return ORC_INVPARAM; /* rate_base.h:528 */

  /* convert_settings, rate_base.h:674-704 */
  if (cfg->quality == 0) { quality = 6; rolloff = 0; } else { quality = 4; rolloff = 1; }
  aa = cfg->allow_aliasing ? cfg->bandwidth : 100;
  bits = 16 + 4 * imax(quality - 3, 0);
  rej = bits * to_dB(2.);
  bw0 = 100 - (100 - cfg->bandwidth) / ((1.6e-6 * rej - 7.5e-4) * rej + .646);

  h = (orc_handle *)calloc(1, sizeof(*h));
  if (!h) return ORC_ENOMEM;
  h->nch = nchannels;
  h->ch = (chan_t *)calloc((size_t)nchannels, sizeof(chan_t));
  h->isamp_max = 1048576;
  if (factor < 1) h->isamp_max = (size_t)(h->isamp_max * factor);
  for (i = 0; i < nchannels; ++i) {
    h->ch[i].in_rate = cfg->in_rate;
    h->ch[i].out_rate = cfg->out_rate;
    chan_setup(h, &h->ch[i], factor, bits, cfg->phase, bw0, aa, rolloff, 1, -1, 400, 0);
  }
  *out = h;
  return ORC_OK;
}

void orc_close(orc_handle **ph)
{
  orc_handle *h;
  int c, i;
  if (!ph || !*ph) return;
  h = *ph;
  for (c = 0; c < h->nch; ++c) {
    for (i = 0; i <= h->ch[c].num_stages; ++i) fifo_free(&h->ch[c].st[i].in);
    free(h->ch[c].st);
  }
  free(h->ch);
  free(h->sh.poly);
  for (i = 0; i < 2; ++i) { free(h->sh.dft[i].taps); free(h->sh.dft[i].spec); }
  free(h);
  *ph = NULL;
}

size_t orc_isamp_max(const orc_handle *h) { return h->isamp_max; }

static void scatter_in(double *dst, const float *src, size_t nch, size_t n) /* deinterleave, :565-569 */
{ size_t k; for (k = 0; k < n; ++k) dst[k] = (double)src[k * nch]; }
static void gather_out(float *dst, const double *src, size_t nch, size_t n) /* interleave, :559-563 */
{ size_t k; for (k = 0; k < n; ++k) dst[k * nch] = (float)src[k]; }

int orc_push(orc_handle *h, const float *ibuf, size_t isamp)
{
  int c;
  if (!h) return ORC_NULLHANDLE;
  if (!ibuf || !isamp) return ORC_OK;
  if (isamp > h->isamp_max) isamp = h->isamp_max;
  for (c = 0; c < h->nch; ++c) {
    double *t = chan_input(&h->ch[c], isamp);
    scatter_in(t, ibuf + c, (size_t)h->nch, isamp);
    chan_process(&h->sh, &h->ch[c]);
  }
  return ORC_OK;
}

int orc_pull(orc_handle *h, float *obuf, size_t osamp, size_t *ogen)
{
  int c;
  size_t got = 0;
  if (!h) return ORC_NULLHANDLE;
  if (!obuf || !osamp) { if (ogen) *ogen = 0; return ORC_OK; }
  for (c = 0; c < h->nch; ++c) {
    const double *s;
    got = osamp;
    s = chan_output(&h->ch[c], &got);
    gather_out(obuf + c, s, (size_t)h->nch, got);
  }
  if (ogen) *ogen = got;
  return ORC_OK;
}

int orc_flow(orc_handle *h, const float *ibuf, float *obuf, size_t isamp, size_t osamp, size_t *iused, size_t *ogen)
{
  size_t dummy_i, dummy_o;
  int c;
  if (!h) return ORC_NULLHANDLE;
  if (!iused) iused = &dummy_i;
  if (!ogen) ogen = &dummy_o;
  if (!ibuf) { isamp = 0; *iused = 0; }
  if (isamp > h->isamp_max) isamp = h->isamp_max;
  for (c = 0; c < h->nch; ++c) {
    size_t got = osamp, got2;
    const double *s = chan_output(&h->ch[c], &got);
    *iused = 0;
    gather_out(obuf + c, s, (size_t)h->nch, got);
    *ogen = got;
    if (isamp) {
      double *t = chan_input(&h->ch[c], isamp);
      scatter_in(t, ibuf + c, (size_t)h->nch, isamp);
      chan_process(&h->sh, &h->ch[c]);
      *iused = isamp;
    }
    if (got < osamp) {
      got2 = osamp - got;
      s = chan_output(&h->ch[c], &got2);
      gather_out(obuf + c + got * (size_t)h->nch, s, (size_t)h->nch, got2);
      *ogen += got2;
    }
  }
  return ORC_OK;
}

int orc_drain(orc_handle *h)
{
  int c;
  if (!h) return ORC_NULLHANDLE;
  for (c = 0; c < h->nch; ++c) chan_flush(&h->sh, &h->ch[c]);
  return ORC_OK;
}

/* ---- introspection ---- */
int orc_num_stages(const orc_handle *h) { return h->ch[0].num_stages; }

int orc_stage_info_get(const orc_handle *h, int idx, orc_stage_info *o)
{
  const stage_t *s;
  if (idx < 0 || idx >= h->ch[0].num_stages) return -1;
  s = &h->ch[0].st[idx];
  memset(o, 0, sizeof(*o));
  o->kind = s->kind;
  o->pre = s->pre; o->pre_post = s->pre_post; o->preload = s->preload;
  o->out_in_ratio = s->out_in_ratio;
  if (s->kind == ORC_STAGE_DFT) {
    const dft_filt *f = &h->sh.dft[s->dft_idx];
    o->L = s->L; o->step_int = s->step_int; o->remL = s->remL0;
    o->num_taps = f->num_taps; o->dft_length = f->dft_length; o->post_peak = f->post_peak;
  } else if (s->kind == ORC_STAGE_POLY) {
    o->L = s->L; o->n = s->n; o->interp_order = s->order; o->phase_bits = s->phase_bits;
    o->at = s->at; o->step = s->step; o->step_int = (int)(s->step >> 32);
  } else {
    o->n = s->hb_n;
  }
  return 0;
}

const double *orc_dft_taps(const orc_handle *h, int which, int *len) { *len = h->sh.dft[which].num_taps; return h->sh.dft[which].taps; }
const double *orc_dft_spectrum(const orc_handle *h, int which, int *len) { *len = h->sh.dft[which].dft_length; return h->sh.dft[which].spec; }
const double *orc_poly_table(const orc_handle *h, int *len) { *len = h->sh.poly_len; return h->sh.poly; }

const double *orc_stage_fifo(const orc_handle *h, int channel, int stage, int *len)
{
  dfifo *f = &h->ch[channel].st[stage].in;
  *len = fifo_count(f);
  return fifo_front(f);
}

int orc_design_trace(const orc_handle *h, orc_design_call *calls, int max_calls)
{
  int i, n = imin(h->ntrace, max_calls);
  for (i = 0; i < n; ++i) calls[i] = h->trace[i];
  return h->ntrace;
}
