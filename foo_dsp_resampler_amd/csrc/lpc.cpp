// See lpc.hpp.  Arithmetic (float window and prediction sums, double autocorrelation and recursion)
// follows lpc/lpc.cpp of the reference so that both sides extrapolate the same samples.
#include "lpc.hpp"

#include <vector>

namespace rsmp {

namespace {

// lpc.cpp:85-93: Welch window w[i] = 1 - ((i + 1 - n2) / n2)^2, n2 = (len + 1) / 2, in float
void welch(std::vector<float> &x, size_t len)
{
  const float n2 = (len + 1) / 2.0f;
  for (size_t i = 0; i < len; ++i) {
    const float k = (float(int(i) + 1) - n2) / n2;
    x[i] *= 1.0f - k * k;
  }
}

// lpc.cpp:96-110: r[j] = sum_i x[i] x[i-j], accumulated in double
void autocorrelate(const std::vector<float> &x, size_t len, std::vector<double> &r, int order)
{
  for (int j = order; j >= 0; --j) {
    double d = 0;
    for (size_t i = size_t(j); i < len; ++i) d += double(x[i]) * x[i - size_t(j)];
    r[size_t(j)] = d;
  }
}

// lpc.cpp:112-165: Levinson-Durbin with early stop, damping and the constant-signal fallback.
// Returns the usable order.
int levinson(const std::vector<double> &r, std::vector<double> &a, int order)
{
  int used = order;
  double err = r[0] * (1. + 1e-10);
  const double floor_ = 1e-9 * r[0] + 1e-10;
  for (int i = 0; i < order; ++i) {
    if (err < floor_) {
      for (int k = i; k < order; ++k) a[size_t(k)] = 0;
      used = i;
      break;
    }
    double refl = -r[size_t(i) + 1];
    for (int j = 0; j < i; ++j) refl -= a[size_t(j)] * r[size_t(i - j)];
    refl /= err;
    a[size_t(i)] = refl;
    int j = 0;
    for (; j < i / 2; ++j) {
      const double t = a[size_t(j)];
      a[size_t(j)] += refl * a[size_t(i - 1 - j)];
      a[size_t(i - 1 - j)] += refl * t;
    }
    if (i & 1) a[size_t(j)] += a[size_t(j)] * refl;
    err *= 1.0 - refl * refl;
  }
  double damp = 0.999;
  for (int j = 0; j < used; ++j) {
    a[size_t(j)] *= damp;
    damp *= 0.999;
  }
  if (used == 0) {
    used = 1;
    a[0] = -1;
  }
  return used;
}

inline float clamp10(float v) { return v > 10.f ? 10.f : v < -10.f ? -10.f : v; }

} // namespace

void lpc_extrapolate(float *data, size_t data_len, int nch, int order, size_t extra_bkwd, size_t extra_fwd)
{
  // one channel at a time in a scratch line [extra_bkwd | data_len | extra_fwd]
  std::vector<float> line(extra_bkwd + data_len + extra_fwd), win(data_len);
  std::vector<double> r(size_t(order) + 1), a(size_t(order) > 0 ? size_t(order) : 1);
  float *x = line.data() + extra_bkwd;
  for (int c = 0; c < nch; ++c) {
    for (float &v : line) v = 0;
    for (size_t i = 0; i < data_len; ++i) win[i] = x[i] = data[i * size_t(nch) + size_t(c)];
    welch(win, data_len);
    autocorrelate(win, data_len, r, order);
    const int used = levinson(r, a, order);

    if (extra_fwd) { // lpc.cpp:170-182
      float *p = x + data_len - used;
      for (size_t i = 0; i < extra_fwd; ++i) {
        float sum = 0;
        for (int j = 0; j < used; ++j) sum -= p[i + size_t(j)] * float(a[size_t(used - 1 - j)]);
        p[size_t(used) + i] = clamp10(sum);
      }
      for (size_t i = data_len; i < data_len + extra_fwd; ++i) data[i * size_t(nch) + size_t(c)] = x[i];
    }
    if (extra_bkwd) { // lpc.cpp:183-195 (time-reversed recursion)
      float *p = x - 1 + used;
      for (size_t i = 0; i < extra_bkwd; ++i) {
        float sum = 0;
        for (int j = 0; j < used; ++j) sum -= p[-ptrdiff_t(i) - j] * float(a[size_t(used - 1 - j)]);
        p[-ptrdiff_t(used) - ptrdiff_t(i)] = clamp10(sum);
      }
      for (ptrdiff_t i = -ptrdiff_t(extra_bkwd); i < 0; ++i)
        data[i * ptrdiff_t(nch) + c] = x[i];
    }
  }
}

} // namespace rsmp
