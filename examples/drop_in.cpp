// Minimal host program against the drop-in ABI (include/ratelib.h), calling it the way the reference plugin's
// resampler_link does (chain.h:36-40) and its on_chunk loop does (foo_dsp_rate.cpp:182-202): open, then
// for every chunk push + pull-until-empty, finally drain + pull-until-empty, close.
//
//   g++ -O2 -Iinclude examples/drop_in.cpp -o drop_in -Lfoo_dsp_resampler_amd -lratelib_amd -Wl,-rpath,$PWD/foo_dsp_resampler_amd
//   ./drop_in 44100 96000 2 1.0     # in_rate out_rate channels seconds  -> prints frame counts and a checksum
//
// Needs an MI355X (the library has no CPU path: init_ratelib fails without a HIP device).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <vector>

#include "ratelib.h"

static void on_alloc_failure() { throw std::bad_alloc(); } // what the plugin registers (foo_dsp_rate.cpp:23-24)

static void check(int err, const char *what)
{
  if (err) {
    std::fprintf(stderr, "%s: %s\n", what, RR_strerror(err));
    std::exit(2);
  }
}

int main(int argc, char **argv)
{
  const size_t in_rate = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 44100;
  const size_t out_rate = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 96000;
  const int nch = argc > 3 ? std::atoi(argv[3]) : 2;
  const double seconds = argc > 4 ? std::atof(argv[4]) : 1.0;

  if (init_ratelib(on_alloc_failure) != 0) {
    std::fprintf(stderr, "init_ratelib failed (no HIP device?)\n");
    return 3;
  }
  RR_config cfg;
  cfg.in_rate = in_rate;
  cfg.out_rate = out_rate;
  cfg.phase = RR_linear;
  cfg.bandwidth = 95.0;
  cfg.allow_aliasing = 0;
  cfg.quality = RR_best;
  RR_handle *h = nullptr;
  check(RR_open(&cfg, nch, &h), "RR_open");

  const size_t total_in = size_t(seconds * double(in_rate));
  const size_t chunk = in_rate / 10; // the plugin stages 1/10 s per chunk
  std::vector<fb_sample_t> in(chunk * nch), out(8192 * size_t(nch));
  size_t produced = 0, pushed = 0;
  double checksum = 0.0;
  auto pull_all = [&] {
    for (;;) {
      size_t got = 0;
      check(RR_pull(h, out.data(), 8192, &got), "RR_pull");
      if (!got) break;
      for (size_t i = 0; i < got * nch; ++i) checksum += std::fabs(double(out[i]));
      produced += got;
    }
  };
  while (pushed < total_in) {
    const size_t n = total_in - pushed < chunk ? total_in - pushed : chunk;
    for (size_t i = 0; i < n; ++i)
      for (int c = 0; c < nch; ++c) in[i * nch + c] = 0.5f * float(std::sin(2.0 * M_PI * (440.0 + 110.0 * c) * double(pushed + i) / double(in_rate)));
    check(RR_push(h, in.data(), n), "RR_push");
    pushed += n;
    pull_all();
  }
  check(RR_drain(h), "RR_drain");
  pull_all();
  RR_close(&h);
  close_ratelib();

  const size_t expected = size_t(double(total_in) * double(out_rate) / double(in_rate) + 0.5);
  std::printf("in %zu frames @ %zu Hz -> out %zu frames @ %zu Hz (expected %zu), mean |sample| %.6f\n", pushed, in_rate,
              produced, out_rate, expected, checksum / double(produced ? produced * nch : 1));
  return produced == expected && h == nullptr ? 0 : 1;
}
