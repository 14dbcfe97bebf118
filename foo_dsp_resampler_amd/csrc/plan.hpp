// Stage planner: turns an RR_config into the chain of stages the reference would build
// (rate/rate_base.h:247-423 `rate_init`, :674-704 `convert_settings`) plus the designed filters.
// Host only; the engine uploads the tables to HBM.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace rsmp {

struct Config { // field-for-field RR_config (rate/ratelib.h:53-63); quality: 0 = best, 1 = normal
  size_t in_rate = 0, out_rate = 0;
  double phase = 50, bandwidth = 95;
  int allow_aliasing = 0;
  int quality = 0;
};

enum class StageKind { Half = 0, Dft = 1, Poly = 2 };

struct DftFilter {
  int num_taps = 0, N = 0, post_peak = 0;
  std::vector<double> taps; // after phase conversion, unscaled
};

struct StageSpec {
  StageKind kind = StageKind::Dft;
  int pre = 0, pre_post = 0, preload = 0; // fifo contract, rate_base.h:100-102
  double out_in_ratio = 0;
  // Dft
  int filt = 0;     // which of the two shared filters
  int L = 1;        // zero-stuffing factor (Dft) / number of phases when rational (Poly)
  int step = 1;     // Dft: +M time-domain decimation, -m frequency-domain decimation by 2^m
  int remL0 = 0;
  // Poly
  int n = 0, order = 0, phase_bits = 0;
  int64_t at0 = 0, step64 = 0; // 32.32 fixed point
  // Half
  int hb_n = 0;
  const double *hb = nullptr;
};

struct DesignCall { double Fp, Fs, Fn, att; int k, num_taps; };

struct ChainPlan {
  Config cfg;
  double factor = 1;
  size_t isamp_max = 0;
  std::vector<StageSpec> stages;
  DftFilter dft[2];
  std::vector<double> poly_table; // [phase][tap][order+1], rate/prepare_coefs.h:20-46
  std::vector<DesignCall> trace;
  std::string describe() const; // JSON, for tests and INTEGRATION
};

// Returns 0 (RR_OK) or 6 (RR_INVPARAM) when the ratio is outside [1/5644.8, 5644.8]
// (rate_base.h:528) or the config is out of the asserted ranges (rate_base.h:276-280).
int make_plan(const Config &cfg, ChainPlan &out);

const double *half_band_coefs(int num_coefs); // 8..13, rate/rate_filters_generic.h:31-70

} // namespace rsmp
