// Host-side mirror of the plugin's DSP object (foo_dsp_rate.{h,cpp} of the reference): the code that sits
// ABOVE the ratelib.h C ABI in foobar2000 -- chunk staging, LPC pre/post-extrapolation of track edges,
// pre-roll dropping, latency accounting -- restated over plain buffers so that the GPU engine can be driven
// exactly the way the plugin drives it.  It talks to the engine only through RR_*.
#pragma once
#include "../../include/ratelib.h"

#include <cstddef>
#include <deque>
#include <vector>

namespace rsmp {

// what the plugin keeps from its preset (dsp_config.h:71-111 RateConfig)
struct RateSettings {
  int out_rate = 48000; // > 0: Hz; -2/-5: x2/x4 up; -3/-4: /2 and /4 down (dsp_config.h:60-69)
  int quality = 0;      // RR_quality
  int allow_aliasing = 0;
  int passband10 = 950; // pass band in tenths of a percent
  int phase = 50;
  unsigned real_rate(unsigned in_rate) const; // dsp_config.h:80-95
  bool no_resample(unsigned in_rate) const { return int(in_rate) == out_rate; } // dsp_config.h:97-101
};

// stand-in for foobar2000's audio_chunk: interleaved float frames plus format
struct AudioChunk {
  std::vector<float> data;
  size_t frames = 0;
  unsigned channels = 0, sample_rate = 0, channel_config = 0;
};

class DspRate {
public:
  explicit DspRate(const RateSettings &s) : cfg_(s) {}
  ~DspRate() { close(); }

  // foo_dsp_rate.cpp:130-210.  Returns true when the chunk must be passed through untouched (no
  // resampling needed); otherwise the chunk is consumed and resampled audio is appended to `out`.
  bool on_chunk(const AudioChunk &chunk, std::deque<AudioChunk> &out);
  void on_endoftrack(std::deque<AudioChunk> &out) { flushwrite(out); }    // foo_dsp_rate.cpp:80
  void on_endofplayback(std::deque<AudioChunk> &out) { flushwrite(out); } // foo_dsp_rate.cpp:82
  void flush();                                                            // foo_dsp_rate.cpp:212-216
  double get_latency() const;                                              // foo_dsp_rate.cpp:315-322
  int last_error() const { return err_; }

private:
  void reinit(unsigned sample_rate, unsigned channels, unsigned channel_config); // foo_dsp_rate.cpp:84-121
  void close();                                                                  // foo_dsp_rate.cpp:123-128
  void flushwrite(std::deque<AudioChunk> &out);                                  // foo_dsp_rate.cpp:218-313
  void emit(std::deque<AudioChunk> &out, const float *frames, size_t n);
  void check(int rr_error);
  float *frame(std::vector<float> &buf, size_t f) { return buf.data() + f * ch_; }

  RateSettings cfg_;
  RR_handle *h_ = nullptr;
  int err_ = 0;
  size_t in_accum_ = 0, out_accum_ = 0; // for get_latency

  // staging: [lead: to_add_ frames | body: inbuf_ frames | tail: to_add_ frames]
  std::vector<float> stage_, outbuf_;
  size_t inbuf_ = 0, outcap_ = 0, prime_ = 0;
  unsigned out_rate_ = 0, in_rate_ = 0, ch_ = 0, chmask_ = 0;
  size_t to_add_ = 0, to_drop_ = 0; // extrapolated frames fed in / resampled frames discarded at each edge
  size_t held_ = 0, dropped_ = 0;
  bool primed_ = false; // the first inbuf_ frames were extrapolated backwards and pushed
};

} // namespace rsmp
