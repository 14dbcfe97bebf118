// See plugin_host.hpp.
#include "plugin_host.hpp"

#include "lpc.hpp"

#include <algorithm>
#include <cstring>

namespace rsmp {

namespace {

unsigned gcd_u(unsigned a, unsigned b)
{
  if (!a || !b) return 0;
  while (b) {
    const unsigned t = a % b;
    a = b;
    b = t;
  }
  return a;
}

// util.h:38-48: lengths (in frames at each rate) of about 1/N s, both at most M
void edge_lengths(unsigned &r1, unsigned &r2, unsigned N, unsigned M)
{
  const unsigned v = gcd_u(r1, r2);
  if (!v) return;
  r1 /= v;
  r2 /= v;
  unsigned n = (v + N - 1) / N;
  const unsigned z = std::max(r1, r2);
  if (z * n > M) n = M / z;
  if (n < 1) n = 1;
  r1 *= n;
  r2 *= n;
}

unsigned pow2_at_least(unsigned n, unsigned start) // util.h:50-55
{
  unsigned p = start;
  while (p < n) p *= 2;
  return p;
}

} // namespace

unsigned RateSettings::real_rate(unsigned in_rate) const
{
  if (out_rate > 0) return unsigned(out_rate);
  switch (out_rate) {
    case -2: return in_rate * 2;
    case -5: return in_rate * 4;
    case -3: return in_rate / 2;
    case -4: return in_rate / 4;
  }
  return in_rate;
}

void DspRate::check(int e)
{ // chain.h:26-29 turns any RR error into an exception; here it is latched
  if (e && !err_) err_ = e;
}

void DspRate::reinit(unsigned sample_rate, unsigned channels, unsigned channel_config)
{
  out_rate_ = cfg_.real_rate(sample_rate);
  RR_config c;
  c.in_rate = sample_rate;
  c.out_rate = out_rate_;
  c.phase = double(cfg_.phase);
  c.bandwidth = double(cfg_.passband10) / 10.0;
  c.allow_aliasing = cfg_.allow_aliasing ? 1 : 0;
  c.quality = RR_quality(cfg_.quality);
  check(RR_open(&c, int(channels), &h_));

  ch_ = channels;
  chmask_ = channel_config;
  in_rate_ = sample_rate;
  in_accum_ = out_accum_ = 0;
  held_ = dropped_ = 0;
  primed_ = false;

  unsigned add = sample_rate, drop = out_rate_;
  edge_lengths(add, drop, 20, 8192u);
  to_add_ = add;
  to_drop_ = drop;
  inbuf_ = std::min(std::max(sample_rate / 10, 2048u), 65536u);
  prime_ = std::max<size_t>(std::min(std::max(sample_rate / 20, 1024u), 16384u), 2 * kLpcOrder + 1);
  outcap_ = std::min(pow2_at_least(out_rate_ / 10, 8192u), 65536u) + to_drop_;
  stage_.assign((to_add_ + inbuf_ + to_add_) * ch_, 0.f);
  outbuf_.assign(outcap_ * ch_, 0.f);
}

void DspRate::close()
{
  RR_close(&h_);
  in_accum_ = out_accum_ = 0;
}

void DspRate::flush()
{
  if (h_) close();
}

double DspRate::get_latency() const
{
  if (in_rate_ && out_rate_) return double(in_accum_) / double(in_rate_) - double(out_accum_) / double(out_rate_);
  return 0;
}

void DspRate::emit(std::deque<AudioChunk> &out, const float *frames, size_t n)
{
  AudioChunk c;
  c.data.assign(frames, frames + n * ch_);
  c.frames = n;
  c.channels = ch_;
  c.sample_rate = out_rate_;
  c.channel_config = chmask_;
  out.push_back(std::move(c));
}

bool DspRate::on_chunk(const AudioChunk &chunk, std::deque<AudioChunk> &out)
{
  if (!h_) {
    if (cfg_.no_resample(chunk.sample_rate)) return true;
    reinit(chunk.sample_rate, chunk.channels, chunk.channel_config);
  } else if (ch_ != chunk.channels || chmask_ != chunk.channel_config || in_rate_ != chunk.sample_rate) {
    flushwrite(out); // old format is finished; the handle is closed afterwards
    if (cfg_.no_resample(chunk.sample_rate)) return true;
    reinit(chunk.sample_rate, chunk.channels, chunk.channel_config);
  }
  if (!h_) return false;

  const float *cur = chunk.data.data();
  size_t left = chunk.frames;
  float *body = frame(stage_, to_add_);
  size_t got = 0;
  do {
    if (!primed_) { // collect the first inbuf_ frames, then extrapolate backwards in front of them
      const size_t take = std::min(left, inbuf_ - held_);
      std::memcpy(body + held_ * ch_, cur, take * ch_ * sizeof(float));
      held_ += take;
      cur += take * ch_;
      left -= take;
      in_accum_ += take;
      if (held_ == inbuf_) {
        lpc_extrapolate_backward(body, prime_, int(ch_), kLpcOrder, to_add_);
        primed_ = true;
        check(RR_push(h_, stage_.data(), to_add_ + inbuf_));
      }
    }
    if (primed_ && left) { // keep the most recent inbuf_ frames around for the end-of-track extrapolation
      if (left < inbuf_) {
        std::memmove(body, body + left * ch_, (inbuf_ - left) * ch_ * sizeof(float));
        std::memcpy(body + (inbuf_ - left) * ch_, cur, left * ch_ * sizeof(float));
      } else
        std::memcpy(body, cur + (left - inbuf_) * ch_, inbuf_ * ch_ * sizeof(float));
      check(RR_push(h_, cur, left));
      cur += left * ch_;
      in_accum_ += left;
      left = 0;
    }
    got = 0;
    check(RR_pull(h_, outbuf_.data(), outcap_, &got));
    size_t skip = to_drop_ - dropped_; // the resampled image of the backward extrapolation
    if (skip) {
      skip = std::min(skip, got);
      got -= skip;
      dropped_ += skip;
    }
    if (got) {
      out_accum_ += got;
      emit(out, outbuf_.data() + skip * ch_, got);
    }
  } while (left || got);

  while (in_accum_ > in_rate_ && out_accum_ > out_rate_) {
    in_accum_ -= in_rate_;
    out_accum_ -= out_rate_;
  }
  return false;
}

// End of stream, three situations (foo_dsp_rate.cpp:218-313):
//  (a) too little audio to fit an LPC model: resample what there is, no extrapolation;
//  (b) less than one staging buffer was seen: extrapolate both edges, drop both images;
//  (c) normal: extrapolate forward from the retained tail, drop the image at the end.
void DspRate::flushwrite(std::deque<AudioChunk> &out)
{
  if (!h_) return;
  float *body = frame(stage_, to_add_);
  size_t got = 0;

  if (!primed_ && !(held_ > 2 * size_t(kLpcOrder))) { // (a)
    check(RR_push(h_, body, held_));
    check(RR_drain(h_));
    for (;;) {
      check(RR_pull(h_, outbuf_.data(), outcap_, &got));
      if (!got) break;
      out_accum_ += got;
      emit(out, outbuf_.data(), got);
    }
    close();
    return;
  }

  bool drop_lead = false;
  if (!primed_) { // (b)
    drop_lead = true;
    const size_t prime = std::min(held_, prime_);
    lpc_extrapolate_backward(body, prime, int(ch_), kLpcOrder, to_add_);
    lpc_extrapolate_forward(body, held_, prime, int(ch_), kLpcOrder, to_add_);
    primed_ = true;
    check(RR_push(h_, stage_.data(), to_add_ + held_ + to_add_));
    check(RR_drain(h_));
    dropped_ = 0;
  } else { // (c)
    lpc_extrapolate_forward(body, inbuf_, prime_, int(ch_), kLpcOrder, to_add_);
    check(RR_push(h_, body + inbuf_ * ch_, to_add_));
    check(RR_drain(h_));
  }

  // emit everything except the first (case b only) and the last to_drop_ resampled frames
  size_t pending = 0; // frames parked at the start of outbuf_
  for (;;) {
    check(RR_pull(h_, outbuf_.data() + pending * ch_, outcap_ - pending, &got));
    if (!got) break;
    size_t skip = drop_lead ? std::min(to_drop_ - dropped_, got) : 0;
    if (skip) { // leading image; case (c) never drops here (foo_dsp_rate.cpp:289-310)
      got -= skip;
      dropped_ += skip;
      std::memmove(outbuf_.data() + pending * ch_, outbuf_.data() + (pending + skip) * ch_, got * ch_ * sizeof(float));
    }
    pending += got;
    const size_t ready = pending - std::min(pending, to_drop_);
    if (ready) {
      out_accum_ += ready;
      emit(out, outbuf_.data(), ready);
      pending -= ready;
      std::memmove(outbuf_.data(), outbuf_.data() + ready * ch_, pending * ch_ * sizeof(float));
    }
  }
  close();
}

} // namespace rsmp
