// Linear-prediction edge extrapolation (host side, float32), used by the plugin layer to pre-roll /
// post-roll the resampler so that track edges do not ring.  Behaviour of lpc/lpc.{h,cpp} of the reference:
// per channel, Welch window -> autocorrelation -> Levinson-Durbin (order <= 32) -> 0.999^k damping ->
// forward / backward prediction clamped to +-10.  north_star: "lpc/ extrapolation stays on the host".
#pragma once
#include <cstddef>

namespace rsmp {

constexpr int kLpcOrder = 32; // lpc/lpc.h:25

// `data` points at frame 0 of `data_len` interleaved frames (nch channels).  Writes `extra_bkwd` frames
// before data[0] and `extra_fwd` frames after data[data_len*nch); the caller owns that space
// (lpc/lpc.h:4-22, lpc.cpp:25-68).
void lpc_extrapolate(float *data, size_t data_len, int nch, int order, size_t extra_bkwd, size_t extra_fwd);

// lpc/lpc.h:29-38: the model is always fitted on `prime_len` frames at the respective edge
inline void lpc_extrapolate_backward(float *data, size_t prime_len, int nch, int order, size_t extra)
{
  lpc_extrapolate(data, prime_len, nch, order, extra, 0);
}
inline void lpc_extrapolate_forward(float *data, size_t data_len, size_t prime_len, int nch, int order, size_t extra)
{
  lpc_extrapolate(data + (data_len - prime_len) * nch, prime_len, nch, order, 0, extra);
}

} // namespace rsmp
