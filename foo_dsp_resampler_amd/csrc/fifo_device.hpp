// Device-side fifo accessors shared by the stage kernels (see kernels.hpp for the coordinate convention).
#pragma once
#include "kernels.hpp"

namespace rsmp {

struct AnyView {
  int is_f32;
  F32View f;
  F64View d;
};

struct ChanRef { // per-channel precomputed addressing
  int is_f32;
  // f32
  const float *ring32;
  const float *ext32;
  long long mask32, ext_begin, ext_end;
  int nch;
  // f64
  const double *ring64;
  long long mask64;
};

__device__ __forceinline__ ChanRef chan_ref(const AnyView &v, int c)
{
  ChanRef r;
  r.is_f32 = v.is_f32;
  if (v.is_f32) {
    const int s = c / v.f.nch, ch = c - s * v.f.nch;
    r.ring32 = v.f.ring + s * v.f.ring_stream_stride + ch;
    r.ext32 = v.f.ext ? v.f.ext + s * v.f.ext_stream_stride + ch : nullptr;
    r.mask32 = v.f.ring_mask;
    r.ext_begin = v.f.ext_begin;
    r.ext_end = v.f.ext_end;
    r.nch = v.f.nch;
    r.ring64 = nullptr;
    r.mask64 = 0;
  } else {
    r.ring64 = v.d.ring + (long long)c * v.d.chan_stride;
    r.mask64 = v.d.mask;
    r.ring32 = r.ext32 = nullptr;
    r.mask32 = r.ext_begin = r.ext_end = 0;
    r.nch = 1;
  }
  return r;
}

__device__ __forceinline__ double fifo_get(const ChanRef &r, long long a)
{
  if (r.is_f32) {
    if (r.ext32 && a >= r.ext_begin && a < r.ext_end) return (double)r.ext32[(a - r.ext_begin) * r.nch];
    return (double)r.ring32[(a & r.mask32) * r.nch];
  }
  return r.ring64[a & r.mask64];
}

__device__ __forceinline__ void fifo_put(const ChanRef &r, long long a, double v)
{
  if (r.is_f32) {
    if (r.ext32 && a >= r.ext_begin && a < r.ext_end) const_cast<float *>(r.ext32)[(a - r.ext_begin) * r.nch] = (float)v;
    else const_cast<float *>(r.ring32)[(a & r.mask32) * r.nch] = (float)v;
  } else
    const_cast<double *>(r.ring64)[a & r.mask64] = v;
}


inline AnyView make_view(bool is_f32, const F32View &f, const F64View &d)
{
  AnyView v;
  v.is_f32 = is_f32 ? 1 : 0;
  v.f = f;
  v.d = d;
  return v;
}

} // namespace rsmp
