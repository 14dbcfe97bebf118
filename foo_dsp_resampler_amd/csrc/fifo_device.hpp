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


// `len` consecutive samples of ONE channel starting at absolute index a0, when they lie contiguously in one buffer:
// kind 1 = float32 frames (element i at p32[i * stride32]), kind 2 = the channel's fp64 ring, kind 0 = split (fifo_get).
struct ChanSpan {
  int kind;
  const float *p32;
  long long stride32;
  const double *p64;
};
__device__ __forceinline__ ChanSpan chan_span(const AnyView &v, int c, long long a0, long long len)
{
  ChanSpan r = {0, nullptr, 1, nullptr};
  if (v.is_f32) {
    const int s = c / v.f.nch, ch = c - s * v.f.nch;
    r.stride32 = v.f.nch;
    if (v.f.ext && a0 >= v.f.ext_begin && a0 + len <= v.f.ext_end) {
      r.kind = 1;
      r.p32 = v.f.ext + s * v.f.ext_stream_stride + (a0 - v.f.ext_begin) * v.f.nch + ch;
    } else if ((!v.f.ext || a0 + len <= v.f.ext_begin || a0 >= v.f.ext_end) && a0 >= 0 && (a0 & v.f.ring_mask) + len <= v.f.ring_mask + 1) {
      r.kind = 1;
      r.p32 = v.f.ring + s * v.f.ring_stream_stride + (a0 & v.f.ring_mask) * v.f.nch + ch;
    }
  } else if (a0 >= 0 && (a0 & v.d.mask) + len <= v.d.mask + 1) {
    r.kind = 2;
    r.p64 = v.d.ring + (long long)c * v.d.chan_stride + (a0 & v.d.mask);
  }
  return r;
}

// Channel pair -> channels.  nchs = 0: pairs run over all C channels of the handle (2p, 2p+1; the last one may be single).
// nchs > 0 (batch handles with an odd channel count per stream): pairs never straddle two streams -- every stream has
// (nchs + 1) / 2 of them and its last channel rides alone with a zero imaginary part, exactly as in a one-stream handle,
// so the bits a stream gets do not depend on its neighbours in the batch.
struct PairCh {
  int ca, cb;
  bool hasb;
};
__device__ __forceinline__ PairCh pair_channels(int pair, int C, int nchs, unsigned pps_magic)
{
  // One branch-free, division-free, wave-uniform form for both cases: with nchs = 0 the whole handle counts as one
  // "stream" of C channels.  pps_magic = ceil(2^32 / pairs per stream) (pair_magic; 0 = one pair per stream), exact for pair < 2^32 / pps.
  const int ncs = nchs > 0 ? nchs : C, pps = (ncs + 1) >> 1;
  const int strm = pps_magic ? (int)__umulhi((unsigned)pair, pps_magic) : pair /* one pair per stream */, pin = pair - strm * pps;
  PairCh r;
  r.ca = __builtin_amdgcn_readfirstlane(strm * ncs + 2 * pin);
  r.cb = r.ca + 1;
  r.hasb = __builtin_amdgcn_readfirstlane(2 * pin + 1 < ncs) != 0;
  return r;
}

// Direct addressing of `len` consecutive samples of channel pair (2*pair, 2*pair+1) starting at absolute index a0,
// when they lie contiguously in one buffer: kind 1 = float32 frames with the two channels side by side (one 8-byte
// word per sample), kind 2 = the two planar fp64 rings, kind 0 = not contiguous (use fifo_get / fifo_put).
struct PairSpan {
  int kind;
  float2 *p2;
  long long fstride; // float2 elements between consecutive frames
  double *pa, *pb;
  bool hasb;
  __device__ __forceinline__ void get(int i, double &x, double &y) const
  {
    if (kind == 1) {
      const float2 f = p2[i * fstride];
      x = (double)f.x;
      y = (double)f.y;
    } else {
      x = pa[i];
      y = hasb ? pb[i] : 0.0;
    }
  }
  __device__ __forceinline__ void put(int i, double x, double y) const
  {
    if (kind == 1) p2[i * fstride] = make_float2((float)x, (float)y);
    else {
      pa[i] = x;
      if (hasb) pb[i] = y;
    }
  }
};

// NPTS samples i0, i0 + istride, ... of a contiguous span (kind 1 or 2) into registers, every load issued before any is
// waited for.  (PairSpan::get in an unrolled loop keeps its kind / hasb tests per element, and the compiler then waits for
// each element's loads at the joins: 8-16 memory round trips in a row at the head of a workgroup.)
template <int NPTS, typename CT> __device__ __forceinline__ void span_load(const PairSpan &sp, int i0, int istride, CT (&dst)[NPTS])
{
  if (sp.kind == 1) {
#pragma unroll
    for (int s = 0; s < NPTS; ++s) {
      const float2 f = sp.p2[(i0 + s * istride) * sp.fstride];
      dst[s].x = (double)f.x;
      dst[s].y = (double)f.y;
    }
  } else { // (pair_span sets pb = pa for a one-channel pair: both loads are unconditional)
#pragma unroll
    for (int s = 0; s < NPTS; ++s) {
      dst[s].x = sp.pa[i0 + s * istride];
      dst[s].y = sp.pb[i0 + s * istride];
    }
    if (!sp.hasb) {
#pragma unroll
      for (int s = 0; s < NPTS; ++s) dst[s].y = 0.0;
    }
  }
}

// (__host__ too: the launchers that pick a lean kernel instance ask this very function, not a copy of its predicate)
__host__ __device__ __forceinline__ PairSpan pair_span(const AnyView &v, int pair, bool hasb, long long a0, long long len, int ca = -1)
{
  if (ca < 0) ca = 2 * pair; // (pair_channels: differs only with an odd channel count per stream, where float frames are never contiguous pairs)
  PairSpan r;
  r.kind = 0;
  r.p2 = nullptr;
  r.fstride = 1;
  r.pa = r.pb = nullptr;
  r.hasb = hasb;
  if (v.is_f32) {
    if (hasb && !(v.f.nch & 1)) {
      const int hp = v.f.nch >> 1, strm = pair / hp, pin = pair - strm * hp;
      float *p = nullptr;
      if (v.f.ext && a0 >= v.f.ext_begin && a0 + len <= v.f.ext_end)
        p = v.f.ext + strm * v.f.ext_stream_stride + (a0 - v.f.ext_begin) * v.f.nch + 2 * pin;
      else if ((!v.f.ext || a0 + len <= v.f.ext_begin || a0 >= v.f.ext_end) && (a0 & v.f.ring_mask) + len <= v.f.ring_mask + 1)
        p = v.f.ring + strm * v.f.ring_stream_stride + (a0 & v.f.ring_mask) * v.f.nch + 2 * pin;
      if (p && (reinterpret_cast<unsigned long long>(p) & 7) == 0) {
        r.kind = 1;
        r.p2 = reinterpret_cast<float2 *>(p);
        r.fstride = hp;
      }
    }
  } else if ((a0 & v.d.mask) + len <= v.d.mask + 1) {
    r.kind = 2;
    r.pa = v.d.ring + (long long)ca * v.d.chan_stride + (a0 & v.d.mask);
    r.pb = hasb ? r.pa + v.d.chan_stride : r.pa;
  }
  return r;
}

// Work item -> (block, channel pair) for the one-workgroup-per-(block, pair) kernels; `lin` = linear workgroup id.
//
// hp <= 1 (stereo or planar data): pair-fastest order.  Blocks are dealt round-robin over the 8 XCDs, so with
// npairs % 8 == 0 the consecutive blocks of one pair land on the same XCD and their input overlap is an L2 hit.
//
// hp >= 2 (interleaved float frames of 2*hp channels): the hp workgroups of one (block, stream) each touch 8 bytes of
// every frame, i.e. the SAME 128-byte lines.  Spread over the 8 XCDs (each with its own L2) every line would be fetched
// by several L2s and written back in pieces (measured on 8-channel frames: 4.6x the output bytes in WRITE_SIZE).  Here
// they get linear ids lin, lin + 8, ..., lin + 8*(hp-1): same XCD under the observed round-robin placement, dispatched
// together, so the pieces meet in one L2.  Placement is a speed matter only; any placement is correct.
__device__ __forceinline__ bool item_map(int lin, int nblocks, int npairs, int hp, int &bl, int &pair)
{
  if (hp <= 1) {
    bl = lin / npairs;
    pair = lin - bl * npairs;
    return bl < nblocks;
  }
  const int xcd = lin & 7, t = lin >> 3;
  const int slot = t / hp, within = t - slot * hp, g = slot * 8 + xcd;
  const int nstreams = npairs / hp;
  if (g >= nblocks * nstreams) return false;
  bl = g / nstreams;
  pair = (g - bl * nstreams) * hp + within;
  return true;
}
inline int item_grid(int nblocks, int npairs, int hp)
{
  if (hp <= 1) return nblocks * npairs;
  const int ngroups = nblocks * (npairs / hp);
  return (ngroups + 7) / 8 * 8 * hp;
}
// pairs per interleaved frame that share cache lines (0 when the grouping does not apply)
inline int frame_pairs(const AnyView &in, const AnyView &out, int C)
{
  const AnyView *v = in.is_f32 ? &in : out.is_f32 ? &out : nullptr;
  if (!v || (v->f.nch & 1) || v->f.nch < 4 || C % v->f.nch) return 0;
  return v->f.nch / 2;
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
