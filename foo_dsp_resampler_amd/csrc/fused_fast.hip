// Lean form of the headline kernel: dft(xL, L in {1,2}, N = 4096) -> vpoly0 on the fp64 matrix pipe, for the blocks whose
// input and output are plain interleaved float frames in one buffer each (what a device-resident push / flow of whole
// blocks looks like; rate/dft_filter.h:60-190 followed by rate/rate_filters_generic.h:272-305).
//
// Same mathematics, same summation order and the same two-round LDS sample image as fused_kernel<.., true> (fused.hip);
// what differs is the bookkeeping around it:
//   * no generic fifo addressing: the host hands over one base pointer per side (FastIo) and launches this kernel only
//     for block ranges it has checked (fused_fast_range); other blocks go to fused_kernel;
//   * the polyphase stage walks TILES (16 output residues x 4 periods = one accumulator pair) in group-major order, each
//     wave a contiguous range of them: the coefficient tile changes only when the residue group does (double-buffered,
//     prefetched a group ahead), the window start of a group comes from a host table instead of an integer division
//     per lane, every store is issued straight after its tile under a per-lane range test (no deferred-store state,
//     no per-step validity masks, no skip loops);
//   * the next tile's samples are always prefetched (clamped address), so the wait before a tile's MFMA chain is for
//     reads issued one whole tile earlier.
// The old kernel spent ~1200 integer / 1650 scalar instructions per wave and had 1000+ SGPR spill moves in its code;
// this one keeps the scalar state of the tile loop in a dozen registers.
#include "fft_device.hpp"
#include "fifo_device.hpp"
#include "kernels.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstdio>

#ifndef RSMP_FAST_FWD8
#define RSMP_FAST_FWD8 1
#endif
// wave priority by phase: 1 = FFT phases above the polyphase phase, 2 = the other way round (experiments)
// twiddles multiplied up from w^1, w^2, w^4, w^8 instead of loaded (fft_device.hpp, TWGEN)
#ifndef RSMP_TWGEN
#define RSMP_TWGEN 1
#endif
// output stores as raw buffer stores with the range check done by the buffer descriptor (no per-tile compare / branch)
#ifndef RSMP_BUFSTORE
#define RSMP_BUFSTORE 1
#endif
// RSMP_EXP_SKIP (same rules): phase ablations of the lean kernel for the cycle budget of DESIGN.md 5a -- bit 0 no polyphase
// rounds, 1 no inverse transform, 2 no forward transform, 3 no output stores, 4 no polyphase LDS reads (first tile's samples
// reused), 5 no MFMAs (loads and stores stay), 6 no sample image / seam writes, 7 no G loads, 8 no input loads
// RSMP_EXP_TAB / RSMP_EXP_LINEAR / RSMP_EXP_HALFMFMA (knobs.hpp; -DRSMP_EXPERIMENTS builds only, WRONG results): bit 0 = every
// coefficient tile is group 0's, bit 1 = every G value is slot 0's (the loads stay, their L1 misses go: upper bounds of what
// smaller / shared tables could buy); lane-linear window reads; 6 of every 14 MFMAs removed
// forward transform (FWD8 form) on twiddles loaded next to the block's input, G issued two passes before it is needed
#ifndef RSMP_FWD_PRETW
#define RSMP_FWD_PRETW 1
#endif
#ifndef RSMP_PRIO
#define RSMP_PRIO 0
#endif
#ifndef RSMP_PFW
#define RSMP_PFW 15
#endif
#ifndef RSMP_PFI
#define RSMP_PFI 8
#endif

namespace rsmp {

namespace {
// G is read once per workgroup (64 KB, twice the vector L1): RSMP_G_NT = 1 loads it non-temporally so that it does not evict
// the twiddle rows and coefficient tiles the workgroups of a CU share
#ifndef RSMP_G_NT
#define RSMP_G_NT 0
#endif
typedef double rsmp_d2v __attribute__((ext_vector_type(2)));
typedef unsigned int rsmp_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 load_g(const double2 *p)
{
#if RSMP_G_NT
  const rsmp_d2v q = __builtin_nontemporal_load(reinterpret_cast<const rsmp_d2v *>(p));
  return make_double2(q.x, q.y);
#else
  return *p;
#endif
}
constexpr int kPad = 32;
constexpr int kSA = kFusedSA, kSB0 = kFusedSB0;
} // namespace

// OUT64: the outputs go to the planar fp64 ring of the next stage's fifo (absolute index & mask: a ring wrap costs nothing)
// instead of interleaved float frames -- chains like 44.1k->192k, whose x2 -> 80/147 pair feeds an x4 stage.
//
// SPLIT (LOG2P = 12; kernel names fused_split_kernel<KS, OMODE> / fused_split2_kernel<KS, OMODE>): the sub-blocked form for x2
// stages whose blocks (8192 ... 32768 points) do not fit a workgroup.
// A block of the reference is nsub workgroups; each computes `len` of the block's valid samples from a 4096-point window of
// the block's inputs: y[2m + r] = IDFT_4096(DFT_4096(x) * G_r)[m], r = 0, 1 -- the block's two polyphase components, the same
// linear convolution as the reference's one long transform, in a different fp64 summation order.  One forward and two inverse
// transforms; component 0 waits in registers while component 1 is computed (256-VGPR budget: two workgroups per CU), then
// both go to ONE LDS image of the whole sub-block and the polyphase stage runs as a single round.  B counts sub-blocks, so the
// seam ring, the block table and seam_kernel work on sub-blocks exactly as they do on blocks.
// OGEN (sub-blocked form only): float frames out, each output wherever the output fifo has it -- the caller's buffer or the
// fifo's ring (RR_push without a destination, outputs beyond the caller's capacity): fifo_put for a channel pair.
// TWO (sub-blocked form, 8192-point blocks): the block is ONE workgroup -- its V samples fit a pair of 4096-point component
// transforms whole -- and the polyphase stage runs in two rounds, the second from the register slots kept across the first.
template <int LOG2P, int KS, bool OUT64, bool SPLIT, bool OGEN, bool TWO = false>
__device__ __forceinline__ void fused_fast_body(const FusedArgs &a, const FastIo &io)
{
  static_assert(!TWO || SPLIT, "two rounds from registers: sub-blocked form only");
  static_assert(!SPLIT || LOG2P == 12, "sub-blocked form: 4096-point components");
  static_assert(!OGEN || (SPLIT && !OUT64), "generic float output exists for the sub-blocked form only");
  constexpr int LOG2N = 12, N = 1 << LOG2N, P = 1 << LOG2P;
  constexpr int T = N / 16, TF = P / 16;
  extern __shared__ __attribute__((aligned(16))) double lds[];

  const int tid = threadIdx.x;
  const int npairs = a.d.C >> 1; // even channel count (host-checked)
  int bl, pair;
  if (!item_map(blockIdx.x, a.d.nblocks, npairs, a.d.hp, bl, pair)) return; // uniform
  const long long B = a.d.B0 + bl;
  const int hp = io.nch >> 1, strm = pair / hp, pin = pair - strm * hp;
  SubBlock sb = {0, 0, 0, 0};
  long long e0_split = 0;
  if constexpr (SPLIT) { // (the launch starts at sub-block 0 of reference block Bref0)
    const int kr = bl / a.d.nsub;
    sb = sub_block(bl - kr * a.d.nsub, a.d.V, a.d.Vs, a.d.Pref);
    e0_split = (a.d.Bref0 + kr) * a.d.q + sb.win;
  }
#if defined(RSMP_EXPERIMENTS) && defined(RSMP_VCONST) // what a block length known at compile time would buy (553-tap filters: 3544)
  const int V = RSMP_VCONST;
#else
  const int V = SPLIT ? sb.len : a.d.V;
#endif
  const bool fwd_active = tid < TF;
  const double2 *__restrict__ Gp = a.d.G;
  double2 *smp = reinterpret_cast<double2 *>(lds) + kPad; // smp[n] = (channel A, channel B) sample n of the block
  const int nm1 = a.n - 1;

  if (RSMP_PRIO == 1) __builtin_amdgcn_s_setprio(3);
  // Builds with -DRSMP_STAMPS_BUILD (tools/build_variant.sh) + RSMP_STAMPS=1 in the environment: per-phase cycle sums of wave 0
  // (s_memtime) in one workgroup of 64, printed when the handle closes.  The product build has no trace of it: the stamp
  // points were branches (and scheduling barriers) in every workgroup.
#ifdef RSMP_STAMPS_BUILD
  const bool stamping = a.stamps && (blockIdx.x & 63) == 5;
  unsigned long long tstamp = stamping ? __builtin_readcyclecounter() : 0;
#define RSMP_STAMP(slot)                                          \
  if (stamping) {                                                  \
    const unsigned long long now = __builtin_readcyclecounter();  \
    if (tid == 0) atomicAdd(a.stamps + (slot), now - tstamp);      \
    tstamp = now;                                                  \
  }
#else
#define RSMP_STAMP(slot)
#endif

  // ---------------------------------------------------------------- load the block (fp32 -> fp64)
  // L = 2 (FWD8): the forward transform has half the points of the inverse one and runs 8 points per thread on ALL
  // waves (fft8_regs); a thread then already holds the 8 distinct spectrum values Zp[tid + (s & 7) * T] its 16
  // inverse-transform inputs need, so the replication exchange disappears.
  constexpr bool FWD8 = (LOG2P == LOG2N - 1) && RSMP_FAST_FWD8;
  constexpr int NLD = FWD8 ? 8 : 16, TL = FWD8 ? T : TF; // points per loading thread, loading threads
  c64 v[16];
  c64 u8[8];
  c64 z0[SPLIT ? 16 : 1]; // sub-blocked form: component 0 of the block (samples 2m), component 1 ends up in `v`
  constexpr bool PRETW = FWD8 && RSMP_FWD_PRETW;
  double2 wf[PRETW ? fft8_tw_regs(LOG2P) : 1];
  if constexpr (PRETW) fft8_tw_load<LOG2P>(wf, tid, a.d.tw_fwd8); // in flight together with the input loads below
  {
    const long long e0 = SPLIT ? e0_split : B * a.d.q;
    const bool ld_active = (FWD8 || fwd_active) && !(RSMP_EXP_SKIP & 256);
    if (RSMP_EXP_SKIP & 256) {
#pragma unroll
      for (int s = 0; s < NLD; ++s) (FWD8 ? u8[s & 7] : v[s]) = {1e-3 * tid, 1e-3 * s};
    }
    if (SPLIT && io.in_unaligned) { // (uniform) a caller's buffer that is only 4-byte aligned: the sub-blocked form has no generic
      // kernel to hand such blocks to, so it reads the two channels separately
      const float *pe = io.in + strm * io.in_stream_stride + 2 * pin, *pr = io.in_ring + strm * io.in_ring_stream_stride + 2 * pin;
#pragma unroll
      for (int s = 0; s < NLD; ++s) {
        const long long e = e0 + tid + s * TL;
        const float *f = e >= io.in_abs0 ? pe + (e - io.in_abs0) * io.nch : pr + (e & io.in_ring_mask) * io.nch;
        v[s] = {(double)f[0], (double)f[1]};
      }
    } else
    if (e0 >= io.in_abs0) { // uniform: the whole block lies in the caller's buffer
      const float2 *p2 = reinterpret_cast<const float2 *>(io.in + strm * io.in_stream_stride + (e0 - io.in_abs0) * io.nch + 2 * pin);
      if (ld_active) {
#pragma unroll
        for (int s = 0; s < NLD; ++s) {
          const float2 f = p2[(unsigned)((tid + s * TL) * hp)]; // (unsigned: scalar base + 32-bit lane offset, no 64-bit address per load)
          (FWD8 ? u8[s & 7] : v[s]) = {(double)f.x, (double)f.y};
        }
      }
    } else if (ld_active) { // first block of a push: its head is the previous push's tail, kept in fifo 0's ring
      const float *pe = io.in + strm * io.in_stream_stride + 2 * pin, *pr = io.in_ring + strm * io.in_ring_stream_stride + 2 * pin;
#pragma unroll
      for (int s = 0; s < NLD; ++s) {
        const long long e = e0 + tid + s * TL;
        const float2 f = *reinterpret_cast<const float2 *>(e >= io.in_abs0 ? pe + (e - io.in_abs0) * io.nch : pr + (e & io.in_ring_mask) * io.nch);
        (FWD8 ? u8[s & 7] : v[s]) = {(double)f.x, (double)f.y};
      }
    }
  }
  RSMP_STAMP(0)
  // ---------------------------------------------------------------- FFT-FIR (as fused_kernel)
  if constexpr (FWD8) {
    double2 g[16];
    if constexpr ((RSMP_EXP_SKIP & 4) != 0) {
#pragma unroll
      for (int s = 0; s < 16; ++s) g[s] = (RSMP_EXP_SKIP & 128) ? make_double2(1e-3 * s, 1e-4 * tid) : load_g(Gp + tid + s * T);
      if constexpr (PRETW) {
#pragma unroll
        for (int i = 0; i < fft8_tw_regs(LOG2P); ++i) u8[i & 7].x += wf[i].x * 1e-30; // keep the loads alive
      }
    } else if constexpr (PRETW) {
      fft8_regs_pre<LOG2P, -1>(u8, tid, wf, lds, [&](int p) {
        if (p == 1) { // two passes (two LDS round trips) ahead of the multiplication
#pragma unroll
          for (int s = 0; s < 16; ++s) g[s] = (RSMP_EXP_SKIP & 128) ? make_double2(1e-3 * s, 1e-4 * tid) : load_g(Gp + (unsigned)(tid + (RSMP_EXP_TAB & 2 ? 0 : s * T)));
        }
      });
    } else { // G in flight during the whole forward transform
#pragma unroll
      for (int s = 0; s < 16; ++s) g[s] = load_g(Gp + tid + (RSMP_EXP_TAB & 2 ? 0 : s * T));
      fft8_regs<LOG2P, -1, RSMP_TWGEN != 0>(u8, tid, a.d.tw_fwd8, lds);
    }
    RSMP_STAMP(1)
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = cmul(u8[s & 7], c64{g[s].x, g[s].y});
    __syncthreads(); // the inverse transform's exchange reuses the LDS the forward one just read
  } else {
  fft_regs<LOG2P, -1, LOG2P == LOG2N ? 2 : 0, RSMP_PFW, RSMP_TWGEN != 0>(v, tid, fwd_active, a.d.tw_fwd, lds);
  RSMP_STAMP(1)
  if constexpr (LOG2P < LOG2N) {
    double2 g[16]; // issued before the exchange so the L2 latency overlaps it
#pragma unroll
    for (int s = 0; s < 16; ++s) g[s] = load_g(Gp + tid + s * T);
    double2 *l2 = reinterpret_cast<double2 *>(lds);
    // Z[tid + s*T] = Zp[tid + (s & 7) * T]: a forward thread (tid < TF) already holds those in its even slots; its odd
    // slots are what thread tid + TF needs
    if (fwd_active) {
#pragma unroll
      for (int u = 0; u < 8; ++u) l2[tid + u * TF] = make_double2(v[2 * u + 1].x, v[2 * u + 1].y);
    }
    __syncthreads();
    c64 z[8];
    if (fwd_active) {
#pragma unroll
      for (int u = 0; u < 8; ++u) z[u] = v[2 * u];
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double2 q = l2[tid - TF + u * TF];
        z[u] = {q.x, q.y};
      }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) v[s] = cmul(z[s & 7], c64{g[s].x, g[s].y});
    __syncthreads();
  } else if constexpr (SPLIT) {
    // X = DFT_4096 of the window stays in `xs` for both components; component 0's samples wait in `z0`
    c64 xs[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      xs[s] = v[s];
      const double2 g = load_g(Gp + tid + s * T);
      v[s] = cmul(xs[s], c64{g.x, g.y});
    }
    fft_regs<LOG2N, +1, 2, RSMP_PFI, RSMP_TWGEN != 0>(v, tid, true, a.d.tw_inv, lds);
    double2 g1[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) g1[s] = load_g(Gp + N + tid + s * T);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      z0[s] = v[s];
      v[s] = cmul(xs[s], c64{g1[s].x, g1[s].y});
    }
    __syncthreads(); // the second inverse transform's exchange reuses the LDS the first one just read
    fft_regs<LOG2N, +1, 2, RSMP_PFI, RSMP_TWGEN != 0>(v, tid, true, a.d.tw_inv, lds);
  } else {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const double2 g = load_g(Gp + tid + s * T);
      v[s] = cmul(v[s], c64{g.x, g.y});
    }
  }
  }
  RSMP_STAMP(2)
  if constexpr (!(RSMP_EXP_SKIP & 2) && !SPLIT) fft_regs<LOG2N, +1, 2, RSMP_PFI, RSMP_TWGEN != 0>(v, tid, true, a.d.tw_inv, lds);
  RSMP_STAMP(3)

  // ---------------------------------------------------------------- stage-1 samples -> LDS (round A) and seam ring
  const int ca = 2 * pair, cb = ca + 1;
  {
    const int slot = (int)(B & a.seam_mask);
    double *seamA = a.seam + ((long long)(slot * (a.d.C + 1) + ca) * 2) * 32;
    double *seamB = a.seam + ((long long)(slot * (a.d.C + 1) + cb) * 2) * 32;
    // The head of the block goes to the seam ring from registers (slot 0); the tail is picked up from the LDS image below
    // (store_tail), where it is 23 consecutive elements -- out of registers it took per-lane range tests and a pair of
    // conditional global stores in every one of the 16 unrolled slots (V is a run-time value): ~300 scalar and ~150 vector
    // instructions per wave for 46 doubles.
    if constexpr (SPLIT && TWO) {
      __syncthreads(); // the image overlays the exchange area of the last transform
      // first image: samples [0, kSplitRaEnd) + 32 = slots 0 .. kSplitSA - 1 of both components and 16 elements of slot kSplitSA
#pragma unroll
      for (int s = 0; s <= kSplitSA; ++s) {
        const int n = 2 * (tid + s * T);
        if (n < V && (s < kSplitSA || tid < kPad / 2)) {
          smp[n] = make_double2(z0[s].x, z0[s].y);
          smp[n + 1] = make_double2(v[s].x, v[s].y);
        }
      }
      if (tid < kPad) {
        smp[tid - kPad] = make_double2(0.0, 0.0);
        if (V < kSplitRaEnd + kPad) smp[V + tid] = make_double2(0.0, 0.0);
      }
      (void)seamA; (void)seamB;
    } else if constexpr (SPLIT) {
      __syncthreads(); // the image overlays the exchange area of the last transform
      // element m of component r is sample 2 (m - shift) + r of the sub-block; the whole sub-block fits the image
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int n = 2 * (tid + s * T - sb.shift);
        if (n >= 0 && n < V) { // (V is even)
          smp[n] = make_double2(z0[s].x, z0[s].y);
          smp[n + 1] = make_double2(v[s].x, v[s].y);
        }
      }
      if (tid < kPad) {
        smp[tid - kPad] = make_double2(0.0, 0.0);
        smp[V + tid] = make_double2(0.0, 0.0);
      }
      (void)seamA; (void)seamB;
    } else {
#pragma unroll
    for (int s = 0; s <= kSA; ++s) {
      const int n = tid + s * T;
      const bool whole = (s + 1) * T <= V;                 // every sample of this slot is valid
      if (!(RSMP_EXP_SKIP & 64)) {
        if (s < kSA ? (whole || n < V) : (tid < kPad && n < V)) smp[n] = make_double2(v[s].x, v[s].y);
      }
    }
    if (tid < nm1 && !(RSMP_EXP_SKIP & 64)) {
      seamA[tid] = v[0].x;
      seamB[tid] = v[0].y;
    }
    if (tid < kPad) { // finite guard values: padded coefficients are zero, 0 * x must stay 0
      smp[tid - kPad] = make_double2(0.0, 0.0);
      if (V < kSA * T + kPad) smp[V + tid] = make_double2(0.0, 0.0);
    }
    }
  }
  __syncthreads();
  if constexpr (SPLIT) { // the head goes to the seam ring from the image (its register slot depends on the sub-block's shift)
    if (tid < nm1) {
      const int slot = (int)(B & a.seam_mask);
      const double2 t = smp[tid];
      a.seam[((long long)(slot * (a.d.C + 1) + ca) * 2) * 32 + tid] = t.x;
      a.seam[((long long)(slot * (a.d.C + 1) + cb) * 2) * 32 + tid] = t.y;
    }
  }
  // the block's last n-1 samples -> seam ring, read back from whichever LDS image holds them (element 0 of `img` = sample n0)
  const int tail0 = V - nm1;
  auto store_tail = [&](const double2 *img, int n0) {
    if (tid < nm1 && !(RSMP_EXP_SKIP & 64)) {
      const int slot = (int)(B & a.seam_mask);
      const double2 t = img[tail0 + tid - n0];
      a.seam[((long long)(slot * (a.d.C + 1) + ca) * 2 + 1) * 32 + tid] = t.x;
      a.seam[((long long)(slot * (a.d.C + 1) + cb) * 2 + 1) * 32 + tid] = t.y;
    }
  };
  // uniform: else the whole tail lies inside the first image (V <= kSB0 * T + n - 1)
  const bool tail_in_b = TWO ? tail0 >= kSplitRbStart : (!SPLIT && tail0 >= kSB0 * T);
  if (!tail_in_b) store_tail(smp, 0);
  RSMP_STAMP(4)

  // ---------------------------------------------------------------- polyphase FIR on v_mfma_f64_4x4x4, tile by tile
  // Lane maps (tools/probe_mfma4.hip): A lane = 16k + 4b + i, B lane = 16k + 4b + j, D lane = 16i + 4b + j:
  // as an A/B lane this lane is (k = hi, block bq, i or j = jq); its D element is (row hi, block bq, period jq).
  const FusedBlock fb = a.blk[bl];
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 4, bq = (lane >> 2) & 3, jq = lane & 3;
  const int pl = a.polyL, step = a.step;
  const int irel_hi = fb.irel_lo + fb.cnt;
  const int frame_bytes = io.nch * 4;
  char *const obytes = (OUT64 || OGEN) ? nullptr
                            : reinterpret_cast<char *>(io.out + strm * io.out_stream_stride + (a.out_offset2 + fb.i_lo - io.out_abs0) * io.nch + 2 * pin);
  // OGEN: absolute index (in the output fifo) of the block's output 0, this pair's slot of a frame in either buffer
  const long long oabs0 = a.out_offset2 + fb.i_lo;
  float *const oext = OGEN ? io.out + strm * io.out_stream_stride + 2 * pin : nullptr;
  float *const oring = OGEN ? io.out_ring + strm * io.out_ring_stream_stride + 2 * pin : nullptr;
  // OUT64: one descriptor per channel over its whole ring; output ib of the block sits at ring slot (o64 + ib) & mask
  const unsigned o64 = OUT64 ? (unsigned)((a.out_offset2 + fb.i_lo) & io.out64_mask) : 0u;
  const unsigned m64 = (unsigned)io.out64_mask;
  __amdgpu_buffer_rsrc_t orsrcA, orsrcB;
  if constexpr (OUT64) {
    double *ra = io.out64 + (long long)(2 * pair) * io.out64_chan_stride;
    orsrcA = __builtin_amdgcn_make_buffer_rsrc(ra, 0, (int)((io.out64_mask + 1) * 8), 0x00020000);
    orsrcB = __builtin_amdgcn_make_buffer_rsrc(ra + io.out64_chan_stride, 0, (int)((io.out64_mask + 1) * 8), 0x00020000);
  }
  const int rloc = 4 * bq + hi; // this lane's output residue within a 16-residue group
  const int ngrp = a.NGRP;
  const double2 *const cfm_lane = a.cfm2 + lane;
  const int *const qtab_lane = a.qtab + bq;

  auto poly_round = [&](int kb, int ke, const double2 *xs, int li_lo, int li_hi) {
    const int ncs = (ke - kb + 3) >> 2; // column steps (4 periods each) of this round
    const int nt = ngrp * ncs;
    const int t0 = (nt * wave) >> 2, t1 = (nt * (wave + 1)) >> 2; // this wave's tiles, group-major
    if (t0 >= t1) return;
    int g = t0 / ncs, c = t0 - g * ncs;
    const int hi_bound = min(irel_hi, ke * pl);
    const int lane_li = fb.base_li + hi + (kb + jq) * step; // window start = lane_li + q(group, block) + c * 4 * step
    const int lane_ib = (kb + jq) * pl + rloc - fb.irel_lo; // output index relative to i_lo = lane_ib + 16 g + c * 4 * pl
    const int cnt = hi_bound - fb.irel_lo;
#if RSMP_BUFSTORE
    // raw buffer over this round's outputs [0, cnt) of the block: frame ib at byte ib * frame_bytes, 8 bytes of it are ours
    const __amdgpu_buffer_rsrc_t orsrc =
        __builtin_amdgcn_make_buffer_rsrc(obytes, 0, (!OUT64 && !OGEN && cnt > 0) ? (cnt - 1) * frame_bytes + 8 : 0, 0x00020000);
#endif
    const int step4 = 4 * step, pl4 = 4 * pl;
    // store offset of a tile = (this lane's part, once per round) + (the tile's part, scalar): one vector add per tile.  Only the
    // last residue group can hold residues >= polyL (polyL not a multiple of 16): its lanes get the dropped offset there.
    const int lane_off0 = __mul24(lane_ib, frame_bytes); // |lane_ib| < 2^23
    const bool lane_dead_last = 16 * (ngrp - 1) + rloc >= pl;

    // Coefficient tiles: current group, next group (in flight).  The window start of a group comes with its tile (odd KS: spare
    // half of the last 16-byte element) and stays a double until the group becomes current, so that nothing waits for the
    // tile load at the point where it is issued.
    // (Measured and NOT kept, round 3: two register sets used alternately group by group, all four (set, sample buffer)
    // combinations spelled out so that no set is ever copied and no s_waitcnt vmcnt(0) sits behind the prefetch -- 168 VGPRs
    // with 19 spilled across round A, 2.37 against 1.96 ms.  The kernel has no registers to spare at three workgroups per CU.)
    double cc[KS], cn[KS];
    double qcd = 0.0, qnd = 0.0;
    int qci = 0, qni = 0;
    auto load_tile = [&](int gg, double (&c_)[KS], double &qd_, int &qi_) {
      constexpr int KSP = (KS + 1) / 2;
      const double2 *cp = cfm_lane + (RSMP_EXP_TAB & 1 ? 0 : gg) * (KSP * 64); // uniform offset; two k-steps per 16-byte load
#pragma unroll
      for (int s2 = 0; s2 < KSP; ++s2) {
        const double2 d = cp[s2 * 64];
        c_[2 * s2] = d.x;
        if (2 * s2 + 1 < KS) c_[2 * s2 + 1] = d.y;
        else qd_ = d.y;
      }
      if (!(KS & 1)) qi_ = qtab_lane[gg * 4];
    };
    auto qof = [&](double qd_, int qi_) { return (KS & 1) ? (int)qd_ : qi_; };
    load_tile(g, cc, qcd, qci);
    load_tile(min(g + 1, ngrp - 1), cn, qnd, qni); // (unconditional, clamped: see the group switch below)

    double2 x0[KS], x1[KS];
    auto fill = [&](double2 (&x)[KS], int q, int cstep) {
#if RSMP_EXP_LINEAR // timing experiment only (WRONG results): lane-linear, conflict-free window addresses
      const int li = li_lo + 32 + lane + ((cstep * 64 + (q & 63)) & 1023);
#else
      const int li = max(li_lo, min(li_hi, lane_li + q + cstep * step4));
#endif
      const double2 *xp = xs + li;
#pragma unroll
      for (int s = 0; s < KS; ++s) x[s] = xp[4 * s];
    };
    int qc = qof(qcd, qci);
    fill(x0, qc, c);

    int left = t1 - t0;
    // one tile: prefetch the next tile's samples into `xn`, run the two accumulation chains on `xc`, store
    auto tile = [&](const double2 (&xc)[KS], double2 (&xn)[KS]) {
      // (loop control in plain ints: uniform bools that live across blocks came back as v_cndmask / v_readfirstlane pairs)
      // (and every condition is recomputed from them where it is used: carried from one block to the next it takes a trip through
      // a vector register)
      // wrap = 1 when this is its group's last tile (c + 1 == ncs), by arithmetic the optimiser cannot see through: as a compare
      // (+ select) it widens the flag through a vector register (v_cndmask, v_readfirstlane, v_cmp_ne) in every tile
      int wrap, keep;
      asm("s_lshr_b32 %0, %1, 31" : "=s"(wrap) : "s"(ncs - 2 - c));
      asm("s_add_i32 %0, %1, -1" : "=s"(keep) : "s"(wrap) : "scc"); // all ones unless the group ends
      const int cnext = (c + 1) & keep, gnext = g + wrap;
      if (wrap != 0 && left > 1) { // uniform.  Volatile so that it STAYS a branch: as a select the conversion runs in every tile and the
                          // tile right behind a switch waits for the tile load that was just issued
        if constexpr (KS & 1) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(qc) : "v"(qnd));
        else asm volatile("v_mov_b32 %0, %1" : "=v"(qc) : "v"(qni));
      }
      if constexpr (!(RSMP_EXP_SKIP & 16)) fill(xn, qc, left > 1 ? cnext : c); // after the last tile: a harmless re-read
      double accA = 0.0, accB = 0.0;
#pragma unroll
      for (int s = 0; s < (RSMP_EXP_HALFMFMA ? 3 : KS); ++s) { // RSMP_EXP_HALFMFMA: timing experiment only (WRONG results)
        if constexpr ((RSMP_EXP_SKIP & 32) != 0) { // no MFMAs: one add per operand keeps the loads alive
          accA += cc[s] + ((RSMP_EXP_SKIP & 16) ? x0[s].x : xc[s].x);
          accB += (RSMP_EXP_SKIP & 16) ? x0[s].y : xc[s].y;
        } else {
          accA = __builtin_amdgcn_mfma_f64_4x4x4f64(cc[s], (RSMP_EXP_SKIP & 16) ? x0[s].x : xc[s].x, accA, 0, 0, 0);
          accB = __builtin_amdgcn_mfma_f64_4x4x4f64(cc[s], (RSMP_EXP_SKIP & 16) ? x0[s].y : xc[s].y, accB, 0, 0, 0);
        }
      }
#if RSMP_EXP_HALFMFMA
#pragma unroll
      for (int s = 3; s < KS; ++s) { accA += xc[s].x * 1e-30; accB += cc[s] * 1e-30 + xc[s].y * 1e-30; } // keep the loads alive
#endif
      const int ib = lane_ib + 16 * g + c * pl4;
      if constexpr (OGEN) {
        if ((unsigned)ib < (unsigned)cnt && 16 * g + rloc < pl) {
          const long long A = oabs0 + ib;
          float *const p = (A >= io.out_abs0 && A < io.out_end) ? oext + (A - io.out_abs0) * io.nch : oring + (A & io.out_ring_mask) * io.nch;
          if (io.out_unaligned) { // (uniform) a caller's buffer that is only 4-byte aligned
            p[0] = (float)accA;
            p[1] = (float)accB;
          } else
            *reinterpret_cast<float2 *>(p) = make_float2((float)accA, (float)accB);
        }
      } else if constexpr (OUT64) {
        // (the descriptor spans the ring, so the block's range test is explicit here; lanes that fail it get an offset
        // behind the ring and the hardware drops their stores)
        const bool ok = (unsigned)ib < (unsigned)cnt && 16 * g + rloc < pl;
        const unsigned off = ok ? ((o64 + (unsigned)ib) & m64) * 8u : 0xffffffffu;
        const rsmp_v2u da = {(unsigned)__double2loint(accA), (unsigned)__double2hiint(accA)};
        const rsmp_v2u db = {(unsigned)__double2loint(accB), (unsigned)__double2hiint(accB)};
        __builtin_amdgcn_raw_buffer_store_b64(da, orsrcA, (int)off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(db, orsrcB, (int)off, 0, 0);
      } else
#if RSMP_BUFSTORE
      {
        // one unconditional buffer store per tile: outputs in front of the block (ib < 0 wraps to a huge offset) and behind
        // the round's last one fail the descriptor's range check and are dropped by the hardware; only the residue test of
        // the last (partial) group needs a select
        unsigned off = (unsigned)(lane_off0 + (16 * g + c * pl4) * frame_bytes);
        if (g == ngrp - 1 && (pl & 15)) { // uniform, and only chains whose polyL is not a multiple of 16 ever take it
          off = lane_dead_last ? 0xffffffffu : off;
          asm volatile("" : "+v"(off)); // (keeps it a branch: as selects it is two more vector instructions in every tile)
        }
        const rsmp_v2u d = {__float_as_uint((float)accA), __float_as_uint((float)accB)};
        if constexpr ((RSMP_EXP_SKIP & 8) != 0) { // no stores: only a result nobody produces would be written
          if (accA == 1.2345e300) __builtin_amdgcn_raw_buffer_store_b64(d, orsrc, (int)off, 0, 0);
        } else
        __builtin_amdgcn_raw_buffer_store_b64(d, orsrc, (int)off, 0, 0);
      }
#else
      if (ib >= 0 && ib < cnt && 16 * g + rloc < pl)
        *reinterpret_cast<float2 *>(obytes + (unsigned)(ib * frame_bytes)) = make_float2((float)accA, (float)accB);
#endif
      if (wrap != 0 && left > 1) { // uniform: the group switches
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("v_mov_b64 %0, %1" : "=v"(cc[s]) : "v"(cn[s]));
        __builtin_amdgcn_sched_barrier(0);
        // The copy is spelled out and fenced so that `cn`'s registers are dead before the next tile load is issued and the load
        // can land in them directly.  (As plain assignments the copy materialises at the END of the block, behind the load:
        // the compiler then lands the tile in scratch registers and moves it into `cn` behind an s_waitcnt vmcnt(0) right
        // after issuing it -- a full L2 round trip, and a drain of the output stores, at every group switch.)  The load is
        // unconditional for the same reason (clamped group index: past the wave's last group a harmless re-read).
        load_tile(min(gnext + 1, ngrp - 1), cn, qnd, qni);
      }
      g = gnext;
      c = cnext;
      --left;
    };
    while (true) {
      tile(x0, x1);
      if (left == 0) break;
      tile(x1, x0);
      if (left == 0) break;
    }
  };

  const bool run = fb.cnt > 0 && !(RSMP_EXP_SKIP & 1);
  if (RSMP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
  if (RSMP_PRIO == 2) __builtin_amdgcn_s_setprio(3);
  // round A: periods whose windows end inside the samples written above
  if constexpr (SPLIT && TWO) {
    if (run) poly_round(0, fb.KA, smp, -kPad, min(V, kSplitRaEnd) + kPad - 4 * KS);
    __syncthreads();
    { // second image: samples [kSplitRbStart, V) from the slots kept in registers, element 0 = sample kSplitRbStart
      double2 *l2 = reinterpret_cast<double2 *>(lds);
#pragma unroll
      for (int s = kSplitSB0; s < 16; ++s) {
        const int n = 2 * (tid + s * T);
        if (n < V) {
          l2[n - kSplitRbStart] = make_double2(z0[s].x, z0[s].y);
          l2[n + 1 - kSplitRbStart] = make_double2(v[s].x, v[s].y);
        }
      }
      if (tid < kPad && V > kSplitRbStart) l2[V - kSplitRbStart + tid] = make_double2(0.0, 0.0);
    }
    __syncthreads();
    if (tail_in_b) store_tail(reinterpret_cast<const double2 *>(lds), kSplitRbStart);
    if (run && fb.KA < fb.K) poly_round(fb.KA, fb.K, reinterpret_cast<const double2 *>(lds) - kSplitRbStart, kSplitRbStart, V + kPad - 4 * KS);
    return;
  } else if constexpr (SPLIT) {
    if (run) poly_round(0, fb.K, smp, -kPad, V + kPad - 4 * KS); // one round over the whole image
    return;
  }
  if (run) poly_round(0, fb.KA, smp, -kPad, min(V, kSA * T) + kPad - 4 * KS);
  RSMP_STAMP(6)
  __syncthreads();
  // round B: the rest of the block's samples replace the image, element 0 = sample kSB0*T
  {
    double2 *l2 = reinterpret_cast<double2 *>(lds);
#pragma unroll
    for (int s = kSB0; s < 16; ++s) {
      const int n = tid + s * T;
      if (n < V && !(RSMP_EXP_SKIP & 64)) l2[n - kSB0 * T] = make_double2(v[s].x, v[s].y);
    }
    if (tid < kPad && V > kSB0 * T) l2[V - kSB0 * T + tid] = make_double2(0.0, 0.0);
  }
  __syncthreads();
  if (tail_in_b) store_tail(reinterpret_cast<const double2 *>(lds), kSB0 * T);
  if (run && fb.KA < fb.K) poly_round(fb.KA, fb.K, reinterpret_cast<const double2 *>(lds) - kSB0 * T, kSB0 * T, V + kPad - 4 * KS);
  RSMP_STAMP(5)
#ifdef RSMP_STAMPS_BUILD
  if (stamping && tid == 0) atomicAdd(a.stamps + 7, 1ull);
#endif
#undef RSMP_STAMP
}

// (thin kernels around one body, so that the lean kernel keeps the name its profiles are filed under)
template <int LOG2P, int KS, bool OUT64>
__global__ __launch_bounds__(256, kFusedWaves) void fused_fast_kernel(FusedArgs a, FastIo io)
{
  fused_fast_body<LOG2P, KS, OUT64, false, false>(a, io);
}
// OMODE: 0 = float frames straight into the caller's buffer, 1 = the next fifo's fp64 ring, 2 = float frames via the fifo (OGEN)
template <int KS, int OMODE>
__global__ __launch_bounds__(256, 2) void fused_split_kernel(FusedArgs a, FastIo io)
{
  fused_fast_body<12, KS, OMODE == 1, true, OMODE == 2>(a, io);
}
// whole 8192-point blocks, polyphase stage in two rounds from registers (TWO)
template <int KS, int OMODE>
__global__ __launch_bounds__(256, 2) void fused_split2_kernel(FusedArgs a, FastIo io)
{
  fused_fast_body<12, KS, OMODE == 1, true, OMODE == 2, true>(a, io);
}

template <int LOG2P, int KS, bool OUT64> static hipError_t launch_fast_t(const FusedArgs &a, const FastIo &io, hipStream_t st)
{
  constexpr int N = 4096;
  size_t lds_bytes = 8 * size_t(fft_lds_doubles_halves(12));
  if (LOG2P < 12) lds_bytes = std::max(lds_bytes, 8 * size_t(std::max(fft_lds_doubles(LOG2P), fft8_lds_doubles(LOG2P))));
  lds_bytes = std::max(lds_bytes, size_t(kPad + kSA * (N / 16) + kPad) * 16);
  static DynLdsOnce attr;
  if (hipError_t e = attr.set(reinterpret_cast<const void *>(&fused_fast_kernel<LOG2P, KS, OUT64>), int(lds_bytes)); e != hipSuccess) return e;
  FusedArgs b = a;
  b.d.hp = io.nch >= 4 ? io.nch / 2 : 0;
  dim3 grid(item_grid(a.d.nblocks, a.d.C / 2, b.d.hp)), block(N / 16);
  hipLaunchKernelGGL((fused_fast_kernel<LOG2P, KS, OUT64>), grid, block, lds_bytes, st, b, io);
  return hipGetLastError();
}

template <int KS, int OMODE, bool TWO> static hipError_t launch_split_t(const FusedArgs &a, const FastIo &io, hipStream_t st)
{
  // the image of the longest sub-block (TWO: the first image, the longer of the two), or the exchange area of the
  // transforms, whichever is larger: <= 80 KB, two per CU
  const size_t lds_max = std::max(size_t(8) * fft_lds_doubles_halves(12), size_t(kPad + (TWO ? kSplitRaEnd : kSplitVsMax) + kPad) * 16);
  const size_t lds_bytes = TWO ? lds_max : std::max(size_t(8) * fft_lds_doubles_halves(12), size_t(kPad + a.d.Vs + kPad) * 16);
  static DynLdsOnce attr;
  const void *fn = TWO ? reinterpret_cast<const void *>(&fused_split2_kernel<KS, OMODE>) : reinterpret_cast<const void *>(&fused_split_kernel<KS, OMODE>);
  if (hipError_t e = attr.set(fn, int(lds_max)); e != hipSuccess) return e;
  FusedArgs b = a;
  b.d.hp = io.nch >= 4 ? io.nch / 2 : 0;
  dim3 grid(item_grid(a.d.nblocks, a.d.C / 2, b.d.hp)), block(256);
  if (TWO) hipLaunchKernelGGL((fused_split2_kernel<KS, OMODE>), grid, block, lds_bytes, st, b, io);
  else hipLaunchKernelGGL((fused_split_kernel<KS, OMODE>), grid, block, lds_bytes, st, b, io);
  return hipGetLastError();
}

// V samples of an 8192-point block in two LDS images: the second must reach the block's end, and every 4-residue block's
// windows (spread qb_spread samples) must fit the overlap of the two
bool fused_split_two_supported(int V, int taps, int ksteps, int qb_spread)
{
  return !knobs().no_split2 && !(V & 1) && V == 8192 - (taps - 1) && V > kSplitVsMax && V - kSplitRbStart + kPad <= kSplitRaEnd + kPad &&
         qb_spread + 4 * ksteps + 4 <= kSplitRaEnd - kSplitRbStart + kPad && ksteps >= 7 && ksteps <= 9;
}

bool fused_split_supported(int log2n, int L, int ksteps)
{
  return !knobs().no_fast && !knobs().no_split && L == 2 && log2n >= 13 && log2n <= 15 && ksteps >= 7 && ksteps <= 9;
}

hipError_t launch_fused_split(int omode, const FusedArgs &a, const FastIo &io, hipStream_t st, const char **kname)
{
  // what the kernel's indexing assumes, checked where the launch is made
  if (a.d.two) { // whole 8192-point blocks: one "sub-block" per block, two rounds
    if (a.d.nsub != 1 || a.d.Vs != a.d.V || a.d.Pref != 4096 || a.d.V <= kSplitVsMax || a.d.V > 8192 || (a.d.V & 1)) return hipErrorInvalidValue;
  } else if (a.d.Vs > kSplitVsMax) return hipErrorInvalidValue;
  if (a.d.nsub < 1 || a.d.Vs < 64 || (a.d.Vs & 1) || (a.d.V & 1) || a.d.nblocks % a.d.nsub || omode < 0 || omode > 2 ||
      (omode == 1 ? !io.out64 : omode == 2 ? !io.out_ring : !io.out) ||
      (a.d.Pref != 4096 && a.d.Pref != 8192 && a.d.Pref != 16384) || a.d.nsub * a.d.Vs < a.d.V || (a.d.nsub - 1) * a.d.Vs >= a.d.V || (io.nch & 1))
    return hipErrorInvalidValue;
  for (int i = 0; i < a.d.nsub; ++i) { // every sub-block's samples must be free of the component transforms' wrap-around
    const SubBlock sb = sub_block(i, a.d.V, a.d.Vs, a.d.Pref);
    const int ov = 2 * a.d.Pref - a.d.V; // taps - 1
    if (sb.win < 0 || sb.shift < 0 || sb.shift + sb.len / 2 + (ov + 1) / 2 > 4096 || (sb.len & 1) || sb.win + 4096 > a.d.Pref) return hipErrorInvalidValue;
  }
#define RSMP_SPLIT_CASE(ks, om)                                                        \
  if (a.KS == ks && omode == om && !a.d.two) {                                         \
    if (kname) *kname = "rsmp::fused_split_kernel<" #ks ", " #om ">";                  \
    return launch_split_t<ks, om, false>(a, io, st);                                   \
  }                                                                                    \
  if (a.KS == ks && omode == om && a.d.two) {                                          \
    if (kname) *kname = "rsmp::fused_split2_kernel<" #ks ", " #om ">";                 \
    return launch_split_t<ks, om, true>(a, io, st);                                    \
  }
  // (9 k-steps: 80 phases at step 147, the windows of a 4-residue block spread over 34 samples)
  RSMP_SPLIT_CASE(7, 0) RSMP_SPLIT_CASE(7, 1) RSMP_SPLIT_CASE(7, 2) RSMP_SPLIT_CASE(8, 0) RSMP_SPLIT_CASE(8, 1) RSMP_SPLIT_CASE(8, 2)
  RSMP_SPLIT_CASE(9, 0) RSMP_SPLIT_CASE(9, 1) RSMP_SPLIT_CASE(9, 2)
#undef RSMP_SPLIT_CASE
  return hipErrorInvalidValue;
}

bool fused_fast_supported(int log2n, int log2p, int ksteps)
{
  return !knobs().no_fast && log2n == 12 && (log2p == 11 || log2p == 12) && (ksteps == 7 || ksteps == 8);
}

#define RSMP_FAST_CASE(p, ks)                                                                        \
  if (log2p == p && a.KS == ks) {                                                                    \
    if (kname) *kname = io.out64 ? "rsmp::fused_fast_kernel<" #p ", " #ks ", true>" : "rsmp::fused_fast_kernel<" #p ", " #ks ", false>"; \
    return io.out64 ? launch_fast_t<p, ks, true>(a, io, st) : launch_fast_t<p, ks, false>(a, io, st); \
  }

hipError_t launch_fused_fast(int log2p, const FusedArgs &a, const FastIo &io, hipStream_t st, const char **kname)
{
  RSMP_FAST_CASE(11, 7) RSMP_FAST_CASE(12, 7) RSMP_FAST_CASE(11, 8) RSMP_FAST_CASE(12, 8)
  return hipErrorInvalidValue;
}

} // namespace rsmp
