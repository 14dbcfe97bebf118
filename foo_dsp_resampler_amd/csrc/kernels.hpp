// Kernel argument blocks and launch entry points (implemented in kernels.hip).
//
// Coordinates: every fifo between stages is addressed by ABSOLUTE sample index since the stream was
// opened (the preload zeros of rate_base.h:417-422 occupy indices [0, preload)).  Rings are powers of
// two, so index -> address is `idx & mask`.  All channels of a handle advance in lock step, which is
// why one set of indices serves every channel.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>

#include "knobs.hpp"

namespace rsmp {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: one process may hold handles on several
// GPUs (RRX_open_batch_on), so "already set" is remembered per (kernel instance, device).  Setting it twice is harmless,
// so a race between two handles' threads needs no lock.
struct DynLdsOnce {
  std::atomic<unsigned long long> mask{0};
  hipError_t set(const void *fn, int bytes)
  {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 64 && ((mask.load(std::memory_order_acquire) >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev < 64) mask.fetch_or(1ull << dev, std::memory_order_release);
    return e;
  }
};

// Interleaved float32 frames: the caller-facing end of the chain (stage-0 input or final output).
// A frame with absolute index a lives in the external buffer when ext != nullptr and
// ext_begin <= a < ext_end, else in the ring.  This is what lets a device-resident push be
// consumed in place and a device-resident pull be produced in place (no staging copy of bulk data).
struct F32View {
  float *ring;
  long long ring_mask;          // frames - 1
  long long ring_stream_stride; // floats between streams
  float *ext;
  long long ext_begin, ext_end; // absolute frame range held by ext
  long long ext_stream_stride;  // floats between streams
  int nch;                      // channels per stream
};

// Planar fp64 ring between two stages: [channel][cap].
struct F64View {
  double *ring;
  long long mask;        // items - 1
  long long chan_stride; // items between channels
};

struct DftArgs {
  const double2 *G;      // N entries: DFT_N(L * h_placed) / N, natural order, e^{-i} convention
  const double2 *Gr;     // dftx_kernel: [L][N/L] spectra of the filter's polyphase components, DFT_P(L * h_placed[L j + r]) / P
  const double2 *tw_fwd; // twiddle table for the forward size P
  const double2 *tw_inv; // twiddle table for the inverse size Nd
  const double2 *tw_fwd8; // forward size P, 8-points-per-thread plan (fft8_regs)
  const double2 *tw_inv8; // inverse size Nd, 8-points-per-thread plan (frequency-domain decimation by 2)
  long long B0;          // absolute index of the first block of this launch
  long long out_offset;  // preload of the destination fifo (absolute index of stage output 0)
  int nblocks;
  int C;                 // total channels (streams * nch)
  int L;                 // zero-stuffing factor
  int c0;                // initial remL (time-domain stuffing phase)
  int V;                 // N - (taps-1): valid filtered samples per block before decimation
  int Vout;              // outputs kept per block when M == 1 (V, or the frequency-domain decimated count)
  int q;                 // inputs consumed per block (frequency-domain paths)
  int M;                 // time-domain decimation step (1 = none)
  int nchs;              // > 0: channels per stream of a batch handle whose pairs must not straddle streams (pair_channels); else 0
  int npairs;            // pair_count(C, nchs), filled in by the launchers
  unsigned pps_magic;    // ceil(2^32 / pairs per stream) when nchs > 0 (pair_magic), filled in by the launchers
  int hp;                // channel pairs per interleaved float frame whose workgroups are co-located (item_map); 0/1 = none
  long long in_limit;    // input items at absolute index >= in_limit read as zero (unused by the engine: always +inf)
  long long clip_lo, clip_hi; // only stage outputs with absolute index in [clip_lo, clip_hi) are stored (always everything)
  // Sub-blocked fused launch (fused_fast_kernel<.., SPLIT>, fused_fast.hip): every block of the reference is computed as
  // `nsub` sub-blocks of `Vs` valid samples (the last one shorter) on 4096-point component transforms; B0 / nblocks then
  // count SUB-blocks (B0 = Bref0 * nsub) and G holds the two component spectra.  nsub = 0: not sub-blocked.
  int nsub, Vs;
  int two;               // 1: whole 8192-point blocks (nsub = 1, Vs = V), polyphase stage in two rounds (kSplitRaEnd / kSplitRbStart)
  int Pref;              // inputs a reference block spans (N / L)
  long long Bref0;       // first reference block of the launch
};

// Geometry of sub-block `i` of a reference block with V valid samples out of N/L = `Pref` inputs (x2 chains): its `len` valid
// samples start `off` samples into the block's valid range; its 4096-point input window starts `win` inputs into the block's
// input span -- as late as the samples allow, but never so late that it would reach past the block's own inputs -- and its
// first valid sample is element `shift` of each component transform's output.
struct SubBlock { int off, len, win, shift; };
__host__ __device__ inline SubBlock sub_block(int i, int V, int Vs, int Pref)
{
  SubBlock s;
  s.off = i * Vs;
  s.len = V - s.off < Vs ? V - s.off : Vs;
  s.win = (s.off >> 1) < Pref - 4096 ? (s.off >> 1) : Pref - 4096;
  s.shift = (s.off >> 1) - s.win;
  return s;
}

// number of channel pairs (= workgroups per block) of a launch, see pair_channels (fifo_device.hpp)
__host__ __device__ inline int pair_count(int C, int nchs) { return nchs > 0 ? (C / nchs) * ((nchs + 1) >> 1) : (C + 1) >> 1; }

__host__ __device__ inline unsigned pair_magic(int C, int nchs)
{
  const unsigned long long pps = (unsigned long long)(((nchs > 0 ? nchs : C) + 1) >> 1);
  return pps <= 1 ? 0u : (unsigned)(((1ull << 32) + pps - 1) / pps); // 0: one pair per stream, no division
}

// Long blocks (N = 32768 ... 131072): N = 16 x M four-step transform in three launches through a workspace (dftbig.hip)
struct BigDftArgs {
  DftArgs d;             // tw_fwd / tw_inv: fft_regs tables of the forward / inverse ROW lengths
  const double2 *twN;    // exp(+2 pi i j / N), j < N
  double2 *w1, *w2;      // workspaces [item][16][Mp] and [item][16][Md]
  int log2n, log2mp, log2md; // block length; forward / inverse row lengths (log2 of P/16 and Nd/16)
  int fdomain_in;        // 1: the block is 16*Mp consecutive stage inputs (L = 1, or xL in the frequency domain); 0: zero stuffing
  int item0;             // first (block, pair) item of this launch (filled in by launch_dft_big)
};
bool big_dft_supported(int log2n, int log2p, int log2nd);
hipError_t launch_dft_big(bool src_f32, bool dst_f32, const F32View &sf, const F64View &sd, const F32View &df, const F64View &dd,
                          BigDftArgs a, int ws_items, hipStream_t st);

// per-block output bookkeeping of the fused launch, computed on the host (64-bit divisions stay there)
struct FusedBlock {
  long long i_lo; // first output whose window starts inside the block (absolute output index)
  int cnt;        // number of such outputs whose window also ends inside the block
  int irel_lo;    // i_lo - kk_lo * polyL, kk_lo = i_lo / polyL (first period touched)
  int base_li;    // kk_lo * step - b0: window start of (period kk_lo, residue r) is qr(r) + base_li
  int K;          // periods touched by [i_lo, i_lo + cnt)
  int KA;         // matrix-pipe variant: periods [0, KA) are computed from the first LDS image, the rest from the second
  // seam outputs in front of this block (window straddles blocks B-1 | B): indices [seam_i0, i_lo); output seam_i0 has
  // phase seam_ph0 and its window starts seam_q0 samples into the [tail of B-1 | head of B] image (seam_kernel)
  long long seam_i0;
  int seam_q0, seam_ph0;
};
// Matrix-pipe variant of the fused kernel (N = 4096, 256 threads): the block's samples sit in LDS in two rounds,
// round A = register slots [0, kFusedSA) of the inverse FFT (samples [0, 256*kFusedSA) plus a 32-sample margin),
// round B = slots [kFusedSB0, 16); the two overlap by (kFusedSA - kFusedSB0) * 256 + 32 samples, which every period's
// windows (all residues) must fit into one way or the other.  RSMP_WG4 sizes them for four workgroups per CU instead of three.
#ifndef RSMP_WG4
#define RSMP_WG4 0
#endif
// RSMP_WG2 (experiment): two workgroups per CU with 256 VGPRs each, the whole block (V <= 3584) in the first image
#ifndef RSMP_WG2
#define RSMP_WG2 0
#endif
constexpr int kFusedSA = RSMP_WG2 ? 14 : RSMP_WG4 ? 9 : 12, kFusedSB0 = kFusedSA - 2, kFusedWaves = RSMP_WG2 ? 2 : RSMP_WG4 ? 4 : 3;

// Closed forms of a block's bookkeeping; evaluated by fused_prep_kernel on the device (one thread per block of the
// launch) and by the engine for its consistency checks.
struct FusedPrepArgs {
  long long b_offset, B0, at0; // as in FusedArgs / DftArgs
  int V, polyL, step, n, nblocks;
  int two_round, KS, qb_max;   // matrix-pipe variant: split of the periods over its two LDS images
  int ra_end, rb_start;        // ... the first image holds samples [0, ra_end) (+ 32), the second starts at rb_start; 0 = the lean
                               // kernel's kFusedSA * 256 / kFusedSB0 * 256
  int qb_min;                  // (smallest / largest window start of a 4-residue block, relative to its period)
  long long clip_lo, clip_hi;  // only outputs with index in [clip_lo, clip_hi) belong to this launch (standalone stage)
  int nsub, Vs;                // sub-blocked launch (DftArgs::nsub): entry k is sub-block k % nsub of reference block B0 + k / nsub
};
__host__ __device__ inline FusedBlock fused_block_info(const FusedPrepArgs &p, int k)
{
  long long b0 = p.b_offset + (p.B0 + k) * (long long)p.V;
  int V = p.V;
  if (p.nsub > 0) {
    const int kr = k / p.nsub, i = k - kr * p.nsub;
    b0 = p.b_offset + (p.B0 + kr) * (long long)p.V + (long long)i * p.Vs;
    V = p.V - i * p.Vs < p.Vs ? p.V - i * p.Vs : p.Vs;
  }
  const long long nlo = b0 * p.polyL - p.at0, nhi = (b0 + V - p.n + 1) * p.polyL - p.at0;
  long long ilo = nlo <= 0 ? 0 : (nlo + p.step - 1) / p.step, ihi = nhi <= 0 ? 0 : (nhi + p.step - 1) / p.step;
  if (ilo < p.clip_lo) ilo = p.clip_lo;
  if (ihi > p.clip_hi) ihi = p.clip_hi;
  FusedBlock fb;
  fb.i_lo = ilo;
  fb.cnt = ihi > ilo ? int(ihi - ilo) : 0;
  const long long kk_lo = ilo / p.polyL;
  fb.irel_lo = int(ilo - kk_lo * p.polyL);
  fb.base_li = int(kk_lo * p.step - b0);
  fb.K = fb.cnt > 0 ? int((ihi - 1) / p.polyL - kk_lo) + 1 : 0;
  fb.KA = fb.K;
  { // first output whose window starts at or behind the previous block's tail: (b0 - (n - 1)) * L - at0 over step, rounded up
    const long long ns = (b0 - (p.n - 1)) * p.polyL - p.at0;
    const long long s0 = ns <= 0 ? 0 : (ns + p.step - 1) / p.step;
    const long long a0 = p.at0 + s0 * p.step, q0 = a0 / p.polyL;
    fb.seam_i0 = s0;
    fb.seam_q0 = int(q0 - (b0 - (p.n - 1)));
    fb.seam_ph0 = int(a0 - q0 * p.polyL);
  }
  const int ra_end = p.ra_end > 0 ? p.ra_end : kFusedSA * 256, rb_start = p.ra_end > 0 ? p.rb_start : kFusedSB0 * 256;
  if (p.two_round && V > ra_end) { // periods whose (padded) windows end inside the first LDS image
    const int a_hi = ra_end + 32 - 4 * p.KS - 3, num = a_hi - fb.base_li - p.qb_max;
    const int ka = num < 0 ? 0 : num / p.step + 1;
    fb.KA = ka < fb.K ? ka : fb.K;
    // a multiple of 4 periods in the first image when the second one can take the rest: the two rounds then need
    // ceil(K / 4) column steps of 4 periods together instead of one more
    const int k4 = fb.KA & ~3;
    if (fb.KA < fb.K && k4 > 0 && fb.base_li + p.qb_min + k4 * p.step >= rb_start) fb.KA = k4;
  }
  return fb;
}
constexpr int kFusedMaxBlocks = 1024; // capacity of the per-stage block table in HBM

// dft -> vpoly0 fused launch (fused.hip)
struct FusedArgs {
  DftArgs d;             // the FFT-FIR part (out_offset unused)
  const double *tab;     // polyphase table [phase][tap]
  const double *cft;     // per-thread coefficient tiles [tap < 32][g < 2][thread], shifted + zero padded
  double *seam;          // [slot][channel (C + 1 of them)][head|tail][32] stage-1 samples at block edges
  long long at0;         // absolute initial clock of the poly stage, units 1/polyL
  long long b_offset;    // preload of the stage-1 fifo (absolute index of the first FFT output)
  long long out_offset2; // preload of the fifo after the poly stage
  int seam_mask;         // slots - 1
  int n, polyL, step;    // taps per phase, phases, clock step
  int span;              // n + largest window offset inside a G-tile
  int NG, KC;            // residue groups, period chunks (NG * KC <= threads)
  int kper;              // periods per chunk
  const double *cfm;     // matrix-pipe variant: A operands [16-residue group][k-step][lane]; null = vector variant
  int NGRP, KS;          // 16-residue groups, k-steps (4 taps each) of a 4-residue block's common window
  int dbg;               // profiling ablations (RSMP_DBG env); 0 in production
  unsigned long long *stamps; // RSMP_STAMPS: per-phase cycle sums [8] (null in production)
  const FusedBlock *blk; // [nblocks] in HBM, written by fused_prep_kernel ahead of the launch
  const int *qtab;       // matrix-pipe variant: window start (at0 + rb*step)/polyL of every 4-residue block rb = 4*i, [NGRP*4]
  const double2 *cfm2;   // the same A operands two k-steps per 16-byte element: [group][(KS + 1) / 2][lane] (lean kernel)
};

// Lean fast path of the matrix-pipe variant (fused_fast.hip): both ends are plain interleaved float frames in one buffer each
struct FastIo {
  const float *in;            // frame `in_abs0` (absolute input index) of stream 0, channel 0
  const float *in_ring;       // frames below in_abs0 (the tail of the previous push): fifo 0's ring, frame (index & in_ring_mask)
  long long in_ring_mask, in_ring_stream_stride;
  float *out;                 // frame `out_abs0` (absolute index in the output fifo) of stream 0, channel 0
  long long in_abs0, out_abs0;
  long long in_stream_stride, out_stream_stride; // floats between streams
  int nch;                    // channels per stream (even)
  int in_unaligned;           // sub-blocked form only: `in` is not 8-byte aligned (channels read one float at a time)
  // sub-blocked form, omode 2: frame with absolute index A is in `out` when out_abs0 <= A < out_end, else in the fifo's ring
  float *out_ring;
  long long out_ring_mask, out_ring_stream_stride, out_end;
  int out_unaligned;          // `out` is not 8-byte aligned
  // OUT64 instances (the polyphase stage feeds another stage): planar fp64 ring of the destination fifo instead of `out`
  double *out64;              // ring of channel 0
  long long out64_mask, out64_chan_stride; // items - 1, items between channels
};
bool fused_fast_supported(int log2n, int log2p, int ksteps);
hipError_t launch_fused_fast(int log2p, const FusedArgs &a, const FastIo &io, hipStream_t st, const char **kname = nullptr);
// the sub-blocked form (fused_split_kernel): x2 stages with 8192- or 16384-point blocks -> vpoly0
constexpr int kSplitVsMax = 5056; // valid samples of a sub-block: (32 + Vs + 32) 16-byte LDS elements, two workgroups per CU
// 8192-point blocks fit ONE pair of component transforms whole (V = 8192 - (taps - 1) samples): one workgroup, the polyphase
// stage in two rounds -- samples [0, kSplitRaEnd) (register slots 0 .. 8 of both components) first, [kSplitRbStart, V) from
// the slots kept in registers (8 .. 15) afterwards
constexpr int kSplitSA = 9, kSplitSB0 = 8, kSplitRaEnd = 2 * kSplitSA * 256, kSplitRbStart = 2 * kSplitSB0 * 256;
bool fused_split_supported(int log2n, int L, int ksteps);
// omode: 0 = float frames straight into FastIo::out (every output of the launch lies inside it, 8-byte aligned), 1 = the next
// fifo's fp64 ring (out64), 2 = float frames wherever the output fifo has them (out / out_ring)
hipError_t launch_fused_split(int omode, const FusedArgs &a, const FastIo &io, hipStream_t st, const char **kname = nullptr);
bool fused_split_two_supported(int V, int taps, int ksteps, int qb_spread); // the whole-block two-round form (8192-point blocks)

struct PolyArgs {
  const double *tab;     // [phase][tap][order+1]
  long long rd;          // absolute index of the stage's read pointer
  long long at;          // clock relative to rd: integer (order 0, units 1/L) or 32.32 fixed point
  long long step;        // same units as `at`
  long long out_abs;     // absolute index (in the destination fifo) of output 0 of this launch
  long long count;       // outputs to produce
  int C, n, L, phase_bits, tile, win;
  int tab_lds;           // order 0: copy the [L][n] table into LDS behind the window (it fits)
  int coop;              // orders 1-3: 8 lanes per output (poly_coop_kernel); needs n % 8 == 0
  int shared_rows;       // orders 1-3: interpolated rows computed once per tile and shared by 16 channels (polyi_kernel);
                         // `win` is then the per-channel window stride in doubles (odd)
};

struct HalfArgs {
  long long rd, out_abs, count;
  int C, ncoef, pre;
  double coef[13];
};

// All launchers return hipSuccess or the launch error; `kname` (optional) receives the name of the kernel instance
// that was picked, as rocprofv3 prints it (static string).
hipError_t launch_dft(int log2n, int log2p, int log2nd, bool src_f32, bool dst_f32, const F32View &sf, const F64View &sd,
                      const F32View &df, const F64View &dd, const DftArgs &a, hipStream_t st, const char **kname = nullptr);
hipError_t launch_poly(int order, bool src_f32, bool dst_f32, const F32View &sf, const F64View &sd, const F32View &df,
                       const F64View &dd, const PolyArgs &a, hipStream_t st, const char **kname = nullptr);
hipError_t launch_half(bool src_f32, bool dst_f32, const F32View &sf, const F64View &sd, const F32View &df,
                       const F64View &dd, const HalfArgs &a, hipStream_t st, const char **kname = nullptr);
bool dft_shape_supported(int log2n, int log2p, int log2nd);
// x4 upsampling on 8192-point blocks as four 2048-point component transforms (dftx.hip); needs DftArgs::Gr
bool dftx_supported(int log2n, int log2p, int log2nd);
hipError_t launch_dftx(int log2n, bool src_f32, bool dst_f32, const F32View &sf, const F64View &sd, const F32View &df,
                       const F64View &dd, const DftArgs &a, hipStream_t st, const char **kname = nullptr);
hipError_t launch_fused(int log2n, int log2p, bool src_f32, bool dst_f32, const F32View &sf, const F64View &sd,
                        const F32View &df, const F64View &dd, const FusedArgs &a, hipStream_t st, const char **kname = nullptr);
// seam_kernel: the outputs whose window straddles two blocks; launch after launch_fused on the same stream
hipError_t launch_seam(bool dst_f32, const F32View &df, const F64View &dd, const FusedArgs &a, hipStream_t st);
bool fused_shape_supported(int log2n, int log2p, int n, int span, int max_seam_outputs);
bool fused_mfma_supported(int log2n, int log2p, int ksteps);
hipError_t launch_fused_prep(const FusedPrepArgs &p, FusedBlock *out, hipStream_t st);

// standalone rational polyphase stage on the matrix pipe (polymf.hip)
struct PolyMfArgs {
  const double *cfm;     // A operands [16-residue group][k-step][lane], as in FusedArgs
  const int *qtab;       // window start of every 4-residue block, as in FusedArgs
  const FusedBlock *blk; // per tile: fused_block_info with V = Vt, n = 1, b_offset = 0, clipped to the launch's outputs
  long long B0;          // first tile of the launch: tile B covers stage-input samples [B*Vt, (B+1)*Vt)
  long long at0;         // absolute initial clock of the stage, units 1/polyL
  long long out_offset;  // preload of the destination fifo
  long long in_limit;    // stage-input samples at absolute index >= in_limit are not written yet: read as zero
  int nblocks, C, Vt, n, polyL, step, NGRP;
  int nchs, npairs;      // as DftArgs::nchs / npairs
  unsigned pps_magic;    // as DftArgs::pps_magic
};
bool polymf_supported(int ksteps);
hipError_t launch_polymf(int ksteps, bool src_f32, bool dst_f32, const F32View &sf, const F64View &sd, const F32View &df,
                         const F64View &dd, const PolyMfArgs &a, hipStream_t st, const char **kname = nullptr);
// element-wise copy of absolute range [a0, a1) of every channel from one fifo view to another
// (ring regrow, carrying the unconsumed tail of an in-place push into the ring, device pulls)
hipError_t launch_copy(bool f32, const F32View &sf, const F64View &sd, const F32View &df, const F64View &dd, long long a0,
                       long long a1, int C, hipStream_t st);

} // namespace rsmp
