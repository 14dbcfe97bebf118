// Fused hot path for the chain  dft(xL, L in {1,2,4}) -> vpoly0 -> caller  (BASELINE configs 44.1k->48k,
// 44.1k->96k, 96k->44.1k; reference: rate/dft_filter.h:60-190 followed by
// rate/rate_filters_generic.h:272-305).
//
// One workgroup = one overlap-save block of one channel PAIR (the two channels are the real and
// imaginary part of one complex transform).  The block's stage-1 samples never leave the CU: after the
// inverse FFT they stay in LDS as (A,B) pairs (this LDS image replaces the fifo of rate/fifo.h between the
// two stages) and the polyphase FIR reads them from there, one aligned ds_read_b128 per tap.  Only 2*(n-1)
// samples per block and channel (the block's head and tail) go to a small HBM "seam" ring so that the
// outputs whose 24-tap window straddles two blocks can be produced by seam_kernel afterwards.
//
// Polyphase mapping: outputs i and i + k*L share a phase, hence their coefficients.  A thread owns G
// consecutive residues r = i mod L and a range of periods k; its G x span coefficients (rows shifted
// and zero-padded to a common window) live in registers for the whole block, so the inner loop is
// one LDS read per G (x 2 channels) fp64 FMAs, and no coefficient traffic at all.
#include "fft_device.hpp"
#include "fifo_device.hpp"
#include "kernels.hpp"

#include <algorithm>
#include <type_traits>
#include <atomic>
#include <cstdlib>
#include <cstdio>

// Profiling switches (knobs.hpp; read once per process, all zero / unset in production).  The RSMP_DBG ablation bits give
// WRONG results and only exist in -DRSMP_EXPERIMENTS builds (RSMP_DBGBITS is the constant 0 otherwise):
//   RSMP_DBG bit 0 (1): skip the polyphase stage       bit 1 (2): skip the inverse FFT      bit 2 (4): skip the forward FFT
//            bit 4 (16): compute the polyphase sums but do not store them
//            bit 5 (32) / 6 (64): matrix-pipe variant, skip round B / round A
//            bit 9 (512): reuse the first B operands of an item (no further LDS reads)
//            bit 10 (1024): do not reload A tiles      bit 8 (256): drain all counters at every RSMP_STAMPS stamp
//   RSMP_STAMPS=1   per-phase cycle sums of wave 0 (s_memtime), printed when the handle closes (intrusive: ~2x slower)
//   RSMP_LDS_PAD=n  add n bytes of LDS per workgroup (occupancy experiments), RSMP_OCC=1 prints the resulting blocks/CU
//   RSMP_NO_MFMA / RSMP_NO_FUSE / RSMP_NO_POLYMF / RSMP_NO_SIDE / RSMP_SLAB_MB: engine.cpp
//
// RSMP_FWD8=1: L = 2 forward transform on all four waves, 8 points per thread (fft8_regs), no replication exchange.
// Parity-green; the FFT part of the kernel gets 21 % faster (1.71 -> 1.36 ms with the polyphase stage skipped) but the
// whole kernel does not (2.67 vs 2.63 ms): the polyphase phase loses the FFT phases it used to overlap with.
#ifndef RSMP_FWD8
#define RSMP_FWD8 0
#endif
#ifdef RSMP_EXPERIMENTS
#define RSMP_DBGBITS (a.dbg)
#else
#define RSMP_DBGBITS 0
#endif
// twiddles per pass prefetched ahead of the preceding LDS exchange (forward / inverse transform of the MF variant)
// RSMP_FINE=1 (variant builds only): per-segment cycle sums of the polyphase item loop of wave 0 in the stamped workgroups,
// with every counter drained at each segment boundary (slots 8..15 of the stamp buffer).
#ifndef RSMP_FINE
#define RSMP_FINE 0
#endif
#ifndef RSMP_PFW
#define RSMP_PFW 15
#endif
#ifndef RSMP_PFI
#define RSMP_PFI 8
#endif

namespace rsmp {

constexpr int kPad = 32;      // LDS guard samples around each channel's block
constexpr int kSpanMax = 32;  // largest window length (taps + offset spread) the register tile supports
// matrix-pipe variant (N = 4096): LDS holds the block's samples in two rounds so that three workgroups fit a CU:
// round A = register slots [0, kSA) i.e. samples [0, 256*kSA) plus kPad more, round B = slots [kSB0, 16)
constexpr int kSA = kFusedSA, kSB0 = kFusedSB0;

// SPAN = window length of a G-tile (compile time, so the whole tap loop is straight-line code and the LDS
// reads are issued ahead of the FMAs); a.span <= SPAN, coefficients beyond a.span are zero.
//
// Grid: 1-D, one workgroup per work item w -> (pair = w % npairs, block = w / npairs).  Blocks are dealt
// round-robin over the 8 XCDs, so with npairs % 8 == 0 the consecutive blocks of one pair land on the same
// XCD and their 2*(taps-1)-sample input overlap is an L2 hit.  (A persistent grid with the coefficient tile
// kept in registers across items was measured 1.6x SLOWER: at the 256-VGPR cap the FFT passes lose their
// load/compute overlap.)
//
// MF variant (polyphase on the fp64 matrix pipe): see the "polyphase FIR on v_mfma_f64_4x4x4" section below;
// SPAN then counts k-steps (4 taps each) of a 4-residue block's common window.
template <int LOG2N, int LOG2P, int G, int SPAN, bool MF>
__global__ __launch_bounds__((1 << LOG2N) / 16, MF ? kFusedWaves : 2) void fused_kernel(AnyView in, AnyView out, FusedArgs a)
{
  constexpr int N = 1 << LOG2N, P = 1 << LOG2P;
  constexpr int T = N / 16, TF = P / 16;
  extern __shared__ __attribute__((aligned(16))) double lds[];

  const int tid = threadIdx.x;
  const int npairs = a.d.npairs;
  const int V = a.d.V;
  const bool fwd_active = tid < TF;
  const double2 *__restrict__ Gp = a.d.G;
  double2 *smp = reinterpret_cast<double2 *>(lds) + kPad; // smp[n] = (channel A, channel B) sample n of a block
  const int nm1 = a.n - 1;

  // ------------------------------------------------------------------ per-thread polyphase constants
  // thread -> (residue group m, period chunk kc); residues r0..r0+G-1 (mod polyL)
  const int pl = a.polyL, step = a.step, at0 = (int)a.at0;
  const int m = MF ? 0 : tid % a.NG, kc = MF ? 0 : tid / a.NG;
  const bool poly_thread = kc < a.KC;
  const int r0 = G * m;
  const int qr0 = (at0 + r0 * step) / pl;
  bool rv[G];
#pragma unroll
  for (int g = 0; g < G; ++g) rv[g] = r0 + g < pl;
  const double *__restrict__ cft = a.cft + tid;

  // RSMP_STAMPS instrumentation: wave 0's view of the phase boundaries (s_memtime, shader clock)
  // (one workgroup in 64 is stamped, so that the stamps do not change what they measure)
  const bool stamping = a.stamps && (blockIdx.x & 63) == 5;
  unsigned long long tstamp = stamping ? __builtin_readcyclecounter() : 0;
#define RSMP_STAMP(slot) \
  if (stamping) { \
    if (RSMP_DBGBITS & 256) __builtin_amdgcn_s_waitcnt(0); \
    const unsigned long long now = __builtin_readcyclecounter(); \
    if (tid == 0) atomicAdd(a.stamps + (slot), now - tstamp); \
    tstamp = now; \
  }

  { // one work item per workgroup (see the note above about a persistent loop)
    int bl, pair;
    if (!item_map(blockIdx.x, a.d.nblocks, npairs, a.d.hp, bl, pair)) return; // uniform
    const long long B = a.d.B0 + bl;
    const PairCh pc = pair_channels(pair, a.d.C, a.d.nchs, a.d.pps_magic);
    const int ca = pc.ca, cb = pc.cb;
    const bool hasb = pc.hasb;

    // ---------------------------------------------------------------- load the block (fp32 -> fp64)
    // L = 2 in the matrix-pipe variant: the forward transform has half the points of the inverse one and runs
    // 8 points per thread on all waves (fft8_regs); a thread then already holds the 8 distinct spectrum values
    // Zp[tid + (s & 7) * T] its 16 inverse-transform inputs need, so the replication exchange disappears
    constexpr bool FWD8 = MF && (LOG2N - LOG2P == 1) && RSMP_FWD8;
    c64 v[16];
    c64 u8[8];
    {
      const long long e0 = B * a.d.q;
      bool fast = false;
      const float2 *p2 = nullptr;
      int fstride = 1; // float2 elements between consecutive frames of this channel pair
      if (in.is_f32 && hasb && !(in.f.nch & 1)) { // channels 2p, 2p+1 sit side by side in every frame
        const int hp = in.f.nch >> 1, strm = pair / hp, pin = pair - strm * hp;
        fstride = hp;
        if (in.f.ext && e0 >= in.f.ext_begin && e0 + P <= in.f.ext_end) {
          const float *p = in.f.ext + strm * in.f.ext_stream_stride + (e0 - in.f.ext_begin) * in.f.nch + 2 * pin;
          fast = (reinterpret_cast<unsigned long long>(p) & 7) == 0;
          p2 = reinterpret_cast<const float2 *>(p);
        } else if ((!in.f.ext || e0 + P <= in.f.ext_begin) && (e0 & in.f.ring_mask) + P <= in.f.ring_mask + 1) {
          const float *p = in.f.ring + strm * in.f.ring_stream_stride + (e0 & in.f.ring_mask) * in.f.nch + 2 * pin;
          fast = (reinterpret_cast<unsigned long long>(p) & 7) == 0;
          p2 = reinterpret_cast<const float2 *>(p);
        }
      }
      if constexpr (FWD8) { // every thread takes 8 points of the P-point forward transform: x[tid + s*T], T = P/8
        if (fast) {
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            const float2 f = p2[(tid + s * T) * fstride];
            u8[s] = {(double)f.x, (double)f.y};
          }
        } else {
          const ChanRef ia = chan_ref(in, ca), ib = chan_ref(in, hasb ? cb : ca);
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            const long long e = e0 + tid + s * T;
            u8[s].x = fifo_get(ia, e);
            u8[s].y = hasb ? fifo_get(ib, e) : 0.0;
          }
        }
      } else if (fwd_active) {
        if (fast) {
#pragma unroll
          for (int s = 0; s < 16; ++s) {
            const float2 f = p2[(tid + s * TF) * fstride];
            v[s] = {(double)f.x, (double)f.y};
          }
        } else {
          const PairSpan sp = in.is_f32 ? PairSpan{0, nullptr, 1, nullptr, nullptr, hasb} : pair_span(in, pair, hasb, e0, P, ca);
          if (sp.kind) { // planar fp64 rings (the producer is another stage), block contiguous in both
            span_load<16>(sp, tid, TF, v);
          } else {
            const ChanRef ia = chan_ref(in, ca), ib = chan_ref(in, hasb ? cb : ca);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
              const long long e = e0 + tid + s * TF;
              v[s].x = fifo_get(ia, e);
              v[s].y = hasb ? fifo_get(ib, e) : 0.0;
            }
          }
        }
      }
    }

    RSMP_STAMP(0)
    // ---------------------------------------------------------------- FFT-FIR (as dft_kernel)
    if constexpr (FWD8) {
      double2 g[16]; // in flight during the whole forward transform (32 + 64 registers live)
#pragma unroll
      for (int s = 0; s < 16; ++s) g[s] = Gp[tid + s * T];
      if (!(RSMP_DBGBITS & 4)) fft8_regs<LOG2P, -1>(u8, tid, a.d.tw_fwd8, lds);
      RSMP_STAMP(1)
#pragma unroll
      for (int s = 0; s < 16; ++s) v[s] = cmul(u8[s & 7], c64{g[s].x, g[s].y});
      __syncthreads(); // the inverse transform's exchange reuses the LDS the forward one just read
    } else {
    if (!(RSMP_DBGBITS & 4)) fft_regs<LOG2P, -1, (MF && LOG2P == LOG2N) ? 2 : 0, MF ? RSMP_PFW : 0>(v, tid, fwd_active, a.d.tw_fwd, lds);
    RSMP_STAMP(1)
    }
    if constexpr (FWD8) {
    } else if constexpr (LOG2P < LOG2N) {
      double2 g[16]; // issued before the exchange so the L2 latency overlaps it
#pragma unroll
      for (int s = 0; s < 16; ++s) g[s] = Gp[tid + s * T];
      double2 *l2 = reinterpret_cast<double2 *>(lds);
      if constexpr (LOG2N - LOG2P == 1) {
        // Z[tid + s*T] = Zp[(tid + s*T) mod P] = Zp[tid + (s & 7) * T]: a forward thread (tid < TF) already
        // holds those in its even slots; its odd slots are what thread tid + TF needs
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
      } else {
        if (fwd_active) {
#pragma unroll
          for (int s = 0; s < 16; ++s) l2[tid + s * TF] = make_double2(v[s].x, v[s].y);
        }
        __syncthreads();
        // Z[tid + s*T] = Zp[(tid + s*T) mod P]: only 16/L distinct entries per thread, each used L times
        constexpr int ND = 16 >> (LOG2N - LOG2P);
#pragma unroll
        for (int s = 0; s < ND; ++s) {
          const double2 z = l2[tid + s * T];
#pragma unroll
          for (int rep = 0; rep < 16 / ND; ++rep)
            v[s + rep * ND] = cmul(c64{z.x, z.y}, c64{g[s + rep * ND].x, g[s + rep * ND].y});
        }
        __syncthreads();
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const double2 g = Gp[tid + s * T];
        v[s] = cmul(v[s], c64{g.x, g.y});
      }
    }
    RSMP_STAMP(2)
    if (!(RSMP_DBGBITS & 2)) fft_regs<LOG2N, +1, MF ? 2 : 0, MF ? RSMP_PFI : 0>(v, tid, true, a.d.tw_inv, lds);
    RSMP_STAMP(3)

    // coefficient tile of this thread: rows of its G phases shifted to a common window start and zero
    // padded, pre-arranged by the host as [tap][g][thread] (coalesced, branch-free); issued here so the
    // L2 latency overlaps the LDS writes below
    double cf[G][MF ? 1 : SPAN];
    if constexpr (!MF) {
#pragma unroll
      for (int mm = 0; mm < SPAN; ++mm)
#pragma unroll
        for (int g = 0; g < G; ++g) cf[g][mm] = cft[(mm * G + g) * T];
    }

    // ---------------------------------------------------------------- stage-1 samples -> LDS
    // (the last FFT pass exchanged nothing, and the exchange before it ended with a barrier)
    {
      const int slot = (int)(B & a.seam_mask);
      double *seamA = a.seam + ((long long)(slot * (a.d.C + 1) + ca) * 2) * 32;
      double *seamB = a.seam + ((long long)(slot * (a.d.C + 1) + cb) * 2) * 32;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int n = tid + s * T;
        if (n < V) {
          // matrix-pipe variant: round A holds samples [0, kSA*T + kPad) only (see below)
          if (!MF || s < kSA || (s == kSA && tid < kPad)) smp[n] = make_double2(v[s].x, v[s].y);
          if (n < nm1) {
            seamA[n] = v[s].x;
            if (hasb) seamB[n] = v[s].y;
          }
          if (n >= V - nm1) {
            seamA[32 + n - (V - nm1)] = v[s].x;
            if (hasb) seamB[32 + n - (V - nm1)] = v[s].y;
          }
        }
      }
      if (tid < kPad) { // finite guard values: padded coefficients are zero, 0 * x must stay 0
        smp[tid - kPad] = make_double2(0.0, 0.0);
        if (!MF || V < kSA * T + kPad) smp[V + tid] = make_double2(0.0, 0.0);
      }
    }
    __syncthreads();
    RSMP_STAMP(4)

    // ---------------------------------------------------------------- polyphase FIR on v_mfma_f64_4x4x4
    // out[r, q, ch] = sum_t c[r][t] * x[ch][w(r) + step*q + t]  (r = output residue mod L, q = period):
    // for a block of 4 consecutive residues the windows start within 3 samples of each other, so with
    // the rows shifted to the block's common start (zero padded to 4*SPAN taps) this is a product
    // A[4 residues x 4 taps] * B[4 taps x 4 periods], accumulated over SPAN k-steps, once per channel.
    // One instruction carries 4 such blocks (16 residues): lane maps measured on gfx950
    // (tools/probe_mfma4.hip): A lane = 16k + 4b + i, B lane = 16k + 4b + j, D lane = 16i + 4b + j.
    // A lane's B operands for both channels are ONE ds_read_b128 of the (A,B)-interleaved LDS samples, feeding
    // two independent accumulation chains (so dependent-MFMA wait states are filled), i.e. 16 bytes of LDS per
    // 8 useful-or-padded FMAs per lane instead of per 4 in the vector version; a lane ends up with both
    // channels of one output frame -> one 8-byte store.  Taps are summed in ascending order like the
    // reference's loop.  The fp64 matrix and vector pipes share hardware (tools/ubench_mfma.hip), so the
    // gain is LDS traffic, issue slots and registers, not peak flops.
    const FusedBlock fb = a.blk[bl];
    if constexpr (MF) {
      constexpr int NW = T / 64;
      const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      const int hi = lane >> 4, bq = (lane >> 2) & 3, jq = lane & 3;
      const int irel_hi = fb.irel_lo + fb.cnt;
      const bool run = !(RSMP_DBGBITS & 1) && fb.cnt > 0;

      bool ofast = false;
      float *obase = nullptr; // frame i_lo's first float of this pair
      int ofs = 2;            // floats between consecutive frames
      {
        const long long o0 = a.out_offset2 + fb.i_lo, o1 = o0 + fb.cnt;
        if (out.is_f32 && hasb && !(out.f.nch & 1)) {
          const int hp = out.f.nch >> 1, strm = pair / hp, pin = pair - strm * hp;
          ofs = out.f.nch;
          if (out.f.ext && o0 >= out.f.ext_begin && o1 <= out.f.ext_end) {
            obase = out.f.ext + strm * out.f.ext_stream_stride + (o0 - out.f.ext_begin) * out.f.nch + 2 * pin;
            ofast = true;
          } else if ((!out.f.ext || o0 >= out.f.ext_end || o1 <= out.f.ext_begin) &&
                     (o0 & out.f.ring_mask) + (o1 - o0) <= out.f.ring_mask + 1) {
            obase = out.f.ring + strm * out.f.ring_stream_stride + (o0 & out.f.ring_mask) * out.f.nch + 2 * pin;
            ofast = true;
          }
          ofast = ofast && (reinterpret_cast<unsigned long long>(obase) & 7) == 0;
        }
      }
      // The whole two-round sequence is instantiated twice and chosen by one uniform branch: FAST writes
      // float2 frames at obase + 32-bit offsets and keeps none of the generic fifo addressing state alive
      // (that state is what used to spill scalar registers inside the loop).
      auto both_rounds = [&](auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const ChanRef oa = chan_ref(out, ca), ob = chan_ref(out, hasb ? cb : ca); // dead code when FAST
        char *const obytes = reinterpret_cast<char *>(obase);
        const int frame_bytes = ofs * 4, period4_bytes = 4 * pl * frame_bytes; // output bytes per frame / per column step

        // A operands are double-buffered: the next item's tile is in flight (L2 latency) while this one computes
        double cn_[SPAN];
        {
          const double *cp = a.cfm + (size_t)(wave >> 1) * SPAN * 64 + lane;
#pragma unroll
          for (int s = 0; s < SPAN; ++s) cn_[s] = cp[s * 64];
        }
        // Outputs of an item are stored at the start of the NEXT item, after that item's wait for its A tile:
        // on gfx950 loads and stores share one in-order counter (vmcnt), so a store issued just before the
        // wait would put a full store round trip into it.
        constexpr int MAXCS = 4; // column steps per item (host: at most 32 periods per block)
        double pA[MAXCS], pB[MAXCS];
        int pend_n = 0;       // column steps waiting to be stored
        int pend_ib = 0;      // this lane's output index (relative to period kk_lo) in the first of them
        int pend_hi = 0;      // exclusive bound of that index: min(irel_hi, ke * pl); -1 when the lane's residue >= pl
        int pend_allv = 0;    // bit u: every lane's output of step u is inside the block's range (uniform)
        int pend_off = 0;     // FAST: byte offset from obase of this lane's frame in the first pending step
        auto store_one = [&](int u) {
          const int ib = pend_ib + u * 4 * pl;
          if (((pend_allv >> u) & 1) || (ib >= fb.irel_lo && ib < pend_hi)) {
            const int orel = ib - fb.irel_lo;
            if (FAST) {
              *reinterpret_cast<float2 *>(obytes + (pend_off + u * period4_bytes)) = make_float2((float)pA[u], (float)pB[u]);
            } else {
              const long long oabs = a.out_offset2 + fb.i_lo + orel;
              fifo_put(oa, oabs, pA[u]);
              if (hasb) fifo_put(ob, oabs, pB[u]);
            }
          }
        };
        auto flush = [&]() {
#pragma unroll
          for (int u = 0; u < MAXCS; u += 2) {
            // Two column steps whose 2 x 64 outputs are all stored, stereo frames: lane rows hi and hi^1 hold adjacent
            // frames, so after a v_permlane16_swap even rows own two adjacent frames of step u and odd rows two of
            // step u+1 -- one 16-byte store per lane instead of two 8-byte ones (a store costs a wave ~390 cycles here).
            if (FAST && ofs == 2 && u + 1 < pend_n && ((pend_allv >> u) & 3) == 3) {
              typedef unsigned u2v __attribute__((ext_vector_type(2)));
              const u2v sA = __builtin_amdgcn_permlane16_swap(__float_as_uint((float)pA[u]), __float_as_uint((float)pA[u + 1]), false, false);
              const u2v sB = __builtin_amdgcn_permlane16_swap(__float_as_uint((float)pB[u]), __float_as_uint((float)pB[u + 1]), false, false);
              const int off = (hi & 1) ? pend_off + (u + 1) * period4_bytes - frame_bytes : pend_off + u * period4_bytes;
              *reinterpret_cast<float4 *>(obytes + off) =
                  make_float4(__uint_as_float(sA.x), __uint_as_float(sB.x), __uint_as_float(sA.y), __uint_as_float(sB.y));
            } else {
              if (u < pend_n) store_one(u);
              if (u + 1 < pend_n) store_one(u + 1);
            }
          }
          pend_n = 0;
        };
        // periods [kb, ke) of the block from the LDS image `xs` (indexed by sample number); li_lo/li_hi clamp
        // the window start of outputs that are not stored anyway (block edges) into the image
        auto poly_round = [&](int kb, int ke, const double2 *xs, int li_lo, int li_hi) {
          const int ncs = (ke - kb + 3) >> 2, half0 = (ncs + 1) >> 1; // column steps of 4 periods
#if RSMP_FINE
#define RSMP_TICK(slot, drain)                                         \
  if (stamping) {                                                       \
    if (drain) __builtin_amdgcn_s_waitcnt(0);                           \
    const unsigned long long now = __builtin_readcyclecounter();       \
    if (tid == 0) atomicAdd(a.stamps + (slot), now - ftick);            \
    ftick = __builtin_readcyclecounter();                               \
  }
          unsigned long long ftick = stamping ? __builtin_readcyclecounter() : 0;
#else
#define RSMP_TICK(slot, drain)
#endif
          for (int it = wave; it < 2 * a.NGRP; it += NW) { // (16-residue group, half of the column steps)
            RSMP_TICK(15, 0)
            const int g = it >> 1, second = (it + (it >> 2)) & 1; // halves alternate so the waves stay balanced
            int cs0 = second ? half0 : 0, cs1 = second ? ncs : half0;
            // column steps whose 64 outputs all lie outside [irel_lo, irel_hi) (block edges) are skipped
            while (cs0 < cs1 && (kb + 4 * cs0 + 3) * pl + 16 * g + 15 < fb.irel_lo) ++cs0;
            while (cs1 > cs0 && (kb + 4 * (cs1 - 1)) * pl + 16 * g >= irel_hi) --cs1;
            RSMP_TICK(8, 0)   // item set-up (skip tests)
            RSMP_TICK(9, 1)   // everything still in flight from the previous item: A tile loads, stores, LDS reads
            double ca_[SPAN];
#pragma unroll
            for (int s = 0; s < SPAN; ++s) ca_[s] = cn_[s];
            flush();
            RSMP_TICK(10, 0)  // conversions + store issue
            if (!(RSMP_DBGBITS & 1024)) {
              const int nx = it + NW < 2 * a.NGRP ? it + NW : wave; // wraps to the first item of the next round
              const double *cp = a.cfm + (nx >> 1) * (SPAN * 64);    // uniform base, lane offset added by the load
#pragma unroll
              for (int s = 0; s < SPAN; ++s) cn_[s] = cp[s * 64 + lane];
            }
            int rb = 16 * g + 4 * bq;
            if (rb >= pl) rb = 0; // idle block: all-zero coefficients, any in-range window will do
            // window start of this lane's block in period kb + jq; later column steps add a uniform 4 * step
            const int qb = (at0 + rb * step) / pl + fb.base_li + hi + (kb + jq) * step;
            const int step4 = 4 * step;
            // B operands of one column step: SPAN ds_read_b128 with immediate offsets off one address; the next
            // step's reads are issued before this step's MFMA chains so their latency hides behind them, into
            // the other of two register images (no copies)
            double2 x0[SPAN], x1[SPAN];
            auto fill = [&](double2 (&x)[SPAN], int cs) {
              const int li = max(li_lo, min(li_hi, qb + cs * step4)); // periods past ke - 1 clamp to li_hi as well
              const double2 *xp = xs + li;
#pragma unroll
              for (int s = 0; s < SPAN; ++s) x[s] = xp[4 * s];
            };
            auto column_step = [&](const double2 (&x)[SPAN], double &accA, double &accB) {
              accA = 0.0;
              accB = 0.0;
#pragma unroll
              for (int s = 0; s < SPAN; ++s) {
                accA = __builtin_amdgcn_mfma_f64_4x4x4f64(ca_[s], x[s].x, accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f64_4x4x4f64(ca_[s], x[s].y, accB, 0, 0, 0);
              }
            };
            if (cs0 < cs1) fill(x0, cs0);
            RSMP_TICK(11, 0)  // A tile load issue, window address (integer division), first fill issue
#if RSMP_FINE
            if (stamping) { __builtin_amdgcn_s_waitcnt(0xc07f); } // lgkmcnt(0) only
#endif
            RSMP_TICK(12, 0)  // first fill's LDS round trip
#pragma unroll
            for (int u = 0; u < MAXCS; ++u) {
              if (cs0 + u < cs1) {
                if (cs0 + u + 1 < cs1 && !(RSMP_DBGBITS & 512)) fill((u & 1) ? x0 : x1, cs0 + u + 1);
                column_step((u & 1) ? x1 : x0, pA[u], pB[u]);
              }
            }
            RSMP_TICK(13, 0)  // column steps
            // bookkeeping for the deferred stores
            const int rD = 16 * g + 4 * bq + hi, k0 = kb + 4 * cs0;
            pend_n = (RSMP_DBGBITS & 16) ? 0 : cs1 - cs0;
            pend_ib = (k0 + jq) * pl + rD;
            pend_off = (pend_ib - fb.irel_lo) * frame_bytes;
            pend_hi = rD < pl ? min(irel_hi, ke * pl) : -1;
            pend_allv = 0;
            if (16 * g + 15 < pl) {
#pragma unroll
              for (int u = 0; u < MAXCS; ++u)
                if (k0 + 4 * u + 3 < ke && (k0 + 4 * u) * pl + 16 * g >= fb.irel_lo && (k0 + 4 * u + 3) * pl + 16 * g + 15 < irel_hi)
                  pend_allv |= 1 << u;
            }
            RSMP_TICK(14, 0)  // store bookkeeping
          }
        };
        // round A: periods whose windows end inside the samples written above
        if (run && !(RSMP_DBGBITS & 64)) poly_round(0, fb.KA, smp, -kPad, min(V, kSA * T) + kPad - 4 * SPAN);
        RSMP_STAMP(6)
        __syncthreads();
        // round B: the rest of the block's samples replace the image, element 0 = sample kSB0*T
        {
          double2 *l2 = reinterpret_cast<double2 *>(lds);
#pragma unroll
          for (int s = kSB0; s < 16; ++s) {
            const int n = tid + s * T;
            if (n < V) l2[n - kSB0 * T] = make_double2(v[s].x, v[s].y);
          }
          if (tid < kPad && V > kSB0 * T) l2[V - kSB0 * T + tid] = make_double2(0.0, 0.0);
        }
        __syncthreads();
        if (run && fb.KA < fb.K && !(RSMP_DBGBITS & 32))
          poly_round(fb.KA, fb.K, reinterpret_cast<const double2 *>(lds) - kSB0 * T, kSB0 * T, V + kPad - 4 * SPAN);
        flush();
      };
      if (ofast) both_rounds(std::true_type{});
      else both_rounds(std::false_type{});
    } else
    // ---------------------------------------------------------------- polyphase FIR from LDS (vector pipe)
    if (!(RSMP_DBGBITS & 1) && poly_thread && fb.cnt > 0) {
      const int irel_hi = fb.irel_lo + fb.cnt;
      const int kper = a.kper; // fixed chunk length (not per block) so that the lane map's bank pattern is static
      const int kr0 = kc * kper, kr1 = min(kr0 + kper, fb.K);

      // output addressing: stereo float frames written as 8/16-byte vectors when the range is contiguous
      bool ofast = false;
      float *obase = nullptr; // points at frame i_lo's first float of this pair
      int ostride = 1;        // float2 elements between consecutive frames
      {
        const long long o0 = a.out_offset2 + fb.i_lo, o1 = o0 + fb.cnt;
        if (out.is_f32 && hasb && !(out.f.nch & 1)) {
          const int hp = out.f.nch >> 1, strm = pair / hp, pin = pair - strm * hp;
          ostride = hp;
          if (out.f.ext && o0 >= out.f.ext_begin && o1 <= out.f.ext_end) {
            obase = out.f.ext + strm * out.f.ext_stream_stride + (o0 - out.f.ext_begin) * out.f.nch + 2 * pin;
            ofast = true;
          } else if ((!out.f.ext || o0 >= out.f.ext_end || o1 <= out.f.ext_begin) &&
                     (o0 & out.f.ring_mask) + (o1 - o0) <= out.f.ring_mask + 1) {
            obase = out.f.ring + strm * out.f.ring_stream_stride + (o0 & out.f.ring_mask) * out.f.nch + 2 * pin;
            ofast = true;
          }
          ofast = ofast && (reinterpret_cast<unsigned long long>(obase) & 7) == 0;
        }
      }

      for (int kr = kr0; kr < kr1; ++kr) {
        const int ib = kr * pl + r0; // output index relative to period kk_lo
        bool ok[G], any = false;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          ok[g] = rv[g] && ib + g >= fb.irel_lo && ib + g < irel_hi;
          any = any || ok[g];
        }
        if (!any) continue;
        const int li = qr0 + kr * step + fb.base_li; // >= -kPad because some output of the tile is interior
        const double2 *xw = smp + li;
        double accA[G], accB[G];
#pragma unroll
        for (int g = 0; g < G; ++g) accA[g] = accB[g] = 0.0;
#pragma unroll
        for (int mm = 0; mm < SPAN; ++mm) {
          const double2 xv = xw[mm];
#pragma unroll
          for (int g = 0; g < G; ++g) {
            accA[g] = fma(cf[g][mm], xv.x, accA[g]);
            accB[g] = fma(cf[g][mm], xv.y, accB[g]);
          }
        }
        if (RSMP_DBGBITS & 16) { if (accA[0] == 12345.678) lds[0] = accB[0] + accA[1] + accB[1]; continue; }
        const int orel = ib - fb.irel_lo; // frame offset from i_lo (-1 for a tile whose first output is a seam output)
        if (ofast) {
          float2 *o2 = reinterpret_cast<float2 *>(obase) + orel * ostride;
          if (G == 2 && ostride == 1 && ok[0] && ok[1] && (reinterpret_cast<unsigned long long>(o2) & 15) == 0) {
            *reinterpret_cast<float4 *>(o2) = make_float4((float)accA[0], (float)accB[0], (float)accA[1], (float)accB[1]);
          } else {
#pragma unroll
            for (int g = 0; g < G; ++g)
              if (ok[g]) o2[g * ostride] = make_float2((float)accA[g], (float)accB[g]);
          }
        } else {
          const ChanRef oa = chan_ref(out, ca), ob = chan_ref(out, hasb ? cb : ca);
          const long long oabs = a.out_offset2 + fb.i_lo + orel;
#pragma unroll
          for (int g = 0; g < G; ++g)
            if (ok[g]) {
              fifo_put(oa, oabs + g, accA[g]);
              if (hasb) fifo_put(ob, oabs + g, accB[g]);
            }
        }
      }
    }
    RSMP_STAMP(5)
    if (stamping && tid == 0) atomicAdd(a.stamps + 7, 1ull);
  }
#undef RSMP_STAMP
}

// Outputs whose window straddles the boundary between block B-1 and block B (or precedes block 0):
// window samples come from the seam ring.  One workgroup = one boundary x 8 channels, 32 lanes per channel;
// the 2*(n-1) samples around the boundary are staged in LDS first so that the tap loop has no dependent
// global loads (the coefficient rows are independent loads, issued 8 taps ahead).
// One workgroup = one boundary x kSeamC channels.  Every channel needs the same coefficient rows (the phase depends on the
// output index only), so the rows of the boundary's outputs are staged in LDS once per workgroup next to the channels'
// [tail | head] windows (one row copy per CHANNEL used to make this kernel an L2-bandwidth problem: 700 MB of row reads per
// launch of the bench workload).  The per-boundary output range and the first output's window / phase come from the block
// table (fused_block_info): no 64-bit divisions here.
constexpr int kSeamC = 64, kSeamWin = 2 * 31 + 3, kSeamOut = 64; // window row padded to an odd number of doubles
__global__ __launch_bounds__(256) void seam_kernel(AnyView out, FusedArgs a)
{
  __shared__ double win[kSeamC][kSeamWin];
  __shared__ double cfs[kSeamOut][32];
  __shared__ int qs[kSeamOut];
  const int bl = blockIdx.x, c0 = blockIdx.y * kSeamC, tid = threadIdx.x;
  const long long B = a.d.B0 + bl;
  const int nm1 = a.n - 1, pl = a.polyL, step = a.step, n = a.n;
  const int nc = min(kSeamC, a.d.C - c0);
  const FusedBlock fb = a.blk[bl];
  const int cnt = min((int)(fb.i_lo - fb.seam_i0), kSeamOut); // at most ~(n - 1) * L / step + 1 outputs per boundary (host-checked <= 64)
  // (no divisions by run-time values in the loops: thread = (row, column) of every tile it touches)
  // (all loads of a phase are issued before the first one is waited for: one memory round trip per phase, not one per row --
  // written as a plain loop the compiler waits for every load before the LDS store behind it, 16 round trips in a row)
  { // [r3] BOTH load phases are issued before anything is waited for (one memory round trip for the kernel's inputs, not two):
    // windows [tail of B-1 | head of B] -- lane k of a 64-lane row, 4 channels per pass, unconditional loads from a clamped
    // address (per-load branches made each of them its own basic block) -- and the coefficient rows of the boundary's outputs
    // -- lane j of a 32-lane row, 8 outputs per pass, only the passes the boundary has outputs for.
    const int k = tid & 63;
    const double *const tails = a.seam + ((long long)((int)((B - 1) & a.seam_mask) * (a.d.C + 1) + c0) * 2 + 1) * 32;
    const double *const heads = a.seam + ((long long)((int)(B & a.seam_mask) * (a.d.C + 1) + c0) * 2) * 32;
    const bool wk = k < 2 * nm1;
    const double *const src = k < nm1 ? tails + k : heads + (min(k, 2 * nm1 - 1) - nm1);
    const bool zero = k < nm1 && B == 0; // (B = 0: the slot read is a valid one of the ring, its value is not used)
    double tw[kSeamC / 4];
#pragma unroll
    for (int i = 0; i < kSeamC / 4; ++i) tw[i] = src[min((tid >> 6) + 4 * i, nc - 1) * 64];
    const int j = min(tid & 31, n - 1);
    double t[kSeamOut / 8];
    int q[kSeamOut / 8];
#pragma unroll
    for (int i = 0; i < kSeamOut / 8; ++i) {
      t[i] = 0.0;
      q[i] = 0;
      if (8 * i < cnt) { // uniform
        const int u = min((tid >> 5) + 8 * i, cnt - 1);
        const unsigned tc = (unsigned)fb.seam_ph0 + (unsigned)u * (unsigned)step; // clock of output seam_i0 + u relative to window seam_q0
        const unsigned qq = tc / (unsigned)pl, ph = tc - qq * (unsigned)pl;
        q[i] = fb.seam_q0 + (int)qq; // window start inside [tail | head], 0 <= . < n-1
        t[i] = a.tab[(long long)ph * n + j];
      }
    }
    if (wk) {
#pragma unroll
      for (int i = 0; i < kSeamC / 4; ++i) {
        const int cl = (tid >> 6) + 4 * i;
        if (cl < nc) win[cl][k] = zero ? 0.0 : tw[i];
      }
    }
#pragma unroll
    for (int i = 0; i < kSeamOut / 8; ++i) {
      const int u = (tid >> 5) + 8 * i;
      if (u < cnt) {
        if ((tid & 31) < n) cfs[u][j] = t[i];
        if ((tid & 31) == 0) qs[u] = q[i];
      }
    }
  }
  __syncthreads();
  { // neighbouring lanes: neighbouring channels of one output; 4 outputs per pass
    const int cl = tid & 63;
    if (cl < nc) {
      const ChanRef oc = chan_ref(out, c0 + cl);
      for (int u = tid >> 6; u < cnt; u += 4) {
        const double *x = win[cl] + qs[u];
        const double *cf = cfs[u];
        double sum = 0.0;
#pragma unroll 8
        for (int j = 0; j < n; ++j) sum = fma(cf[j], x[j], sum); // the reference's tap order
        fifo_put(oc, a.out_offset2 + fb.seam_i0 + u, sum);
      }
    }
  }
}

__global__ __launch_bounds__(256) void fused_prep_kernel(FusedPrepArgs p, FusedBlock *out)
{
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < p.nblocks) out[k] = fused_block_info(p, k);
}

hipError_t launch_fused_prep(const FusedPrepArgs &p, FusedBlock *out, hipStream_t st)
{
  hipLaunchKernelGGL(fused_prep_kernel, dim3((p.nblocks + 255) / 256), dim3(256), 0, st, p, out);
  return hipGetLastError();
}

template <int LOG2N, int LOG2P, int G, int SPAN, bool MF>
static hipError_t launch_fused_t(const AnyView &in, const AnyView &out, const FusedArgs &a, hipStream_t st)
{
  constexpr int N = 1 << LOG2N;
  const size_t lds_pad = knobs().lds_pad; // occupancy experiments
  size_t lds_bytes = 8 * size_t(fft_lds_doubles(LOG2N));
  if (MF) { // half-round exchanges for 4096-point transforms, two-round sample image (fused_kernel, MF part)
    lds_bytes = 8 * size_t(fft_lds_doubles_halves(LOG2N));
    if (LOG2P < LOG2N) lds_bytes = std::max(lds_bytes, 8 * size_t(std::max(fft_lds_doubles(LOG2P), fft8_lds_doubles(LOG2P))));
    lds_bytes = std::max(lds_bytes, size_t(kPad + kSA * (N / 16) + kPad) * 16);
  }
  lds_bytes += lds_pad;
  static DynLdsOnce attr;
  static std::atomic<bool> occ_printed{false};
  if (hipError_t e = attr.set(reinterpret_cast<const void *>(&fused_kernel<LOG2N, LOG2P, G, SPAN, MF>), int(lds_bytes)); e != hipSuccess) return e;
  {
    if (knobs().occ && !occ_printed.exchange(true)) {
      int nb = -1;
      hipError_t eo = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&fused_kernel<LOG2N, LOG2P, G, SPAN, MF>), N / 16, lds_bytes);
      fprintf(stderr, "RSMP_OCC lds %zu blocks/CU %d (%s)\n", lds_bytes, nb, hipGetErrorString(eo));
    }
  }
  FusedArgs b = a;
  b.d.hp = frame_pairs(in, out, a.d.C);
  b.d.npairs = pair_count(a.d.C, a.d.nchs);
  b.d.pps_magic = pair_magic(a.d.C, a.d.nchs);
  dim3 grid(item_grid(a.d.nblocks, b.d.npairs, b.d.hp)), block(N / 16);
  hipLaunchKernelGGL((fused_kernel<LOG2N, LOG2P, G, SPAN, MF>), grid, block, lds_bytes, st, in, out, b);
  return hipGetLastError();
}

hipError_t launch_seam(bool dst_f32, const F32View &df, const F64View &dd, const FusedArgs &a, hipStream_t st)
{
  const AnyView out = make_view(dst_f32, df, dd);
  dim3 sgrid(a.d.nblocks, (a.d.C + kSeamC - 1) / kSeamC), sblock(256);
  hipLaunchKernelGGL(seam_kernel, sgrid, sblock, 0, st, out, a);
  return hipGetLastError();
}

bool fused_mfma_supported(int log2n, int log2p, int ksteps)
{
  return log2n == 12 && (log2p == 11 || log2p == 12) && (ksteps == 7 || ksteps == 8);
}

bool fused_shape_supported(int log2n, int log2p, int n, int span, int max_seam_outputs)
{
  if (log2n < 11 || log2n > 13) return false;
  if (log2p > log2n || log2p < log2n - 2) return false;
  return n >= 4 && n <= 32 && span <= kSpanMax && max_seam_outputs <= 64;
}

#define RSMP_FUSED_CASE(n, p)                                                     \
  if (log2n == n && log2p == p) {                                                 \
    if (kname) *kname = "rsmp::fused_kernel<" #n ", " #p ", 2, 32, false>";       \
    return launch_fused_t<n, p, 2, kSpanMax, false>(in, out, a, st);              \
  }
#define RSMP_FUSED_EXACT(n, p, sp)                                                \
  if (log2n == n && log2p == p && a.span == sp) {                                 \
    if (kname) *kname = "rsmp::fused_kernel<" #n ", " #p ", 2, " #sp ", false>";  \
    return launch_fused_t<n, p, 2, sp, false>(in, out, a, st);                    \
  }
#define RSMP_FUSED_MF(n, p, ks)                                                   \
  if (log2n == n && log2p == p && a.KS == ks) {                                   \
    if (kname) *kname = "rsmp::fused_kernel<" #n ", " #p ", 2, " #ks ", true>";   \
    return launch_fused_t<n, p, 2, ks, true>(in, out, a, st);                     \
  }

hipError_t launch_fused(int log2n, int log2p, bool src_f32, bool dst_f32, const F32View &sf, const F64View &sd,
                        const F32View &df, const F64View &dd, const FusedArgs &a, hipStream_t st, const char **kname)
{
  const AnyView in = make_view(src_f32, sf, sd), out = make_view(dst_f32, df, dd);
  if (a.cfm) { // polyphase on the matrix pipe
    RSMP_FUSED_MF(12, 11, 7) RSMP_FUSED_MF(12, 12, 7) RSMP_FUSED_MF(12, 11, 8) RSMP_FUSED_MF(12, 12, 8)
    return hipErrorInvalidValue;
  }
  // exact-window variants for the chains the plugin's rate matrix produces at Best (24 taps/phase)
  RSMP_FUSED_EXACT(12, 11, 25) RSMP_FUSED_EXACT(12, 11, 26) RSMP_FUSED_EXACT(12, 12, 26) RSMP_FUSED_EXACT(12, 12, 27)
  // generic variants: window padded to kSpanMax
  RSMP_FUSED_CASE(11, 11) RSMP_FUSED_CASE(11, 10) RSMP_FUSED_CASE(11, 9)
  RSMP_FUSED_CASE(12, 12) RSMP_FUSED_CASE(12, 11) RSMP_FUSED_CASE(12, 10)
  RSMP_FUSED_CASE(13, 13) RSMP_FUSED_CASE(13, 12) RSMP_FUSED_CASE(13, 11)
  return hipErrorInvalidValue;
}

} // namespace rsmp
