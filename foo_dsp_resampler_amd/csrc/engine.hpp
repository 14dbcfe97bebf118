// Stream engine: owns the device-resident fifos of one handle (S streams x nch channels, all
// advancing in lock step), mirrors the reference's fifo/stage accounting on the host with integers
// only (rate/rate_base.h:425-468, rate/dft_filter.h:78-84, rate/rate_filters_generic.h:275-304), and
// turns every push into a short sequence of kernel launches on one HIP stream.
//
// No sample data is ever inspected on the host: how many frames each stage can produce after a push
// is a pure function of the counters, so everything is enqueued asynchronously and the host only
// synchronises when the caller needs bytes back (pull to host memory).
#pragma once
#include "kernels.hpp"
#include "plan.hpp"

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace rsmp {

enum { kOk = 0, kNoMem = 1, kInternal = 2, kNullHandle = 3, kRateError = 4, kUninit = 5, kInvParam = 6 };

// integer-only mirror of the reference's per-channel state (identical for every channel)
struct Book {
  std::vector<long long> wr, rd; // per fifo (num_stages + 1), absolute counts
  struct St { long long B = 0; int remL = 0, remM = 0; long long at = 0; };
  std::vector<St> st;
  size_t samples_in = 0, samples_out = 0; // rate_base.h:226
  long long trimmed = 0; // frames fifo_trim_to removed from the last fifo so far (rate_base.h:465): stage outputs are
                         // indexed absolutely, so the last fifo's positions lag them by this much after a drain
};

// Makes `device` the calling thread's current HIP device for the lifetime of the object and puts the previous one back
// afterwards.  Every C-ABI entry that takes a handle runs under one (capi.cpp): a handle lives on the device it was opened
// on, whatever device the calling thread has selected since (the plugin is ONE process whose converter threads each own
// handles, chain.h:36; with RATELIB_AMD_DEVICES / RRX_open_batch_on those handles may sit on different GPUs).
// Host-only (no device state): can dft stage i of `plan` and the polyphase stage behind it run as the sub-blocked fused kernel
// for handles of `nch` channels per stream, and with what geometry (Engine::split_geometry; RRX_describe_dispatch)?
bool split_geometry(const ChainPlan &plan, int nch, int i, int &nsub, int &vs);

class DeviceScope {
public:
  explicit DeviceScope(int device)
  {
    if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
    if (device >= 0 && device != prev_) ok_ = hipSetDevice(device) == hipSuccess, switched_ = ok_;
  }
  ~DeviceScope() { if (switched_ && prev_ >= 0) (void)hipSetDevice(prev_); }
  DeviceScope(const DeviceScope &) = delete;
  DeviceScope &operator=(const DeviceScope &) = delete;
  bool ok() const { return ok_; }
private:
  int prev_ = -1;
  bool ok_ = true, switched_ = false;
};

class Engine {
public:
  // device: HIP device index the handle lives on; -1 = the calling thread's current device.  kInvParam for an index the
  // process does not have, kUninit for a device that is not gfx950.
  static int create(const Config &cfg, int nch, int nstreams, int device, Engine **out);
  ~Engine();

  int device() const { return device_; }

  const ChainPlan &plan() const { return plan_; }
  int nch() const { return nch_; }
  int nstreams() const { return S_; }
  // batch handles with an odd channel count per stream: channel pairs stay inside a stream (pair_channels), so that every
  // stream gets the bits its own handle would give
  int pair_nchs() const { return (S_ > 1 && (nch_ & 1)) ? nch_ : 0; }
  size_t isamp_max() const { return plan_.isamp_max; }
  size_t available() const { return size_t(book_.wr.back() - book_.rd.back()); }

  // Use the caller's stream for everything from now on (nullptr = the default stream, like any HIP call; own = true:
  // back to the handle's own stream).  Work already queued on the previous stream is ordered before anything queued
  // later (event on the old stream, wait on the new).  The caller's stream is never destroyed by the handle.
  int set_stream(hipStream_t s, bool own = false);
  hipStream_t stream() const { return stream_; }
  int sync();
  // test hook: make the n-th device allocation from now on (n >= 1, process-wide) fail with hipErrorOutOfMemory; 0 disarms
  static void fail_alloc_after(int n);

  // Host-memory API (RR_push / RR_pull / RR_flow semantics). Buffers: [stream][frame][channel] with
  // `stream_stride` frames between streams (ignored when there is one stream).
  int push_host(const float *ibuf, size_t stream_stride, size_t isamp);
  int pull_host(float *obuf, size_t stream_stride, size_t osamp, size_t *ogen);
  int flow_host(const float *ibuf, size_t in_stride, float *obuf, size_t out_stride, size_t isamp, size_t osamp,
                size_t *iused, size_t *ogen);
  // Device-memory API: same semantics, pointers are HBM addresses, nothing is synchronised.
  int push_device(const float *ibuf, size_t stream_stride, size_t isamp);
  int pull_device(float *obuf, size_t stream_stride, size_t osamp, size_t *ogen);
  int flow_device(const float *ibuf, size_t in_stride, float *obuf, size_t out_stride, size_t isamp, size_t osamp,
                  size_t *iused, size_t *ogen);
  int drain();

  // optional per-kernel timing: HIP events recorded on the launch stream around every stage launch
  void set_profiling(bool on);
  int read_profile(double *hot_ms, long long *hot_launches, double *other_ms, long long *other_launches);
  // same records aggregated per kernel, as JSON: [{"kernel": "...", "hot": 0|1, "launches": n, "ms": t}, ...]; clears them
  int read_profile_json(std::string &out);

private:
  Engine() = default;
  int init(const Config &cfg, int nch, int nstreams);

  struct Ring { // device ring of fifo f
    void *buf = nullptr;
    long long cap = 0; // items (f64) or frames (f32), power of two
    bool f32 = false;
  };
  struct ExtIn { const float *ptr = nullptr; long long begin = 0, end = 0, stride_floats = 0; };
  struct ExtOut { float *ptr = nullptr; long long begin = 0, end = 0, stride_floats = 0; };

  // keep_direct: frames written straight to d_out stay in the output fifo (not counted as pulled): the host mirror
  int feed(const float *d_in, size_t stride_frames, size_t isamp, float *d_out, size_t out_stride, size_t out_cap,
           size_t *direct_out, bool keep_direct = false);
  int feed_impl(const float *d_in, size_t stride_frames, size_t isamp, float *d_out, size_t out_stride, size_t out_cap,
                size_t *direct_out, bool keep_direct);
  // more_slabs: another time slab of the same push follows (seam kernels may then run beside the next slab's launches)
  int advance(Book &b, size_t n_new, bool launch, const ExtIn &ein, const ExtOut &eout, bool more_slabs = false);
  // a dft stage that is fused with the polyphase stage behind it leaves its launch to that stage
  struct Pending { long long B0 = 0; int nblocks = 0; DftArgs args = {}; int log2n = 0, log2p = 0; };
  // one pass of rate_process over the chain: what advance() hands to the per-stage functions
  struct Pass { Book &b; bool launch; const ExtIn &ein; const ExtOut &eout; bool more_slabs; Pending pend; };
  int advance_dft(Pass &ps, int i);
  int advance_poly(Pass &ps, int i);
  int advance_half(Pass &ps, int i);
  int launch_fused_pair(Pass &ps, int i, long long count, long long step);
  bool split_geometry(int i, int &nsub, int &vs) const;
  int launch_polymf_stage(Pass &ps, int i, long long count, long long step);
  int launch_poly_stage(Pass &ps, int i, long long count, long long step);
  int ensure_ring(int f, long long live_needed);
  int copy_out(float *dst, size_t stride_frames, size_t frames, bool to_host);
  F32View f32_view(int f, const ExtIn *ein, const ExtOut *eout) const;
  F64View f64_view(int f) const;
  void note_input(Book &b, size_t n) const;
  int upload(const void *src, size_t bytes, void **dst);
  const double2 *twiddles(int log2m);
  const double2 *twiddles8(int log2m);
  void free_garbage();

  ChainPlan plan_;
  int nch_ = 0, S_ = 0, C_ = 0;
  int device_ = 0; // HIP device of every allocation, stream, event and launch of this handle (recorded by create)
  hipStream_t stream_ = nullptr; // where work is queued: own_ or the caller's (RRX_set_stream)
  hipStream_t own_ = nullptr;    // created by init, destroyed by the destructor; never the caller's
  hipEvent_t ev_switch_ = nullptr;
  // A failure after the counters of a push / drain started moving leaves them skewed against the device fifos:
  // the handle is then poisoned and every later data call returns kInternal (rate_uni.c has no recovery either).
  bool poisoned_ = false;
  int fail(int rc) { if (rc != kOk) poisoned_ = true; return rc; }
  int dev_alloc(void **p, size_t bytes);
  // side stream for the seam kernels: seam(k) only depends on fused(k), so it runs beside fused(k+1)
  hipStream_t side_ = nullptr;
  hipEvent_t ev_fused_ = nullptr, ev_seam_[2] = {nullptr, nullptr};
  long long seam_launches_ = 0; // seam launch k records ev_seam_[k & 1]
  bool side_pending_ = false;
  int join_side();
  Book book_;
  std::vector<Ring> rings_;
  std::vector<void *> garbage_;
  // device tables
  double2 *d_G_[2] = {nullptr, nullptr};
  double2 *d_Gr_[2] = {nullptr, nullptr}; // dftx_kernel: spectra of the polyphase components of the same filters
  double *d_poly_ = nullptr;
  double2 *d_tw_[20] = {};
  double2 *d_tw8_[20] = {}; // 8-points-per-thread plans (fft8_regs)
  // staging for host pushes / drains
  // dft stages with the reference's long blocks (N >= 32768): four-step transform through a workspace (dftbig.hip)
  struct BigDft { bool on = false; double2 *twN = nullptr, *w1 = nullptr, *w2 = nullptr; int ws_items = 0; };
  std::vector<BigDft> big_;         // indexed by stage
  // fused dft->vpoly0 path
  struct Fuse { bool on = false; int span = 0, NG = 0, KC = 0, kper = 0; double *seam = nullptr; double *cft = nullptr; int slots = 0;
                double *cfm = nullptr; int NGRP = 0, KS = 0, qb_max = 0, qb_min = 0; int *qtab = nullptr; double2 *cfm2 = nullptr;
                FusedBlock *blk_dev = nullptr; int blk_cap = 0;
                // sub-blocked form (fused_fast_kernel<.., SPLIT>): nsub sub-blocks of Vs samples per block, component spectra
                int nsub = 0, Vs = 0; double2 *Gs = nullptr; };
  std::vector<Fuse> fuse_;            // indexed by the dft stage
  // standalone matrix-pipe polyphase stage (polymf.hip), indexed by the poly stage
  struct PolyMf { double *cfm = nullptr; int *qtab = nullptr; FusedBlock *blk = nullptr; int KS = 0, NGRP = 0, Vt = 0, blk_cap = 0; };
  std::vector<PolyMf> polymf_;
  struct ProfRec { hipEvent_t e0, e1; bool hot; const char *name; };
  std::vector<ProfRec> prof_;
  bool profiling_ = false;
  int prof_begin(bool hot, const char *name = "");
  void prof_name(int idx, const char *name) { if (idx >= 0 && name) prof_[idx].name = name; }
  void prof_end(int idx);
  int dbg_ = 0;           // RSMP_DBG ablation bits (0 in production)
  bool no_side_ = false;  // RSMP_NO_SIDE: keep seam kernels on the main stream
  unsigned long long *stamps_ = nullptr; // RSMP_STAMPS: device buffer of per-phase cycle sums
  float *d_stage_ = nullptr;
  size_t stage_floats_ = 0;
  // Pinned host staging for RR_push / RR_pull (the plugin's 1-8 k-frame chunks): the caller's pageable buffer is copied
  // into page-locked memory on the CPU, so the H2D copy is a true asynchronous DMA and a push returns without waiting for
  // the device; two input slots alternate, each guarded by an event recorded behind its copy.  Pushes larger than
  // kPinnedMaxBytes keep the direct (runtime-staged) path.
  struct Pinned { float *p = nullptr; size_t floats = 0; hipEvent_t done = nullptr; bool pending = false; };
  Pinned pin_in_[2], pin_out_;
  int pin_k_ = 0;
  static constexpr size_t kPinnedMaxBytes = size_t(64) << 20;
  // Plugin-sized pushes (foo_dsp_rate.cpp:182-202: 1-8 k frames, push then pull-until-empty) skip both copy commands:
  //  * the kernels read the push straight out of the page-locked slot (it is device-visible), no H2D copy;
  //  * when the output fifo is empty at the push, the last stage writes straight into a page-locked "mirror" of the frames
  //    it produces ([mir_begin_, mir_end_) of the output fifo, absolute indices); RR_pull then waits for the push's last
  //    kernel and copies from the mirror on the CPU, no D2H copy.  Frames still in the mirror when something other than a
  //    host pull wants them (another push, a device pull, a request past its end) are first spilled into the device ring.
  static constexpr size_t kZeroCopyMaxBytes = size_t(1) << 20;
  Pinned pin_mir_;
  long long mir_begin_ = 0, mir_end_ = 0; // frames of the output fifo held by the mirror (empty: begin == end)
  size_t mir_stride_ = 0;                 // frames between streams in the mirror
  int spill_mirror();
  int pinned_reserve(Pinned &b, size_t floats);
  void pinned_free(Pinned &b);
  size_t slab_frames_ = 0;
};

bool device_is_gfx950(int device = -1); // the device (-1: the current one) runs gfx950 code objects

} // namespace rsmp
