// Stream engine. See engine.hpp.
#include "engine.hpp"
#include <cstdio>

#include "design.hpp"

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>

namespace rsmp {

namespace {

bool pow2_ge2(int x) { return x >= 2 && !(x & (x - 1)); }
// x2 / x4 upsampling runs in the frequency domain (forward transform of N/L points, spectrum replicated: dft_filter.h:86-104).
// Larger powers of two (x32 and beyond as a whole: e.g. 8000 -> 352800) take the kernels' time-domain zero-stuffing
// branch instead: the reference's replication IS the spectrum of the block zero-stuffed from slot 0, so the two agree
// to fp64 rounding; the reference's counters (remL untouched on this branch) are mirrored as they are.
bool fdomain_up(int L) { return pow2_ge2(L) && L <= 4; }
int ilog2(long long v) { int l = 0; while ((1LL << l) < v) ++l; return l; }
long long next_pow2(long long v) { long long p = 1; while (p < v) p <<= 1; return p; }

#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t e_ = (expr);                            \
    if (e_ != hipSuccess) return e_ == hipErrorOutOfMemory ? kNoMem : kInternal; \
  } while (0)

// host mirror of fft_device.hpp's schedule
} // namespace

const Knobs &knobs()
{
  static const Knobs k = [] {
    Knobs r;
    auto on = [](const char *name) { return getenv(name) != nullptr; };
    r.no_fuse = on("RSMP_NO_FUSE");
    r.no_mfma = on("RSMP_NO_MFMA");
    r.no_polymf = on("RSMP_NO_POLYMF");
    r.no_fast = on("RSMP_NO_FAST");
    r.no_split = on("RSMP_NO_SPLIT");
    r.no_split2 = on("RSMP_NO_SPLIT2");
    r.no_dftx = on("RSMP_NO_DFTX");
    r.no_polyi = on("RSMP_NO_POLYI");
    r.no_polycoop = on("RSMP_NO_POLYCOOP");
    r.spread_vector = on("RSMP_SPREAD_VECTOR");
    r.no_side = on("RSMP_NO_SIDE");
    r.no_graph = on("RSMP_NO_GRAPH");
    r.stamps = on("RSMP_STAMPS");
    r.occ = on("RSMP_OCC");
    r.test_hooks = on("RSMP_TEST_HOOKS");
    if (const char *v = getenv("RSMP_SLAB_MB")) r.slab_mb = atof(v) > 0 ? atof(v) : r.slab_mb;
    if (const char *v = getenv("RSMP_SEAM_RING_MB")) r.seam_ring_mb = atof(v) > 0 ? atof(v) : r.seam_ring_mb;
    if (const char *v = getenv("RSMP_LDS_PAD")) r.lds_pad = size_t(std::max(0, atoi(v)));
#ifdef RSMP_EXPERIMENTS
    if (const char *v = getenv("RSMP_DBG")) r.dbg = atoi(v);
#endif
    return r;
  }();
  return k;
}

// true when the current device can run this library's code object (built with --offload-arch=gfx950 only)
bool device_is_gfx950(int device)
{
  int dev = device;
  hipDeviceProp_t prop;
  if ((dev < 0 && hipGetDevice(&dev) != hipSuccess) || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
  return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

namespace {

int first_radix(int log2m) { return (log2m & 3) ? (1 << (log2m & 3)) : 16; }
int num_passes(int log2m) { return (log2m + 3) / 4; }

} // namespace

int Engine::create(const Config &cfg, int nch, int nstreams, int device, Engine **out)
{
  if (!out) return kInvParam;
  *out = nullptr;
  if (nch < 1 || nstreams < 1) return kInvParam;
  if ((long long)nch * nstreams > INT_MAX / 4096) return kNoMem; // parameter guard: more channels than any device could hold fifos for
  int dev_count = 0;
  if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count < 1) return kUninit; // no HIP device: fail loudly
  if (device < 0 && hipGetDevice(&device) != hipSuccess) return kUninit;
  if (device >= dev_count) return kInvParam; // a device this process does not have
  DeviceScope on(device);
  if (!on.ok()) return kInternal;
  Engine *e = new (std::nothrow) Engine();
  if (!e) return kNoMem;
  e->device_ = device;
  int rc = e->init(cfg, nch, nstreams);
  if (rc != kOk) {
    delete e;
    return rc;
  }
  *out = e;
  return kOk;
}

namespace { std::atomic<int> g_fail_alloc{0}; }

// honoured only when the process was started with RSMP_TEST_HOOKS set (knobs.hpp): a production process cannot arm it
void Engine::fail_alloc_after(int n) { g_fail_alloc.store(knobs().test_hooks && n > 0 ? n : 0); }

// every device allocation of the engine goes through here (RR_ENOMEM mapping, test failpoint)
int Engine::dev_alloc(void **p, size_t bytes)
{
  *p = nullptr;
  for (int cur = g_fail_alloc.load(std::memory_order_relaxed); cur > 0;) // one atomic step per allocation (compare-exchange countdown)
    if (g_fail_alloc.compare_exchange_weak(cur, cur - 1)) {
      if (cur == 1) return kNoMem;
      break;
    }
  const hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess) return kOk;
  (void)hipGetLastError(); // the failure is reported through the return code; do not leave it sticky
  *p = nullptr;
  return e == hipErrorOutOfMemory ? kNoMem : kInternal;
}

#define ALLOC_TRY(ptr, bytes)                                             \
  do {                                                                    \
    int rc_ = dev_alloc(reinterpret_cast<void **>(ptr), (bytes));         \
    if (rc_ != kOk) return rc_;                                           \
  } while (0)

int Engine::upload(const void *src, size_t bytes, void **dst)
{
  ALLOC_TRY(dst, bytes);
  HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return kOk;
}

const double2 *Engine::twiddles(int log2m)
{
  if (d_tw_[log2m]) return d_tw_[log2m];
  // [pass >= 1][r-1][k] = exp(+2 pi i r k / (16 Ns)); see fft_device.hpp
  std::vector<double2> tab;
  long long ns = first_radix(log2m);
  for (int p = 1; p < num_passes(log2m); ++p, ns *= 16)
    for (int r = 1; r < 16; ++r)
      for (long long k = 0; k < ns; ++k) {
        const long double th = 2.0L * 3.14159265358979323846264338327950288L * (long double)(r * k) / (long double)(16 * ns);
        tab.push_back(make_double2((double)cosl(th), (double)sinl(th)));
      }
  if (tab.empty()) tab.push_back(make_double2(1, 0));
  void *d = nullptr;
  if (upload(tab.data(), tab.size() * sizeof(double2), &d) != kOk) return nullptr;
  d_tw_[log2m] = static_cast<double2 *>(d);
  return d_tw_[log2m];
}

const double2 *Engine::twiddles8(int log2m)
{
  if (d_tw8_[log2m]) return d_tw8_[log2m];
  // radix-8 passes with a final radix 8 / 4 / 2: [pass >= 1][r-1][k] = exp(+2 pi i r k / (R Ns)); see fft8_regs
  std::vector<double2> tab;
  const int np = (log2m + 2) / 3, rl = (log2m % 3) ? (1 << (log2m % 3)) : 8;
  long long ns = 8;
  for (int p = 1; p < np; ++p, ns *= 8) {
    const int R = p + 1 == np ? rl : 8;
    for (int r = 1; r < R; ++r)
      for (long long k = 0; k < ns; ++k) {
        const long double th = 2.0L * 3.14159265358979323846264338327950288L * (long double)(r * k) / (long double)(R * ns);
        tab.push_back(make_double2((double)cosl(th), (double)sinl(th)));
      }
  }
  if (tab.empty()) tab.push_back(make_double2(1, 0));
  void *d = nullptr;
  if (upload(tab.data(), tab.size() * sizeof(double2), &d) != kOk) return nullptr;
  d_tw8_[log2m] = static_cast<double2 *>(d);
  return d_tw8_[log2m];
}

// Can dft stage i and the rational polyphase stage behind it run as the sub-blocked fused kernel (fused_split_kernel)?  If
// so: sub-blocks per block and valid samples per sub-block.  The kernel has no generic form behind it, so everything it
// needs is decided here, once, from the plan: x2 stage first in the chain (float frames in), even channel count, blocks of
// 8192 ... 32768 points whose filter leaves >= 1024 valid samples in a pair of 4096-point component transforms, matrix-pipe tiles.
bool Engine::split_geometry(int i, int &nsub, int &vs) const { return rsmp::split_geometry(plan_, nch_, i, nsub, vs); }

bool split_geometry(const ChainPlan &plan_, int nch_, int i, int &nsub, int &vs)
{
  nsub = vs = 0;
  const Knobs &kn = knobs();
  const int ns = int(plan_.stages.size());
  if (i != 0 || i + 1 >= ns || kn.no_fuse || kn.no_mfma || (nch_ & 1)) return false;
  const StageSpec &d = plan_.stages[i], &p = plan_.stages[i + 1];
  if (d.kind != StageKind::Dft || p.kind != StageKind::Poly || p.order != 0 || d.step != 1 || d.L != 2 || d.remL0 != 0) return false;
  const DftFilter &f = plan_.dft[d.filt];
  const int log2n = ilog2(f.N), V = f.N - (f.num_taps - 1), Pref = f.N / 2;
  const int pstep = int(p.step64 >> 32), at0 = int(p.at0 >> 32);
  int d4 = 0;
  for (int rb = 0; rb < p.L; rb += 4) {
    const long long a0 = at0 + (long long)rb * pstep, a1 = at0 + (long long)std::min(rb + 3, p.L - 1) * pstep;
    d4 = std::max(d4, int(a1 / p.L - a0 / p.L));
  }
  const int KS = std::max(7, (p.n + d4 + 3) / 4);
  const int max_seam = int(((long long)(p.n - 1) * p.L + pstep - 1) / pstep) + 1;
  if ((V & 1) || p.L < 64 || p.n > 32 || max_seam > 64 || !fused_split_supported(log2n, d.L, KS)) return false;
  if (Pref == 4096) { // 8192-point blocks: the whole block from ONE pair of component transforms, polyphase stage in two rounds
    int qb_min = at0 / p.L, qb_max = qb_min;
    for (int rb = 0; rb < p.L; rb += 4) qb_max = std::max(qb_max, int((at0 + (long long)rb * pstep) / p.L));
    if (fused_split_two_supported(V, f.num_taps, KS, qb_max - qb_min) && V >= 2 * p.n) {
      nsub = 1;
      vs = V;
      return true;
    }
  }
  // the longest sub-block the component transforms leave valid (and the LDS image holds), then an even split
  const int vmax = std::min(kSplitVsMax, (2 * 4096 - (f.num_taps - 1)) & ~1);
  if (vmax < 1024) return false;
  const int n_ = (V + vmax - 1) / vmax, v_ = (((V + n_ - 1) / n_) + 1) & ~1;
  if ((n_ - 1) * v_ >= V) return false; // (cannot happen for V >> nsub)
  for (int k = 0; k < n_; ++k) {
    const SubBlock sb = sub_block(k, V, v_, Pref);
    if ((sb.len & 1) || sb.len < 2 * p.n || 2 * sb.shift + sb.len + (f.num_taps - 1) > 8192) return false;
  }
  nsub = n_;
  vs = v_;
  return true;
}

int Engine::init(const Config &cfg, int nch, int nstreams)
{
  int rc = make_plan(cfg, plan_);
  if (rc) return rc;
  nch_ = nch;
  S_ = nstreams;
  C_ = nch * nstreams;

  if (!device_is_gfx950(device_)) return kUninit; // the code object is built for gfx950 only: refuse here, not at the first launch
  HIP_TRY(hipStreamCreateWithFlags(&own_, hipStreamNonBlocking));
  stream_ = own_;
  HIP_TRY(hipEventCreateWithFlags(&ev_switch_, hipEventDisableTiming));
  HIP_TRY(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&ev_fused_, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&ev_seam_[0], hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&ev_seam_[1], hipEventDisableTiming));

  const Knobs &kn = knobs(); // the environment was read once per process; nothing below or on the launch path calls getenv
  dbg_ = kn.dbg;
  no_side_ = kn.no_side;
  if (kn.stamps) {
    ALLOC_TRY(&stamps_, 16 * sizeof(unsigned long long));
    HIP_TRY(hipMemset(stamps_, 0, 16 * sizeof(unsigned long long)));
  }
  const int ns = int(plan_.stages.size());
  book_.wr.assign(ns + 1, 0);
  book_.rd.assign(ns + 1, 0);
  book_.st.assign(ns, Book::St());
  rings_.assign(ns + 1, Ring());
  big_.assign(ns, BigDft());
  for (int i = 0; i <= ns; ++i) rings_[i].f32 = (i == 0 || i == ns);

  double bytes_per_in_frame = 0, rate = 1;
  for (int i = 0; i < ns; ++i) {
    const StageSpec &sp = plan_.stages[i];
    book_.wr[i] = sp.preload; // rate_base.h:417-422
    Book::St &st = book_.st[i];
    if (sp.kind == StageKind::Dft) {
      const DftFilter &f = plan_.dft[sp.filt];
      const int Ng = f.N;
      const int log2n = ilog2(Ng);
      const int log2p = fdomain_up(sp.L) ? log2n - ilog2(sp.L) : log2n;
      const int log2nd = sp.step < 0 ? log2n + sp.step : log2n;
      int sub_n = 0, sub_v = 0;
      const bool sub = split_geometry(i, sub_n, sub_v); // runs as fused_split_kernel: no transform of the block's own length
      const bool big = log2n > 14 && !sub; // the reference's long blocks: four-step transform (dftbig.hip)
      if (!sub && (big ? !big_dft_supported(log2n, log2p, log2nd) : !dft_shape_supported(log2n, log2p, log2nd))) return kInvParam;
      st.remL = sp.remL0;
      if (!d_G_[sp.filt]) { // G = DFT_N(L * h placed at (i + N - taps + 1) mod N) / N, rate_base.h:173-175
        std::vector<cplx> g(Ng);
        for (int i2 = 0; i2 < f.num_taps; ++i2) g[(i2 + Ng - f.num_taps + 1) & (Ng - 1)] = f.taps[i2] * sp.L;
        fft_inplace(g, -1);
        std::vector<double2> G(Ng);
        for (int k = 0; k < Ng; ++k) G[k] = make_double2(g[k].real() / Ng, g[k].imag() / Ng);
        void *d = nullptr;
        if ((rc = upload(G.data(), G.size() * sizeof(double2), &d)) != kOk) return rc;
        d_G_[sp.filt] = static_cast<double2 *>(d);
      }
      if (!big && !d_Gr_[sp.filt] && sp.step == 1 && fdomain_up(sp.L) && dftx_supported(log2n, log2p, log2nd)) {
        // G_r = DFT_P(L * h_placed[L j + r]) / P, r < L: the block's L polyphase components (dftx.hip)
        const int Lx = sp.L, Px = Ng / Lx;
        std::vector<double2> Gr(size_t(Lx) * Px);
        for (int r = 0; r < Lx; ++r) {
          std::vector<cplx> g(Px);
          for (int i2 = 0; i2 < f.num_taps; ++i2) {
            const int m = (i2 + Ng - f.num_taps + 1) & (Ng - 1);
            if (m % Lx == r) g[m / Lx] = f.taps[i2] * sp.L;
          }
          fft_inplace(g, -1);
          for (int k = 0; k < Px; ++k) Gr[size_t(r) * Px + k] = make_double2(g[k].real() / Px, g[k].imag() / Px);
        }
        void *d = nullptr;
        if ((rc = upload(Gr.data(), Gr.size() * sizeof(double2), &d)) != kOk) return rc;
        d_Gr_[sp.filt] = static_cast<double2 *>(d);
      }
      if (big) {
        BigDft &bg = big_[i];
        bg.on = true;
        if (!twiddles(log2p - 4) || !twiddles(log2nd - 4)) return kNoMem;
        std::vector<double2> tw(Ng);
        for (int j = 0; j < Ng; ++j) {
          const long double th = 2.0L * 3.14159265358979323846264338327950288L * (long double)j / (long double)Ng;
          tw[j] = make_double2((double)cosl(th), (double)sinl(th));
        }
        void *d = nullptr;
        if ((rc = upload(tw.data(), tw.size() * sizeof(double2), &d)) != kOk) return rc;
        bg.twN = static_cast<double2 *>(d);
        // (block, pair) items in flight per round of the three launches: at most 256 MB per workspace
        const int npairs = pair_count(C_, pair_nchs());
        bg.ws_items = int(std::max<long long>(1, std::min<long long>((256LL << 20) / (16LL * Ng), std::max(4 * npairs, 16))));
        ALLOC_TRY(&bg.w1, size_t(bg.ws_items) * (size_t(1) << log2p) * sizeof(double2));
        ALLOC_TRY(&bg.w2, size_t(bg.ws_items) * (size_t(1) << log2nd) * sizeof(double2));
      } else if (!sub && (!twiddles(log2p) || !twiddles(log2nd))) return kNoMem;
      double r = double(sp.L);
      if (sp.step > 0) r /= sp.step; else r /= double(1 << -sp.step);
      rate *= r;
    } else if (sp.kind == StageKind::Poly) {
      st.at = sp.order == 0 ? (sp.at0 >> 32) : sp.at0;
      if (!d_poly_) {
        void *d = nullptr;
        if ((rc = upload(plan_.poly_table.data(), plan_.poly_table.size() * sizeof(double), &d)) != kOk) return rc;
        d_poly_ = static_cast<double *>(d);
      }
      rate *= sp.out_in_ratio;
    } else {
      rate *= 0.5;
    }
    if (i + 1 < ns) bytes_per_in_frame += rate * 8.0 * C_;
  }
  // dft(step 1, L = 1,2,4) directly followed by a rational polyphase stage runs as ONE kernel (fused.hip)
  fuse_.assign(ns, Fuse());
  size_t fused_slab_cap = 0;
  for (int i = 0; i + 1 < ns; ++i) {
    const StageSpec &d = plan_.stages[i], &p = plan_.stages[i + 1];
    if (d.kind != StageKind::Dft || p.kind != StageKind::Poly || p.order != 0 || d.step != 1) continue;
    if (!(d.L == 1 || (pow2_ge2(d.L) && d.L <= 4))) continue;
    const DftFilter &f = plan_.dft[d.filt];
    const int log2n = ilog2(f.N), log2p = log2n - (d.L == 1 ? 0 : ilog2(d.L));
    const int G = 2, pstep = int(p.step64 >> 32), at0 = int(p.at0 >> 32);
    int dmax = 0;
    const int NG = (p.L + G - 1) / G;
    for (int m = 0; m < NG; ++m) {
      const long long a0 = at0 + (long long)(G * m) * pstep, a1 = at0 + (long long)std::min(G * m + G - 1, p.L - 1) * pstep;
      dmax = std::max(dmax, int(a1 / p.L - a0 / p.L));
    }
    const int max_seam = int(((long long)(p.n - 1) * p.L + pstep - 1) / pstep) + 1;
    // Blocks too long for one workgroup (8192 / 16384 points): the sub-blocked form of the lean kernel, for the chains it
    // covers -- x2 stage first in the chain (float frames in), even channel count, matrix-pipe tiles; the polyphase stage
    // either last (float frames out) or feeding a further stage (fp64 ring out).  There is no generic kernel behind it, so
    // everything is decided here.
    int split_nsub = 0, split_vs = 0;
    split_geometry(i, split_nsub, split_vs);
    const bool split = split_nsub > 0;
    const int threads = split ? 256 : f.N / 16;
    if (!split && (NG > threads || !fused_shape_supported(log2n, log2p, p.n, p.n + dmax, max_seam))) continue;
    if (kn.no_fuse) continue;
    Fuse &fu = fuse_[i];
    fu.on = true;
    fu.nsub = split_nsub;
    fu.Vs = split_vs;
    if (split) { // G_r = DFT_4096(L * h_placed[2 j + r]) / 4096 with h placed at -(taps - 1) .. 0 mod 8192: the block's two components
      if (!twiddles(12)) return kNoMem;
      std::vector<double2> Gs(size_t(2) * 4096);
      for (int r = 0; r < 2; ++r) {
        std::vector<cplx> g(4096);
        for (int i2 = 0; i2 < f.num_taps; ++i2) {
          const int m = (i2 + 8192 - f.num_taps + 1) & 8191;
          if ((m & 1) == r) g[m >> 1] = f.taps[i2] * d.L;
        }
        fft_inplace(g, -1);
        for (int k = 0; k < 4096; ++k) Gs[size_t(r) * 4096 + k] = make_double2(g[k].real() / 4096, g[k].imag() / 4096);
      }
      void *dg = nullptr;
      if ((rc = upload(Gs.data(), Gs.size() * sizeof(double2), &dg)) != kOk) return rc;
      fu.Gs = static_cast<double2 *>(dg);
    }
    fu.span = p.n + dmax;
    fu.NG = NG;
    fu.KC = split ? 1 : threads / NG; // (vector variant's thread map: unused by the sub-blocked form, whose NG may exceed 256)
    // blocks per launch: as many as a seam ring of at most 1280 MB allows (0.4 % of the card; round 2's 320 MB cut a 963 379-frame
    // push of 256 stereo streams into three launches, each with its own prep / seam kernels and gaps; two launches worth of slots: seam(k) still
    // reads its slots while fused(k+1) fills the next ones)
    fu.blk_cap = 64;
    while (fu.blk_cap < kFusedMaxBlocks && double(C_ + 1) * double(4 * fu.blk_cap) * 512 <= kn.seam_ring_mb * 1048576.0) fu.blk_cap *= 2;
    while (fu.blk_cap < 8 * split_nsub) fu.blk_cap *= 2; // (sub-blocked: the table counts sub-blocks; at least 6 whole blocks per launch)
    fu.slots = 2 * fu.blk_cap;
    ALLOC_TRY(&fu.blk_dev, size_t(2 * fu.blk_cap) * sizeof(FusedBlock)); // two halves: launch k uses half k & 1 (see advance)
    const size_t bytes = size_t(C_ + 1) * fu.slots * 2 * 32 * sizeof(double);
    ALLOC_TRY(&fu.seam, bytes);
    HIP_TRY(hipMemset(fu.seam, 0, bytes));
    const int V = f.N - (f.num_taps - 1);
    // periods a block can touch: ceil(outputs per block / L) + 1; chunk length fixed from it
    const int Kmax = int(((long long)V * p.L / pstep + p.L - 1) / p.L) + 2;
    fu.kper = (Kmax + fu.KC - 1) / fu.KC;
    if (!split) { // coefficient tiles, one per thread of the fused kernel: tile[mm][g] = row(phase of residue G*m+g)[mm - d_g]
      std::vector<double> tiles(size_t(32) * G * threads, 0.0);
      for (int t = 0; t < threads; ++t) {
        const int m = t % NG, kc = t / NG;
        if (kc >= fu.KC) continue;
        const int q0 = (at0 + G * m * pstep) / p.L;
        for (int g = 0; g < G; ++g) {
          const int r = G * m + g;
          if (r >= p.L) continue;
          const int ar = at0 + r * pstep, q = ar / p.L, ph = ar - q * p.L, dsh = q - q0;
          for (int j = 0; j < p.n; ++j)
            if (j + dsh < 32) tiles[(size_t(j + dsh) * G + g) * threads + t] = plan_.poly_table[size_t(ph) * p.n + j];
        }
      }
      void *d = nullptr;
      if ((rc = upload(tiles.data(), tiles.size() * sizeof(double), &d)) != kOk) return rc;
      fu.cft = static_cast<double *>(d);
    }
    { // matrix-pipe variant (fused.hip): A operands of v_mfma_f64_4x4x4, lane = 16k + 4b + i holds the
      // coefficient of residue 16g + 4b + i at tap 4s + k of its 4-residue block's common window
      int d4 = 0;
      for (int rb = 0; rb < p.L; rb += 4) {
        const long long a0 = at0 + (long long)rb * pstep, a1 = at0 + (long long)std::min(rb + 3, p.L - 1) * pstep;
        d4 = std::max(d4, int(a1 / p.L - a0 / p.L));
      }
      const int KS = std::max(7, (p.n + d4 + 3) / 4), NGRP = (p.L + 15) / 16;
      // window starts of the 4-residue blocks span [qb_min, qb_max]; the two-round sample image of the kernel
      // needs every period to fit one of the rounds (fused.hip, kSA / kSB0)
      int qb_min = at0 / p.L, qb_max = qb_min;
      for (int rb = 0; rb < p.L; rb += 4) qb_max = std::max(qb_max, int((at0 + (long long)rb * pstep) / p.L));
      // ... and at most 32 periods per block (4 column steps per item), enough phases to fill 16-row tiles
      // (The four periods of a tile -- lanes j = 0..3 of a B read, `step` samples apart -- fall on the same LDS banks when
      // step is a multiple of 8 samples: 96k->44.1k (step 320), 48k->44.1k (160).  Such chains used to be kept on the vector
      // variant (3.90 against 3.74 ms at the time); with the lean kernel's later gains the matrix-pipe variant wins despite
      // its 4-way conflicts: 96k->44.1k 3.26 against 3.74 ms, 48k->44.1k +25 %.  RSMP_SPREAD_VECTOR=1 restores the old choice.)
      bool lanes_spread = true;
      if (kn.spread_vector)
        for (int j1 = 0; j1 < 4; ++j1)
          for (int j2 = j1 + 1; j2 < 4; ++j2)
            if (((j2 - j1) * pstep) % 16 == 0) lanes_spread = false;
      const bool rounds_ok = (qb_max - qb_min) + 4 * KS + 4 <= (kFusedSA - kFusedSB0) * 256 + 32 && Kmax <= 32 && p.L >= 64 && lanes_spread;
      if (split || (!kn.no_mfma && fused_mfma_supported(log2n, log2p, KS) && rounds_ok)) { // (split: one image, no rounds to fit)
        std::vector<double> am(size_t(NGRP) * KS * 64, 0.0);
        for (int g = 0; g < NGRP; ++g)
          for (int s = 0; s < KS; ++s)
            for (int lane = 0; lane < 64; ++lane) {
              const int k = lane >> 4, bq = (lane >> 2) & 3, ii = lane & 3;
              const int rb = 16 * g + 4 * bq, r = rb + ii;
              if (r >= p.L) continue;
              const int qb = (at0 + rb * pstep) / p.L;
              const int ar = at0 + r * pstep, q = ar / p.L, ph = ar - q * p.L;
              const int j = 4 * s + k - (q - qb);
              if (j >= 0 && j < p.n) am[(size_t(g) * KS + s) * 64 + lane] = plan_.poly_table[size_t(ph) * p.n + j];
            }
        void *dm = nullptr;
        if ((rc = upload(am.data(), am.size() * sizeof(double), &dm)) != kOk) return rc;
        fu.cfm = static_cast<double *>(dm);
        { // the same tiles, two k-steps per 16-byte element (half the load instructions in the lean kernel)
          const int KSP = (KS + 1) / 2;
          std::vector<double2> am2(size_t(NGRP) * KSP * 64, make_double2(0.0, 0.0));
          for (int g = 0; g < NGRP; ++g)
            for (int s = 0; s < KS; ++s)
              for (int lane = 0; lane < 64; ++lane) {
                double2 &d = am2[(size_t(g) * KSP + s / 2) * 64 + lane];
                (s & 1 ? d.y : d.x) = am[(size_t(g) * KS + s) * 64 + lane];
              }
          if (KS & 1) // the spare half of the last element carries the lane's window start (as qtab: block bq = (lane >> 2) & 3)
            for (int g = 0; g < NGRP; ++g)
              for (int lane = 0; lane < 64; ++lane) {
                int rb = 16 * g + 4 * ((lane >> 2) & 3);
                if (rb >= p.L) rb = 0;
                am2[(size_t(g) * KSP + KSP - 1) * 64 + lane].y = double((at0 + rb * pstep) / p.L);
              }
          void *d2 = nullptr;
          if ((rc = upload(am2.data(), am2.size() * sizeof(double2), &d2)) != kOk) return rc;
          fu.cfm2 = static_cast<double2 *>(d2);
        }
        fu.NGRP = NGRP;
        fu.KS = KS;
        fu.qb_max = qb_max;
        fu.qb_min = qb_min;
        std::vector<int> qt(size_t(NGRP) * 4);
        for (int g = 0; g < NGRP; ++g)
          for (int bq = 0; bq < 4; ++bq) {
            int rb = 16 * g + 4 * bq;
            if (rb >= p.L) rb = 0; // idle block: all-zero coefficients, any in-range window will do
            qt[size_t(g) * 4 + bq] = (at0 + rb * pstep) / p.L;
          }
        void *dq = nullptr;
        if ((rc = upload(qt.data(), qt.size() * sizeof(int), &dq)) != kOk) return rc;
        fu.qtab = static_cast<int *>(dq);
      }
    }
    const size_t per_launch = size_t(fu.blk_cap / std::max(1, fu.nsub) - 2) * size_t((V - d.remL0 + d.L - 1) / d.L);
    // frames of chain input per launch: divide by the rate of everything ahead of the dft stage
    double ahead = 1;
    for (int k = 0; k < i; ++k) ahead *= plan_.stages[k].kind == StageKind::Half ? 0.5 : plan_.stages[k].out_in_ratio;
    const size_t cap = size_t(double(per_launch) / std::max(ahead, 1e-9));
    fused_slab_cap = fused_slab_cap ? std::min(fused_slab_cap, cap) : cap;
  }
  // rational polyphase stages that are not fused run as a matrix-pipe stage of their own (polymf.hip) when
  // they have enough phases to fill 16-row tiles
  polymf_.assign(ns, PolyMf());
  for (int i = 0; i < ns; ++i) {
    const StageSpec &p = plan_.stages[i];
    if (p.kind != StageKind::Poly || p.order != 0 || (i > 0 && fuse_[i - 1].on) || p.L < 64) continue;
    if (kn.no_mfma || kn.no_polymf) continue;
    const int pstep = int(p.step64 >> 32), at0 = int(p.at0 >> 32);
    int d4 = 0;
    for (int rb = 0; rb < p.L; rb += 4) {
      const long long a0 = at0 + (long long)rb * pstep, a1 = at0 + (long long)std::min(rb + 3, p.L - 1) * pstep;
      d4 = std::max(d4, int(a1 / p.L - a0 / p.L));
    }
    const int KS = std::max(7, (p.n + d4 + 3) / 4), NGRP = (p.L + 15) / 16;
    if (!polymf_supported(KS)) continue;
    // tile length: at most 2048 stage-input samples and at most 30 output periods per tile
    int Vt = 2048;
    while (Vt > 256 && ((long long)Vt * p.L / pstep + p.L - 1) / p.L + 2 > 30) Vt -= 256;
    if (((long long)Vt * p.L / pstep + p.L - 1) / p.L + 2 > 30) continue;
    std::vector<double> am(size_t(NGRP) * KS * 64, 0.0);
    for (int g = 0; g < NGRP; ++g)
      for (int s = 0; s < KS; ++s)
        for (int lane = 0; lane < 64; ++lane) {
          const int k = lane >> 4, bq = (lane >> 2) & 3, ii = lane & 3;
          const int rb = 16 * g + 4 * bq, r = rb + ii;
          if (r >= p.L) continue;
          const int qb = (at0 + rb * pstep) / p.L;
          const int ar = at0 + r * pstep, q = ar / p.L, ph = ar - q * p.L;
          const int j = 4 * s + k - (q - qb);
          if (j >= 0 && j < p.n) am[(size_t(g) * KS + s) * 64 + lane] = plan_.poly_table[size_t(ph) * p.n + j];
        }
    PolyMf &pm = polymf_[i];
    void *dm = nullptr;
    if ((rc = upload(am.data(), am.size() * sizeof(double), &dm)) != kOk) return rc;
    pm.cfm = static_cast<double *>(dm);
    {
      std::vector<int> qt(size_t(NGRP) * 4);
      for (int g = 0; g < NGRP; ++g)
        for (int bq = 0; bq < 4; ++bq) {
          int rb = 16 * g + 4 * bq;
          if (rb >= p.L) rb = 0;
          qt[size_t(g) * 4 + bq] = (at0 + rb * pstep) / p.L;
        }
      void *dq = nullptr;
      if ((rc = upload(qt.data(), qt.size() * sizeof(int), &dq)) != kOk) return rc;
      pm.qtab = static_cast<int *>(dq);
    }
    pm.KS = KS;
    pm.NGRP = NGRP;
    pm.Vt = Vt;
    pm.blk_cap = 4096;
    ALLOC_TRY(&pm.blk, size_t(pm.blk_cap) * sizeof(FusedBlock));
  }
  // the fifo between two fused stages carries no bulk data
  bytes_per_in_frame = 0;
  rate = 1;
  for (int i = 0; i < ns; ++i) {
    const StageSpec &sp = plan_.stages[i];
    if (sp.kind == StageKind::Dft) rate *= sp.step > 0 ? double(sp.L) / sp.step : double(sp.L) / double(1 << -sp.step);
    else if (sp.kind == StageKind::Poly) rate *= sp.out_in_ratio;
    else rate *= 0.5;
    if (i + 1 < ns && !fuse_[i].on) bytes_per_in_frame += rate * 8.0 * C_;
  }
  // Bound the fp64 fifos between stages: a push is cut into time slabs, each slab runs through every stage
  // before the next one starts.  Measured on the 3-stage 44.1k->192k chain (32 streams x 8 ch): 96 MB slabs 15.0,
  // 192 MB 15.5, 400 MB 16.6, 1600 MB 17.3 Gsamples/s -- launch size matters more than Infinity-Cache residency.
  const double budget = kn.slab_mb * 1024 * 1024;
  slab_frames_ = bytes_per_in_frame > 0 ? size_t(budget / bytes_per_in_frame) : plan_.isamp_max;
  slab_frames_ = std::max<size_t>(slab_frames_, 8192);
  slab_frames_ = std::min<size_t>(slab_frames_, std::max<size_t>(plan_.isamp_max, 1));
  if (fused_slab_cap) slab_frames_ = std::max<size_t>(1024, std::min(slab_frames_, fused_slab_cap));
  for (int i = 0; i <= ns; ++i)
    if ((rc = ensure_ring(i, std::max<long long>(book_.wr[i], 1))) != kOk) return rc;
  HIP_TRY(hipStreamSynchronize(stream_));
  return kOk;
}

void Engine::set_profiling(bool on)
{
  profiling_ = on;
  for (ProfRec &r : prof_) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  prof_.clear();
}

int Engine::prof_begin(bool hot, const char *name)
{
  if (!profiling_) return -1;
  ProfRec r;
  r.hot = hot;
  r.name = name;
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return -1;
  (void)hipEventRecord(r.e0, stream_);
  prof_.push_back(r);
  return int(prof_.size()) - 1;
}

void Engine::prof_end(int idx)
{
  if (idx >= 0) (void)hipEventRecord(prof_[idx].e1, stream_);
}

// Sums the recorded launch durations (after synchronising the stream) and clears the records.
// "hot" = the kernel that carries the bulk of the work of its chain (fused dft->poly, or the dft stage).
int Engine::read_profile(double *hot_ms, long long *hot_launches, double *other_ms, long long *other_launches)
{
  HIP_TRY(hipStreamSynchronize(stream_));
  double h = 0, o = 0;
  long long hn = 0, on = 0;
  for (ProfRec &r : prof_) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      if (r.hot) { h += ms; ++hn; } else { o += ms; ++on; }
    }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  prof_.clear();
  if (hot_ms) *hot_ms = h;
  if (hot_launches) *hot_launches = hn;
  if (other_ms) *other_ms = o;
  if (other_launches) *other_launches = on;
  return kOk;
}

int Engine::read_profile_json(std::string &out)
{
  HIP_TRY(hipStreamSynchronize(stream_));
  struct Agg { const char *name; bool hot; long long n; double ms; };
  std::vector<Agg> agg;
  for (ProfRec &r : prof_) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      Agg *a = nullptr;
      for (Agg &x : agg)
        if (std::strcmp(x.name, r.name) == 0 && x.hot == r.hot) a = &x;
      if (!a) {
        agg.push_back(Agg{r.name, r.hot, 0, 0.0});
        a = &agg.back();
      }
      a->n += 1;
      a->ms += ms;
    }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  prof_.clear();
  out = "[";
  char buf[64];
  for (size_t i = 0; i < agg.size(); ++i) {
    out += i ? ", {" : "{";
    out += "\"kernel\": \"";
    out += agg[i].name;
    snprintf(buf, sizeof buf, "\", \"hot\": %d, \"launches\": %lld, \"ms\": %.6f}", agg[i].hot ? 1 : 0, agg[i].n, agg[i].ms);
    out += buf;
  }
  out += "]";
  return kOk;
}

// make the main stream wait for the seam kernels still running on the side stream
int Engine::join_side()
{
  if (!side_pending_) return kOk;
  HIP_TRY(hipStreamWaitEvent(stream_, ev_seam_[(seam_launches_ - 1) & 1], 0)); // the side stream is in order
  side_pending_ = false;
  return kOk;
}

int Engine::pinned_reserve(Pinned &b, size_t floats)
{
  if (!b.done) HIP_TRY(hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
  if (b.floats >= floats) return kOk;
  if (b.p) {
    // growing: queued work may still read or write the old block (a mirror being spilled into the device ring, a copy out of
    // it); nothing that is queued may outlive the block
    HIP_TRY(hipStreamSynchronize(stream_));
    (void)hipHostFree(b.p);
  }
  b.p = nullptr;
  b.floats = 0;
  size_t want = 4096;
  while (want < floats) want <<= 1;
  void *p = nullptr;
  if (hipHostMalloc(&p, want * sizeof(float), hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return kNoMem;
  }
  b.p = static_cast<float *>(p);
  b.floats = want;
  return kOk;
}

void Engine::pinned_free(Pinned &b)
{
  if (b.p) (void)hipHostFree(b.p);
  if (b.done) (void)hipEventDestroy(b.done);
  b = Pinned();
}

void Engine::free_garbage()
{
  for (void *p : garbage_) (void)hipFree(p);
  garbage_.clear();
}

Engine::~Engine()
{
  DeviceScope on(device_);
  (void)hipStreamSynchronize(stream_); // nullptr is the default stream (RRX_set_stream(h, NULL)): it too may hold queued work on our buffers
  if (own_ && own_ != stream_) (void)hipStreamSynchronize(own_);
  free_garbage();
  for (Ring &r : rings_) if (r.buf) (void)hipFree(r.buf);
  for (double2 *&g : d_G_) if (g) (void)hipFree(g);
  for (double2 *&g : d_Gr_) if (g) (void)hipFree(g);
  if (d_poly_) (void)hipFree(d_poly_);
  for (double2 *&t : d_tw_) if (t) (void)hipFree(t);
  for (double2 *&t : d_tw8_) if (t) (void)hipFree(t);
  set_profiling(false);
  if (d_stage_) (void)hipFree(d_stage_);
  pinned_free(pin_in_[0]);
  pinned_free(pin_in_[1]);
  pinned_free(pin_out_);
  pinned_free(pin_mir_);
  if (stamps_) {
    unsigned long long h[16] = {};
    if (hipMemcpy(h, stamps_, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[7] && h[9])
      fprintf(stderr, "RSMP_FINE per workgroup (wave 0): setup %.0f  drain %.0f  flush %.0f  Aload+addr+fill %.0f  first-LDS %.0f  steps %.0f  bookkeeping %.0f  loop %.0f\n",
              double(h[8]) / h[7], double(h[9]) / h[7], double(h[10]) / h[7], double(h[11]) / h[7], double(h[12]) / h[7], double(h[13]) / h[7],
              double(h[14]) / h[7], double(h[15]) / h[7]);
    if (h[7])
      fprintf(stderr, "RSMP_STAMPS workgroups %llu  avg cycles: load %.0f  fwd %.0f  mul %.0f  inv %.0f  cf+smp %.0f  polyA %.0f  poly(B) %.0f  total %.0f\n",
              h[7], double(h[0]) / h[7], double(h[1]) / h[7], double(h[2]) / h[7], double(h[3]) / h[7], double(h[4]) / h[7],
              double(h[6]) / h[7], double(h[5]) / h[7], double(h[0] + h[1] + h[2] + h[3] + h[4] + h[5] + h[6]) / h[7]);
    (void)hipFree(stamps_);
  }
  for (Fuse &f : fuse_) {
    if (f.seam) (void)hipFree(f.seam);
    if (f.cft) (void)hipFree(f.cft);
    if (f.cfm) (void)hipFree(f.cfm);
    if (f.qtab) (void)hipFree(f.qtab);
    if (f.cfm2) (void)hipFree(f.cfm2);
    if (f.Gs) (void)hipFree(f.Gs);
    if (f.blk_dev) (void)hipFree(f.blk_dev);
  }
  for (BigDft &b : big_) {
    if (b.twN) (void)hipFree(b.twN);
    if (b.w1) (void)hipFree(b.w1);
    if (b.w2) (void)hipFree(b.w2);
  }
  for (PolyMf &m : polymf_) {
    if (m.cfm) (void)hipFree(m.cfm);
    if (m.qtab) (void)hipFree(m.qtab);
    if (m.blk) (void)hipFree(m.blk);
  }
  if (side_) { (void)hipStreamSynchronize(side_); (void)hipStreamDestroy(side_); }
  if (ev_fused_) (void)hipEventDestroy(ev_fused_);
  for (hipEvent_t &e : ev_seam_) if (e) (void)hipEventDestroy(e);
  if (ev_switch_) (void)hipEventDestroy(ev_switch_);
  if (own_) (void)hipStreamDestroy(own_); // never the caller's stream
}

int Engine::set_stream(hipStream_t s, bool own)
{
  hipStream_t next = own ? own_ : s; // s == nullptr is the device's default (legacy) stream, as in every HIP call
  if (next == stream_) return kOk;
  { int rcj = join_side(); if (rcj) return rcj; } // seam kernels still on the side stream belong to the old stream's work
  HIP_TRY(hipEventRecord(ev_switch_, stream_));
  HIP_TRY(hipStreamWaitEvent(next, ev_switch_, 0));
  stream_ = next;
  return kOk;
}

int Engine::sync()
{
  HIP_TRY(hipStreamSynchronize(stream_));
  free_garbage();
  return kOk;
}

F32View Engine::f32_view(int f, const ExtIn *ein, const ExtOut *eout) const
{
  const Ring &r = rings_[f];
  F32View v;
  v.ring = static_cast<float *>(r.buf);
  v.ring_mask = r.cap - 1;
  v.ring_stream_stride = r.cap * nch_;
  v.ext = nullptr;
  v.ext_begin = v.ext_end = 0;
  v.ext_stream_stride = 0;
  v.nch = nch_;
  if (ein && ein->ptr) {
    v.ext = const_cast<float *>(ein->ptr);
    v.ext_begin = ein->begin;
    v.ext_end = ein->end;
    v.ext_stream_stride = ein->stride_floats;
  } else if (eout && eout->ptr) {
    v.ext = eout->ptr;
    v.ext_begin = eout->begin;
    v.ext_end = eout->end;
    v.ext_stream_stride = eout->stride_floats;
  }
  return v;
}

F64View Engine::f64_view(int f) const
{
  const Ring &r = rings_[f];
  F64View v;
  v.ring = static_cast<double *>(r.buf);
  v.mask = r.cap - 1;
  v.chan_stride = r.cap;
  return v;
}

// Make ring f able to hold `live_needed` items/frames counted from book_.rd[f]; existing live data
// [rd, wr) is carried over to the new ring at the same absolute indices.
int Engine::ensure_ring(int f, long long live_needed)
{
  Ring &r = rings_[f];
  if (r.buf && r.cap >= live_needed) return kOk;
  { int rcj = join_side(); if (rcj) return rcj; } // seam kernels on the side stream may still write the old ring
  const long long cap = next_pow2(std::max<long long>({live_needed, r.cap * 2, 4096}));
  const size_t bytes = r.f32 ? size_t(cap) * nch_ * S_ * sizeof(float) : size_t(cap) * C_ * sizeof(double);
  void *nb = nullptr;
  ALLOC_TRY(&nb, bytes);
  HIP_TRY(hipMemsetAsync(nb, 0, bytes, stream_));
  if (r.buf) {
    Ring old = r;
    r.buf = nb;
    r.cap = cap;
    const long long a0 = book_.rd[f], a1 = book_.wr[f];
    F32View sf = {}, df = {};
    F64View sd = {}, dd = {};
    if (r.f32) {
      df = f32_view(f, nullptr, nullptr);
      sf = df;
      sf.ring = static_cast<float *>(old.buf);
      sf.ring_mask = old.cap - 1;
      sf.ring_stream_stride = old.cap * nch_;
    } else {
      dd = f64_view(f);
      sd = dd;
      sd.ring = static_cast<double *>(old.buf);
      sd.mask = old.cap - 1;
      sd.chan_stride = old.cap;
    }
    HIP_TRY(launch_copy(r.f32, sf, sd, df, dd, a0, a1, C_, stream_));
    garbage_.push_back(old.buf);
  } else {
    r.buf = nb;
    r.cap = cap;
  }
  return kOk;
}

void Engine::note_input(Book &b, size_t n) const
{ // rate_base.h:436-441
  b.samples_in += n;
  while (b.samples_in > plan_.cfg.in_rate && b.samples_out > plan_.cfg.out_rate) {
    b.samples_in -= plan_.cfg.in_rate;
    b.samples_out -= plan_.cfg.out_rate;
  }
}

// What the per-stage functions of one pass share (see Engine::Pass): stage i reads fifo i and writes fifo i + 1.
#define RSMP_STAGE_LOCALS(ps, i)                                                                                   \
  Book &b = (ps).b;                                                                                                \
  const bool launch = (ps).launch;                                                                                 \
  const ExtIn &ein = (ps).ein;                                                                                     \
  const ExtOut &eout = (ps).eout;                                                                                  \
  const int ns = int(plan_.stages.size());                                                                         \
  const StageSpec &sp = plan_.stages[i];                                                                           \
  Book::St &st = b.st[i];                                                                                          \
  long long &rd = b.rd[i];                                                                                         \
  long long &wro = b.wr[(i) + 1];                                                                                  \
  const long long occ = b.wr[i] - rd;                                                                              \
  const bool src_f32 = (i) == 0, dst_f32 = (i) + 1 == ns;                                                          \
  const long long rd_before = rd, wro_before = wro;                                                                \
  const long long out_offset = (i) + 1 < ns ? plan_.stages[(i) + 1].preload : -b.trimmed;                          \
  const F32View nof = {};                                                                                          \
  const F64View nod = {};                                                                                          \
  Pending &pend = (ps).pend;                                                                                       \
  /* what the destination ring must be able to hold once this stage has run */                                    \
  auto dst_need = [&](long long wr_after) {                                                                        \
    if (dst_f32 && eout.ptr) return std::max<long long>(0, wr_after - std::max(eout.end, b.rd[(i) + 1]));         \
    return wr_after - b.rd[(i) + 1];                                                                               \
  };                                                                                                               \
  (void)rd_before; (void)wro_before; (void)out_offset; (void)occ; (void)src_f32; (void)dst_f32; (void)launch;      \
  (void)ein; (void)nof; (void)nod; (void)pend; (void)st; (void)dst_need

// One pass of rate_process (rate_base.h:425-432) after `n_new` frames were appended to fifo 0: every stage runs once over
// what is available.  With launch == false only the counters move (used to size a drain or a host mirror).
int Engine::advance(Book &b, size_t n_new, bool launch, const ExtIn &ein, const ExtOut &eout, bool more_slabs)
{
  note_input(b, n_new);
  b.wr[0] += (long long)n_new;
  Pass ps{b, launch, ein, eout, more_slabs, Pending()};
  for (int i = 0; i < int(plan_.stages.size()); ++i) {
    const StageKind kind = plan_.stages[i].kind;
    const int rc = kind == StageKind::Dft ? advance_dft(ps, i) : kind == StageKind::Poly ? advance_poly(ps, i) : advance_half(ps, i);
    if (rc) return rc;
  }
  return kOk;
}

// dft_stage_fn (dft_filter.h:60-190): the reference's block loop on the counters, then one launch for all blocks of the pass
int Engine::advance_dft(Pass &ps, int i)
{
  RSMP_STAGE_LOCALS(ps, i);
  DftArgs &pend_args = pend.args;
  int &pend_log2n = pend.log2n, &pend_log2p = pend.log2p;
  const DftFilter &f = plan_.dft[sp.filt];
  const int N = f.N, ov = f.num_taps - 1, V = N - ov, L = sp.L;
  const bool stuffing = L != 1 && !pow2_ge2(L);
  const int kept = sp.step < 0 ? N - ((((1 << -sp.step) - 1) * N + ov) >> -sp.step) : V; // dft_filter.h:187
  long long num_in = std::max<long long>(0, occ);
  const long long B0 = st.B;
  int nblocks = 0;
  while (st.remL + (long long)L * num_in >= N) { // dft_filter.h:78-84
    const int span = V - st.remL + L - 1;
    const int take = span / L, rem = span % L;
    rd += take;
    num_in -= take;
    if (stuffing) st.remL = L - 1 - rem;
    if (sp.step > 1) {
      const int j = (V - st.remM + sp.step - 1) / sp.step; // dft_filter.h:150-152
      st.remM = st.remM + j * sp.step - V;
      wro += j;
    } else
      wro += kept;
    ++nblocks;
    ++st.B;
  }
  const bool fused = fuse_[i].on;
  if (launch && nblocks) {
    if (!fused) {
      int rc = ensure_ring(i + 1, dst_need(wro));
      if (rc) return rc;
    }
    const int Ng = N, Vg = V;
    const int log2n = ilog2(Ng);
    DftArgs a;
    a.G = d_G_[sp.filt];
    a.Gr = d_Gr_[sp.filt];
    const int log2p = fdomain_up(L) ? log2n - ilog2(L) : log2n;
    const int log2nd = sp.step < 0 ? log2n + sp.step : log2n;
    const bool big = big_[i].on;
    const bool sub = fused && fuse_[i].nsub > 0; // sub-blocked fused launch: 4096-point tables, set where it is launched
    a.tw_fwd = sub ? nullptr : twiddles(big ? log2p - 4 : log2p);
    a.tw_inv = sub ? nullptr : twiddles(big ? log2nd - 4 : log2nd);
    a.tw_fwd8 = (!sub && !big && log2p >= 6 && log2p <= 13) ? twiddles8(log2p) : nullptr;
    if (!sub && !big && log2p >= 6 && log2p <= 13 && !a.tw_fwd8) return kNoMem;
    a.tw_inv8 = (!sub && !big && log2nd >= 6 && log2nd <= 12) ? twiddles8(log2nd) : nullptr;
    if (!sub && !big && log2nd >= 6 && log2nd <= 12 && !a.tw_inv8) return kNoMem;
    a.B0 = B0;
    a.out_offset = out_offset;
    a.nblocks = nblocks;
    a.C = C_;
    a.L = L;
    a.c0 = pow2_ge2(L) ? 0 : sp.remL0; // the frequency-domain branch places input 0 of a block at slot 0 whatever remL is
    a.V = Vg;
    a.Vout = sp.step < 0 ? Ng - ((((1 << -sp.step) - 1) * Ng + ov) >> -sp.step) : Vg;
    a.q = (Vg - sp.remL0 + L - 1) / L;
    a.M = sp.step > 1 ? sp.step : 1;
    a.hp = 0; // set by the launchers (frame_pairs)
    a.nchs = pair_nchs();
    a.in_limit = 0x7fffffffffffffffLL;
    a.clip_lo = -0x7fffffffffffffffLL;
    a.clip_hi = 0x7fffffffffffffffLL;
    a.nsub = a.Vs = a.Pref = a.two = 0;
    a.Bref0 = 0;
    if (big) {
      const BigDft &bg = big_[i];
      BigDftArgs ba;
      ba.d = a;
      ba.twN = bg.twN;
      ba.w1 = bg.w1;
      ba.w2 = bg.w2;
      ba.log2n = log2n;
      ba.log2mp = log2p - 4;
      ba.log2md = log2nd - 4;
      ba.fdomain_in = (log2p < log2n || L == 1) ? 1 : 0;
      ba.item0 = 0;
      const int pi = prof_begin(true, "rsmp::big_cols_fwd_kernel + big_rows_kernel + big_cols_inv_kernel");
      HIP_TRY(launch_dft_big(src_f32, dst_f32, src_f32 ? f32_view(i, &ein, nullptr) : nof, src_f32 ? nod : f64_view(i),
                             dst_f32 ? f32_view(i + 1, nullptr, &eout) : nof, dst_f32 ? nod : f64_view(i + 1), ba, bg.ws_items,
                             stream_));
      prof_end(pi);
    } else
    if (fused) { // launched together with the polyphase stage below
      pend.B0 = B0;
      pend.nblocks = nblocks;
      pend_args = a;
      pend_log2n = log2n;
      pend_log2p = log2p;
    } else if (a.Gr && sp.step == 1 && fdomain_up(L) && dftx_supported(log2n, log2p, log2nd)) {
      const int pi = prof_begin(true);
      const char *kn = nullptr;
      HIP_TRY(launch_dftx(log2n, src_f32, dst_f32, src_f32 ? f32_view(i, &ein, nullptr) : nof, src_f32 ? nod : f64_view(i),
                          dst_f32 ? f32_view(i + 1, nullptr, &eout) : nof, dst_f32 ? nod : f64_view(i + 1), a, stream_, &kn));
      prof_name(pi, kn);
      prof_end(pi);
    } else {
    const int pi = prof_begin(true);
    const char *kn = nullptr;
    HIP_TRY(launch_dft(log2n, log2p, log2nd, src_f32, dst_f32, src_f32 ? f32_view(i, &ein, nullptr) : nof,
                       src_f32 ? nod : f64_view(i), dst_f32 ? f32_view(i + 1, nullptr, &eout) : nof,
                       dst_f32 ? nod : f64_view(i + 1), a, stream_, &kn));
    prof_name(pi, kn);
    prof_end(pi);
    }
  }
  return kOk;
}

// vpoly0..3 (rate_filters_generic.h:272-305, 376-504): output count from the clock, then the launch of whichever kernel serves
// this stage (fused with the dft stage in front of it, matrix-pipe stage of its own, or the generic polyphase kernels)
int Engine::advance_poly(Pass &ps, int i)
{
  RSMP_STAGE_LOCALS(ps, i);
  const long long num_in = std::max<long long>(0, occ - sp.pre_post); // rate_base.h:130
  long long count = 0, at_end = st.at;
  const long long step = sp.order == 0 ? (sp.step64 >> 32) : sp.step64;
  const long long lim = sp.order == 0 ? num_in * sp.L : (num_in << 32);
  if (st.at < lim) count = (lim - st.at + step - 1) / step; // rate_filters_generic.h:281 / :477
  at_end = st.at + count * step;
  const bool fused = i > 0 && fuse_[i - 1].on;
  if (launch && fused) {
    if ((count != 0) != (pend.nblocks != 0)) return kInternal;
    if (pend.nblocks) {
      const int rc = launch_fused_pair(ps, i, count, step);
      if (rc) return rc;
    }
  } else if (launch && count && polymf_[i].cfm) {
    const int rc = launch_polymf_stage(ps, i, count, step);
    if (rc) return rc;
  } else if (launch && count) {
    const int rc = launch_poly_stage(ps, i, count, step);
    if (rc) return rc;
  }
  if (sp.order == 0) {
    rd += at_end / sp.L; // rate_filters_generic.h:302-304
    st.at = at_end % sp.L;
  } else {
    rd += at_end >> 32; // rate_filters_generic.h:499-500
    st.at = at_end & 0xffffffffLL;
  }
  wro += count;
  return kOk;
}

// the dft stage i - 1 and this rational polyphase stage as ONE kernel per block range (fused_fast.hip / fused.hip) + seam_kernel
int Engine::launch_fused_pair(Pass &ps, int i, long long count, long long step)
{
  RSMP_STAGE_LOCALS(ps, i);
  const bool more_slabs = ps.more_slabs;
  DftArgs &pend_args = pend.args;
  const int pend_log2n = pend.log2n, pend_log2p = pend.log2p;
  {
    int rc = ensure_ring(i + 1, dst_need(wro + count));
    if (rc) return rc;
    const Fuse &fu = fuse_[i - 1];
    FusedArgs fa;
    fa.d = pend_args;
    fa.tab = d_poly_;
    fa.seam = fu.seam;
    fa.cft = fu.cft;
    fa.at0 = sp.at0 >> 32;
    fa.b_offset = sp.preload;
    fa.out_offset2 = out_offset;
    fa.seam_mask = fu.slots - 1;
    fa.n = sp.n;
    fa.polyL = sp.L;
    fa.step = int(step);
    fa.span = fu.span;
    fa.NG = fu.NG;
    fa.KC = fu.KC;
    fa.kper = fu.kper;
    fa.cfm = fu.cfm;
    fa.qtab = fu.qtab;
    fa.cfm2 = fu.cfm2;
    fa.NGRP = fu.NGRP;
    fa.KS = fu.KS;
    fa.dbg = dbg_;
    fa.stamps = stamps_;
    const bool split = fu.nsub > 0;
    const int ntab = split ? pend.nblocks * fu.nsub : pend.nblocks; // table entries = workgroups per pair: blocks, or sub-blocks
    if (ntab > fu.blk_cap) return kInternal;
    FusedPrepArgs pa; // output bookkeeping of each block (closed forms in kernels.hpp), evaluated on the device
    pa.b_offset = fa.b_offset;
    pa.B0 = pend.B0;
    pa.at0 = fa.at0;
    pa.V = fa.d.V;
    pa.polyL = sp.L;
    pa.step = int(step);
    pa.n = sp.n;
    pa.nblocks = ntab;
    pa.nsub = fu.nsub;
    pa.Vs = fu.Vs;
    const bool split_two = split && fu.nsub == 1 && fu.Vs > kSplitVsMax; // whole 8192-point blocks, two rounds (split_geometry)
    pa.two_round = fu.cfm != nullptr && (!split || split_two);
    pa.ra_end = split_two ? kSplitRaEnd : 0;
    pa.rb_start = split_two ? kSplitRbStart : 0;
    pa.KS = fu.KS;
    pa.qb_max = fu.qb_max;
    pa.qb_min = fu.qb_min;
    pa.clip_lo = 0;
    pa.clip_hi = 0x7fffffffffffffffLL;
    if (!split)
      for (int k : {0, pend.nblocks - 1}) // same closed forms on the host: bounds the kernels rely on
        if (fused_block_info(pa, k).K > (fu.cfm ? 32 : fu.KC * fu.kper)) return kInternal;
    // side stream for the seam kernel only when nothing downstream in this pass reads the seam outputs (poly is the last stage)
    // ... and only when another slab of this push follows: seam(k) then runs beside fused(k+1).  Behind the LAST fused
    // launch of a push the side stream has nothing to overlap with but the small carry copy, and the two cross-queue
    // hand-overs (event -> side stream -> join) cost more than they hide: 2.455 against 2.505 ms per step measured.
    const bool seam_on_side = more_slabs && !(profiling_ || !dst_f32 || no_side_);
    // seam(k-2), possibly still pending on the side stream, reads the seam-ring slots AND the half of the block table
    // that this launch is about to overwrite: both the table fill and the fused launch wait for it.  (The table has
    // two halves, launch k uses half k & 1: seam(k-1) may still be reading the other one.  With ONE table, a push cut
    // into three launches lost the seam outputs of its first blocks: tests/test_gpu_round3.py::test_cfg0_bench_shape_*.)
    if (seam_launches_ >= 2) HIP_TRY(hipStreamWaitEvent(stream_, ev_seam_[seam_launches_ & 1], 0));
    FusedBlock *const blk_half = fu.blk_dev + size_t(seam_launches_ & 1) * fu.blk_cap;
    { const int pp = prof_begin(false, "rsmp::fused_prep_kernel"); HIP_TRY(launch_fused_prep(pa, blk_half, stream_)); prof_end(pp); }
    fa.blk = blk_half;
    // the fused launch emits exactly the outputs [wro, wro + count): windows ending before wr of fifo i
    const long long endnum = (b.wr[i] - sp.n + 1) * sp.L - fa.at0;
    if (wro - out_offset + count != (endnum <= 0 ? 0 : (endnum + step - 1) / step)) return kInternal;
    const bool s32 = i - 1 == 0;
    // Blocks whose input span and outputs lie in the caller's buffers as plain interleaved frames go to the lean
    // kernel (fused_fast.hip); the others (the block that straddles ring and buffer, ring wrap, odd channel counts,
    // fp64 rings on either side) to the generic one.  At most three launches: generic head, lean middle, generic tail.
    int f0 = 0, f1 = 0;
    FastIo io = {};
    if (split) {
      // Sub-blocked form: ONE launch of nblocks * nsub workgroups per channel pair.  Every sub-block's 4096-frame window lies
      // inside its block's own input span, i.e. in the caller's buffer or, below it, in fifo 0's ring; outputs go to the next
      // fifo's fp64 ring at any position.  (Decided when the handle was opened: first stage, even channels, not the last stage.)
      if (!s32 || (nch_ & 1) || !ein.ptr || (ein.stride_floats & 1)) return kInternal;
      io.in = ein.ptr;
      io.in_ring = static_cast<const float *>(rings_[0].buf);
      io.in_ring_mask = rings_[0].cap - 1;
      io.in_ring_stream_stride = rings_[0].cap * nch_;
      io.in_abs0 = ein.begin;
      io.in_stream_stride = ein.stride_floats;
      io.nch = nch_;
      io.in_unaligned = (reinterpret_cast<uintptr_t>(ein.ptr) & 7) ? 1 : 0;
      int omode = 1;
      if (!dst_f32) {
        io.out64 = static_cast<double *>(rings_[i + 1].buf);
        io.out64_mask = rings_[i + 1].cap - 1;
        io.out64_chan_stride = rings_[i + 1].cap;
      } else {
        // float frames out: straight into the caller's buffer when every output of this launch lies inside it (a flow / a
        // mirrored push), else output by output wherever the fifo has it (RR_push: everything into the ring)
        const bool ext_ok = eout.ptr && !(eout.stride_floats & 1);
        const bool all_inside = ext_ok && wro >= eout.begin && wro + count <= eout.end;
        omode = (all_inside && !(reinterpret_cast<uintptr_t>(eout.ptr) & 7)) ? 0 : 2;
        io.out = ext_ok ? eout.ptr : nullptr;
        io.out_abs0 = ext_ok ? eout.begin : 0;
        io.out_end = ext_ok ? eout.end : 0; // (empty range: every output goes to the ring)
        io.out_stream_stride = ext_ok ? eout.stride_floats : 0;
        io.out_unaligned = (ext_ok && (reinterpret_cast<uintptr_t>(eout.ptr) & 7)) ? 1 : 0;
        io.out_ring = static_cast<float *>(rings_[i + 1].buf);
        io.out_ring_mask = rings_[i + 1].cap - 1;
        io.out_ring_stream_stride = rings_[i + 1].cap * nch_;
        if (eout.ptr && !ext_ok) return kInternal; // (an odd frame stride with an even channel count cannot happen)
        if (omode == 2 && !io.out_ring) return kInternal;
      }
      FusedArgs fr = fa;
      fr.d.G = fu.Gs;
      fr.d.tw_fwd = fr.d.tw_inv = twiddles(12);
      if (!fr.d.tw_fwd) return kNoMem;
      fr.d.nsub = fu.nsub;
      fr.d.Vs = fu.Vs;
      fr.d.two = split_two ? 1 : 0;
      fr.d.Pref = 1 << pend_log2p;
      fr.d.Bref0 = pend.B0;
      fr.d.B0 = pend.B0 * fu.nsub;
      fr.d.nblocks = ntab;
      fa = fr; // seam_kernel below: sub-block indices, same table
      const int pi = prof_begin(true);
      const char *kn = nullptr;
      HIP_TRY(launch_fused_split(omode, fr, io, stream_, &kn));
      prof_name(pi, kn);
      prof_end(pi);
    } else
    if (fu.cfm && s32 && dst_f32 && !(nch_ & 1) && ein.ptr && eout.ptr && fused_fast_supported(pend_log2n, pend_log2p, fu.KS) &&
        !(reinterpret_cast<uintptr_t>(ein.ptr) & 7) && !(reinterpret_cast<uintptr_t>(eout.ptr) & 7) && !(ein.stride_floats & 1) &&
        !(eout.stride_floats & 1)) {
      const long long P = 1LL << pend_log2p, q = fa.d.q;
      // (a block that starts below the caller's buffer takes its head from fifo 0's ring: the lean kernel handles that too)
      const long long hi = (ein.end - P) >= 0 ? (ein.end - P) / q - pend.B0 + 1 : 0;
      f0 = 0;
      f1 = int(std::min<long long>(pend.nblocks, hi));
      // outputs of blocks [f0, f1) must lie inside the caller's output buffer
      while (f0 < f1) {
        const FusedBlock b0 = fused_block_info(pa, f0), b1 = fused_block_info(pa, f1 - 1);
        if (out_offset + b0.i_lo < eout.begin) { ++f0; continue; }
        if (out_offset + b1.i_lo + b1.cnt > eout.end) { --f1; continue; }
        break;
      }
      if (f0 >= f1) f0 = f1 = 0;
      io.in = ein.ptr;
      io.in_ring = static_cast<const float *>(rings_[0].buf);
      io.in_ring_mask = rings_[0].cap - 1;
      io.in_ring_stream_stride = rings_[0].cap * nch_;
      io.out = eout.ptr;
      io.in_abs0 = ein.begin;
      io.out_abs0 = eout.begin;
      io.in_stream_stride = ein.stride_floats;
      io.out_stream_stride = eout.stride_floats;
      io.nch = nch_;
    } else if (fu.cfm && s32 && !dst_f32 && !(nch_ & 1) && ein.ptr && fused_fast_supported(pend_log2n, pend_log2p, fu.KS) &&
               !(reinterpret_cast<uintptr_t>(ein.ptr) & 7) && !(ein.stride_floats & 1)) {
      // the polyphase stage feeds another stage: same lean kernel, its outputs into the next fifo's fp64 ring (any ring
      // position: the kernel masks the index), so only the input side limits the range
      const long long P = 1LL << pend_log2p, q = fa.d.q;
      const long long hi = (ein.end - P) >= 0 ? (ein.end - P) / q - pend.B0 + 1 : 0;
      f0 = 0;
      f1 = int(std::max<long long>(0, std::min<long long>(pend.nblocks, hi)));
      io.in = ein.ptr;
      io.in_ring = static_cast<const float *>(rings_[0].buf);
      io.in_ring_mask = rings_[0].cap - 1;
      io.in_ring_stream_stride = rings_[0].cap * nch_;
      io.in_abs0 = ein.begin;
      io.in_stream_stride = ein.stride_floats;
      io.nch = nch_;
      io.out64 = static_cast<double *>(rings_[i + 1].buf);
      io.out64_mask = rings_[i + 1].cap - 1;
      io.out64_chan_stride = rings_[i + 1].cap;
    }
    auto launch_range = [&](int b0, int b1, bool fast) -> int {
      if (b0 >= b1) return kOk;
      FusedArgs fr = fa;
      fr.d.B0 = fa.d.B0 + b0;
      fr.d.nblocks = b1 - b0;
      fr.blk = fa.blk + b0;
      const int pi = prof_begin(true);
      const char *kn = nullptr;
      if (fast) HIP_TRY(launch_fused_fast(pend_log2p, fr, io, stream_, &kn));
      else
        HIP_TRY(launch_fused(pend_log2n, pend_log2p, s32, dst_f32, s32 ? f32_view(0, &ein, nullptr) : nof,
                             s32 ? nod : f64_view(i - 1), dst_f32 ? f32_view(i + 1, nullptr, &eout) : nof,
                             dst_f32 ? nod : f64_view(i + 1), fr, stream_, &kn));
      prof_name(pi, kn);
      prof_end(pi);
      return kOk;
    };
    if (!split) {
      { int rl = launch_range(0, f0, false); if (rl) return rl; }
      { int rl = launch_range(f0, f1, true); if (rl) return rl; }
      { int rl = launch_range(f1, pend.nblocks, false); if (rl) return rl; }
    }
    if (!seam_on_side) {
      { int rcj = join_side(); if (rcj) return rcj; } // (a seam kernel of an earlier launch of this push may still be on the side stream)
      const int ps = prof_begin(false, "rsmp::seam_kernel");
      HIP_TRY(launch_seam(dst_f32, dst_f32 ? f32_view(i + 1, nullptr, &eout) : nof, dst_f32 ? nod : f64_view(i + 1), fa, stream_));
      prof_end(ps);
    } else {
      HIP_TRY(hipEventRecord(ev_fused_, stream_));
      HIP_TRY(hipStreamWaitEvent(side_, ev_fused_, 0));
      HIP_TRY(launch_seam(dst_f32, dst_f32 ? f32_view(i + 1, nullptr, &eout) : nof, dst_f32 ? nod : f64_view(i + 1), fa, side_));
      HIP_TRY(hipEventRecord(ev_seam_[seam_launches_ & 1], side_));
      ++seam_launches_;
      side_pending_ = true;
    }
  }
  return kOk;
}

// rational polyphase stage on the matrix pipe as a stage of its own (polymf.hip)
int Engine::launch_polymf_stage(Pass &ps, int i, long long count, long long step)
{
  RSMP_STAGE_LOCALS(ps, i);
  int rc = ensure_ring(i + 1, dst_need(wro + count));
  if (rc) return rc;
  const PolyMf &pm = polymf_[i];
  const long long at0 = sp.at0 >> 32, i_begin = wro_before - out_offset, i_end = i_begin + count;
  const long long q_first = (at0 + i_begin * step) / sp.L, q_last = (at0 + (i_end - 1) * step) / sp.L;
  for (long long t0 = q_first / pm.Vt, t_end = q_last / pm.Vt + 1; t0 < t_end; t0 += pm.blk_cap) {
    FusedPrepArgs pa; // per-tile bookkeeping: the closed forms of the fused path with V = Vt, no seam exclusion
    pa.b_offset = 0;
    pa.B0 = t0;
    pa.at0 = at0;
    pa.V = pm.Vt;
    pa.polyL = sp.L;
    pa.step = int(step);
    pa.n = 1;
    pa.nblocks = int(std::min<long long>(pm.blk_cap, t_end - t0));
    pa.two_round = 0;
    pa.ra_end = pa.rb_start = 0;
    pa.nsub = pa.Vs = 0;
    pa.KS = pm.KS;
    pa.qb_max = 0;
    pa.qb_min = 0;
    pa.clip_lo = i_begin;
    pa.clip_hi = i_end;
    for (int k : {0, pa.nblocks - 1})
      if (fused_block_info(pa, k).K > 32) return kInternal; // 4 column steps x 2 halves x 4 periods
    { const int pp = prof_begin(false, "rsmp::fused_prep_kernel"); HIP_TRY(launch_fused_prep(pa, pm.blk, stream_)); prof_end(pp); }
    PolyMfArgs a;
    a.cfm = pm.cfm;
    a.qtab = pm.qtab;
    a.blk = pm.blk;
    a.B0 = t0;
    a.at0 = at0;
    a.out_offset = out_offset;
    a.in_limit = b.wr[i];
    a.nblocks = pa.nblocks;
    a.C = C_;
    a.Vt = pm.Vt;
    a.n = sp.n;
    a.polyL = sp.L;
    a.step = int(step);
    a.NGRP = pm.NGRP;
    a.nchs = pair_nchs();
    const int pi = prof_begin(false);
    const char *kn = nullptr;
    HIP_TRY(launch_polymf(pm.KS, src_f32, dst_f32, src_f32 ? f32_view(i, &ein, nullptr) : nof, src_f32 ? nod : f64_view(i),
                          dst_f32 ? f32_view(i + 1, nullptr, &eout) : nof, dst_f32 ? nod : f64_view(i + 1), a, stream_, &kn));
    prof_name(pi, kn);
    prof_end(pi);
  }
  return kOk;
}

// the generic polyphase kernels: poly_kernel<order>, poly_coop_kernel, polyi_kernel (kernels.hip)
int Engine::launch_poly_stage(Pass &ps, int i, long long count, long long step)
{
  RSMP_STAGE_LOCALS(ps, i);
  int rc = ensure_ring(i + 1, dst_need(wro + count));
  if (rc) return rc;
  PolyArgs a;
  a.tab = d_poly_;
  a.rd = rd_before;
  a.at = st.at;
  a.step = step;
  a.out_abs = wro_before;
  a.count = count;
  a.C = C_;
  a.n = sp.n;
  a.L = sp.L;
  a.phase_bits = sp.phase_bits;
  const double in_per_out = sp.order == 0 ? double(step) / sp.L : double(step) / 4294967296.0;
  // LDS: window of the tile, plus (rational stages) the whole coefficient table when both fit 64 KB;
  // the kernel's 32-bit clock needs tile * step < 2^31
  const size_t tab_bytes = sp.order == 0 ? size_t(sp.L) * sp.n * 8 : 0;
  const bool tab_lds = sp.order == 0 && tab_bytes <= 40 * 1024;
  const double lds_for_window = tab_lds ? double(64 * 1024 - 16) - double(tab_bytes) : 48.0 * 1024;
  int tile = 2048;
  while (tile > 256 && ((tile * in_per_out + sp.n + 4) * 8 > lds_for_window ||
                        (sp.order == 0 && double(tile + 256) * double(step) >= 2147483648.0)))
    tile >>= 1;
  if (sp.order == 0 && double(tile + 256) * double(step) >= 2147483648.0) return kInternal;
  a.tile = tile;
  a.win = (int(tile * in_per_out) + sp.n + 4 + 1) & ~1;
  a.tab_lds = tab_lds;
  a.coop = sp.order >= 1 && sp.n % 8 == 0 && !knobs().no_polycoop;
  a.shared_rows = 0;
  if (sp.order >= 1 && sp.n <= 32 && !knobs().no_polyi) { // (polyi_kernel maps 32 lanes along a coefficient row: n <= 32) // the channels of a handle share the clock: share the interpolated rows
    const double wl = 128 * in_per_out + sp.n + 4; // window of a 128-output tile (kPolyiTile)
    if ((128.0 * sp.n + 16.0 * (wl + 1)) * 8 <= 150.0 * 1024) { // rows + 16 channel windows must fit LDS
      a.shared_rows = 1;
      a.tile = 128;
      a.win = int(wl) | 1;
    }
  }
  const int pi = prof_begin(false);
  const char *kn = nullptr;
  HIP_TRY(launch_poly(sp.order, src_f32, dst_f32, src_f32 ? f32_view(i, &ein, nullptr) : nof,
                      src_f32 ? nod : f64_view(i), dst_f32 ? f32_view(i + 1, nullptr, &eout) : nof,
                      dst_f32 ? nod : f64_view(i + 1), a, stream_, &kn));
  prof_name(pi, kn);
  prof_end(pi);
  return kOk;
}

// h8..h13 (rate_filters_generic.h:31-262)
int Engine::advance_half(Pass &ps, int i)
{
  RSMP_STAGE_LOCALS(ps, i);
  const long long avail = std::max<long long>(0, occ - sp.pre_post);
  const long long num_out = (avail + 1) / 2; // rate_filters_generic.h:83
  if (launch && num_out) {
    int rc = ensure_ring(i + 1, dst_need(wro + num_out));
    if (rc) return rc;
    HalfArgs a;
    a.rd = rd_before;
    a.out_abs = wro_before;
    a.count = num_out;
    a.C = C_;
    a.ncoef = sp.hb_n;
    a.pre = sp.pre;
    for (int k = 0; k < 13; ++k) a.coef[k] = k < sp.hb_n ? sp.hb[k] : 0.0;
    const int pi = prof_begin(false);
    const char *kn = nullptr;
    HIP_TRY(launch_half(src_f32, dst_f32, src_f32 ? f32_view(i, &ein, nullptr) : nof, src_f32 ? nod : f64_view(i),
                        dst_f32 ? f32_view(i + 1, nullptr, &eout) : nof, dst_f32 ? nod : f64_view(i + 1), a, stream_, &kn));
    prof_name(pi, kn);
    prof_end(pi);
  }
  if (2 * num_out <= occ) rd += 2 * num_out; // fifo_read refuses to over-read, fifo.h:169
  wro += num_out;
  return kOk;
}
#undef RSMP_STAGE_LOCALS

// Append `isamp` frames that live in device memory at d_in ([stream][frame][ch], `stride_frames`
// between streams) and run the chain.  If d_out is given and the output fifo is empty, up to
// out_cap produced frames are written straight into d_out (and counted as pulled).
int Engine::feed(const float *d_in, size_t stride_frames, size_t isamp, float *d_out, size_t out_stride, size_t out_cap,
                 size_t *direct_out, bool keep_direct)
{
  if (poisoned_) return kInternal;
  // from here on the counters move together with the launches: any failure leaves them skewed -> poison the handle
  return fail(feed_impl(d_in, stride_frames, isamp, d_out, out_stride, out_cap, direct_out, keep_direct));
}

// The frames the mirror still holds move into the device ring of the output fifo (same absolute indices), so that every
// other reader finds the whole fifo in one place.
int Engine::spill_mirror()
{
  const int f = int(rings_.size()) - 1;
  const long long base = mir_begin_, a0 = std::max(book_.rd[f], mir_begin_), a1 = mir_end_;
  mir_begin_ = mir_end_ = 0;
  if (a1 <= a0) return kOk;
  int rc = ensure_ring(f, book_.wr[f] - book_.rd[f]);
  if (rc) return rc;
  F32View dst = f32_view(f, nullptr, nullptr), src = dst;
  src.ext = pin_mir_.p; // frame `base` of stream 0
  src.ext_begin = base;
  src.ext_end = a1;
  src.ext_stream_stride = (long long)mir_stride_ * nch_;
  F64View nod = {};
  HIP_TRY(launch_copy(true, src, nod, dst, nod, a0, a1, C_, stream_));
  return kOk;
}

int Engine::feed_impl(const float *d_in, size_t stride_frames, size_t isamp, float *d_out, size_t out_stride, size_t out_cap,
                      size_t *direct_out, bool keep_direct)
{
  if (!keep_direct && mir_end_ > mir_begin_) { // (a push that does not mirror, a device push / flow: one place for the fifo)
    int rcs = spill_mirror();
    if (rcs) return rcs;
  }
  if (direct_out) *direct_out = 0;
  ExtIn ein;
  ein.ptr = d_in;
  ein.begin = book_.wr[0];
  ein.end = ein.begin + (long long)isamp;
  ein.stride_floats = (long long)stride_frames * nch_;
  ExtOut eout;
  const long long wr_out0 = book_.wr.back();
  if (d_out && out_cap && book_.rd.back() == wr_out0) {
    eout.ptr = d_out;
    eout.begin = wr_out0;
    eout.end = wr_out0 + (long long)out_cap;
    eout.stride_floats = (long long)out_stride * nch_;
  }
  size_t done = 0;
  while (done < isamp) {
    const size_t n = std::min(slab_frames_, isamp - done);
    int rc = advance(book_, n, true, ein, eout, done + n < isamp);
    if (rc) return rc;
    done += n;
  }
  // carry the part of this push that no stage has consumed yet into ring 0 (while the seam kernels, which only write
  // stage outputs, still run on the side stream)
  const long long a0 = std::max(book_.rd[0], ein.begin), a1 = book_.wr[0];
  if (a1 > a0) {
    int rc = ensure_ring(0, book_.wr[0] - book_.rd[0]);
    if (rc) return rc;
    F32View src = f32_view(0, &ein, nullptr), dst = f32_view(0, nullptr, nullptr);
    F64View nod = {};
    HIP_TRY(launch_copy(true, src, nod, dst, nod, a0, a1, C_, stream_));
  }
  { int rcj = join_side(); if (rcj) return rcj; }
  if (eout.ptr) {
    const long long produced = book_.wr.back() - wr_out0;
    const long long direct = std::min<long long>(produced, (long long)out_cap);
    if (!keep_direct) {
      book_.rd.back() += direct;
      book_.samples_out += size_t(direct);
    }
    if (direct_out) *direct_out = size_t(direct);
  }
  return kOk;
}

int Engine::push_device(const float *ibuf, size_t stream_stride, size_t isamp)
{
  if (poisoned_) return kInternal;
  if (!ibuf || !isamp) return kOk; // rate_base.h:623
  if (isamp > plan_.isamp_max) isamp = plan_.isamp_max; // silently truncated, rate_base.h:624
  return feed(ibuf, S_ > 1 ? stream_stride : isamp, isamp, nullptr, 0, 0, nullptr);
}

int Engine::push_host(const float *ibuf, size_t stream_stride, size_t isamp)
{
  if (poisoned_) return kInternal;
  if (!ibuf || !isamp) return kOk;
  if (isamp > plan_.isamp_max) isamp = plan_.isamp_max;
  const size_t need = isamp * size_t(nch_) * size_t(S_);
  const size_t row = isamp * nch_ * sizeof(float);
  if (need * sizeof(float) <= kZeroCopyMaxBytes) {
    // plugin-sized push: the kernels read it in place from the page-locked slot; the slot is released by an event behind the
    // push's last kernel (two slots alternate, so this only ever waits for the push before the previous one)
    Pinned &slot = pin_in_[pin_k_ ^= 1];
    if (slot.pending) {
      HIP_TRY(hipEventSynchronize(slot.done));
      slot.pending = false;
    }
    int rp = pinned_reserve(slot, need);
    if (rp) return rp;
    for (int s = 0; s < S_; ++s) std::memcpy(slot.p + size_t(s) * isamp * nch_, ibuf + size_t(s) * stream_stride * nch_, row);
    // mirror the output when the fifo is empty (the plugin pulls until it is): the exact number of frames this push makes
    // available comes from a dry run of the counters
    float *mir = nullptr;
    size_t cap = 0;
    if (!plan_.stages.empty() && available() == 0 && mir_end_ == mir_begin_) { // (no stages: the input ring IS the output fifo)
      Book trial = book_;
      const ExtIn no_in;
      const ExtOut no_out;
      for (size_t done = 0; done < isamp;) {
        const size_t n = std::min(slab_frames_, isamp - done);
        int rc = advance(trial, n, false, no_in, no_out);
        if (rc) return rc;
        done += n;
      }
      cap = size_t(trial.wr.back() - trial.rd.back());
      if (cap && cap * size_t(nch_) * size_t(S_) * sizeof(float) <= kZeroCopyMaxBytes) {
        int rm = pinned_reserve(pin_mir_, cap * size_t(nch_) * size_t(S_));
        if (rm) return rm;
        mir = pin_mir_.p;
      }
    }
    size_t direct = 0;
    const long long wr0 = book_.wr.back();
    int rc = feed(slot.p, isamp, isamp, mir, cap, mir ? cap : 0, &direct, true);
    if (rc) return rc;
    if (mir) {
      mir_begin_ = wr0;
      mir_end_ = wr0 + (long long)direct;
      mir_stride_ = cap;
      HIP_TRY(hipEventRecord(pin_mir_.done, stream_)); // what RR_pull waits for
      pin_mir_.pending = true;
    }
    HIP_TRY(hipEventRecord(slot.done, stream_));
    slot.pending = true;
    return kOk;
  }
  if (need > stage_floats_) {
    HIP_TRY(hipStreamSynchronize(stream_));
    if (d_stage_) (void)hipFree(d_stage_);
    d_stage_ = nullptr;
    stage_floats_ = 0;
    ALLOC_TRY(&d_stage_, need * sizeof(float));
    stage_floats_ = need;
  }
  if (need * sizeof(float) <= kPinnedMaxBytes) {
    Pinned &slot = pin_in_[pin_k_ ^= 1];
    if (slot.pending) { // the copy that last used this slot (two pushes ago) must have left it
      HIP_TRY(hipEventSynchronize(slot.done));
      slot.pending = false;
    }
    int rp = pinned_reserve(slot, need);
    if (rp) return rp;
    for (int s = 0; s < S_; ++s) std::memcpy(slot.p + size_t(s) * isamp * nch_, ibuf + size_t(s) * stream_stride * nch_, row);
    HIP_TRY(hipMemcpyAsync(d_stage_, slot.p, need * sizeof(float), hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipEventRecord(slot.done, stream_));
    slot.pending = true;
  } else if (S_ == 1) HIP_TRY(hipMemcpyAsync(d_stage_, ibuf, row, hipMemcpyHostToDevice, stream_));
  else HIP_TRY(hipMemcpy2DAsync(d_stage_, row, ibuf, stream_stride * nch_ * sizeof(float), row, S_, hipMemcpyHostToDevice, stream_));
  return feed(d_stage_, isamp, isamp, nullptr, 0, 0, nullptr);
}

// copy `frames` frames starting at the output read pointer to dst; does not move the pointer
int Engine::copy_out(float *dst, size_t stride_frames, size_t frames, bool to_host)
{
  const int f = int(rings_.size()) - 1;
  const Ring &r = rings_[f];
  const long long rd = book_.rd[f];
  if (to_host) {
    const size_t total = frames * size_t(nch_) * size_t(S_) * sizeof(float);
    const bool pinned = total <= kPinnedMaxBytes;
    if (pinned) {
      int rp = pinned_reserve(pin_out_, total / sizeof(float));
      if (rp) return rp;
    }
    size_t done = 0;
    while (done < frames) { // at most two segments (ring wrap)
      const long long pos = (rd + (long long)done) & (r.cap - 1);
      const size_t n = std::min<size_t>(frames - done, size_t(r.cap - pos));
      const size_t row = n * nch_ * sizeof(float);
      const float *src = static_cast<const float *>(r.buf) + pos * nch_;
      float *d = pinned ? pin_out_.p + done * nch_ : dst + done * nch_;
      const size_t dpitch = (pinned ? frames : stride_frames) * nch_ * sizeof(float);
      HIP_TRY(hipMemcpy2DAsync(d, dpitch, src, size_t(r.cap) * nch_ * sizeof(float), row, S_, hipMemcpyDeviceToHost, stream_));
      done += n;
    }
    HIP_TRY(hipStreamSynchronize(stream_));
    if (pinned)
      for (int s = 0; s < S_; ++s)
        std::memcpy(dst + size_t(s) * stride_frames * nch_, pin_out_.p + size_t(s) * frames * nch_, frames * nch_ * sizeof(float));
    free_garbage(); // everything queued before this point has finished: retired rings / drain buffers can go
  } else {
    ExtOut eo;
    eo.ptr = dst;
    eo.begin = rd;
    eo.end = rd + (long long)frames;
    eo.stride_floats = (long long)stride_frames * nch_;
    F32View src = f32_view(f, nullptr, nullptr), dv = f32_view(f, nullptr, &eo);
    F64View nod = {};
    HIP_TRY(launch_copy(true, src, nod, dv, nod, rd, rd + (long long)frames, C_, stream_));
  }
  return kOk;
}

int Engine::pull_host(float *obuf, size_t stream_stride, size_t osamp, size_t *ogen)
{
  if (poisoned_) {
    if (ogen) *ogen = 0;
    return kInternal;
  }
  if (!obuf || !osamp) { // rate_base.h:647
    if (ogen) *ogen = 0;
    return kOk;
  }
  const size_t n = std::min(osamp, available());
  if (n) {
    const long long rd = book_.rd.back();
    if (mir_end_ > mir_begin_ && rd >= mir_begin_ && rd + (long long)n <= mir_end_) {
      // everything asked for sits in the page-locked mirror of the last push: wait for that push, copy on the CPU
      if (pin_mir_.pending) {
        HIP_TRY(hipEventSynchronize(pin_mir_.done));
        pin_mir_.pending = false;
      }
      const size_t dstride = S_ > 1 ? stream_stride : n;
      for (int s = 0; s < S_; ++s)
        std::memcpy(obuf + size_t(s) * dstride * nch_, pin_mir_.p + (size_t(s) * mir_stride_ + size_t(rd - mir_begin_)) * nch_,
                    n * nch_ * sizeof(float));
      if (rd + (long long)n == mir_end_) mir_begin_ = mir_end_ = 0;
    } else {
      if (mir_end_ > mir_begin_) { // the request reaches past the mirror: the device ring takes what is left of it first
        int rcs = fail(spill_mirror());
        if (rcs) return rcs;
      }
      int rc = copy_out(obuf, S_ > 1 ? stream_stride : n, n, true);
      if (rc) return rc;
    }
    book_.rd.back() += (long long)n;
    book_.samples_out += n; // rate_base.h:448
  }
  if (ogen) *ogen = n;
  return kOk;
}

int Engine::pull_device(float *obuf, size_t stream_stride, size_t osamp, size_t *ogen)
{
  if (poisoned_) {
    if (ogen) *ogen = 0;
    return kInternal;
  }
  if (!obuf || !osamp) {
    if (ogen) *ogen = 0;
    return kOk;
  }
  const size_t n = std::min(osamp, available());
  if (n) {
    if (mir_end_ > mir_begin_) {
      int rcs = fail(spill_mirror());
      if (rcs) return rcs;
    }
    int rc = copy_out(obuf, S_ > 1 ? stream_stride : n, n, false);
    if (rc) return rc;
    book_.rd.back() += (long long)n;
    book_.samples_out += n;
  }
  if (ogen) *ogen = n;
  return kOk;
}

// rate_base.h:571-614: deliver what is ready, take the input, deliver again
int Engine::flow_host(const float *ibuf, size_t in_stride, float *obuf, size_t out_stride, size_t isamp, size_t osamp,
                      size_t *iused, size_t *ogen)
{
  size_t n1 = 0, n2 = 0;
  if (!ibuf) isamp = 0;
  if (isamp > plan_.isamp_max) isamp = plan_.isamp_max;
  int rc = pull_host(obuf, out_stride, osamp, &n1);
  if (rc) return rc;
  if (isamp && (rc = push_host(ibuf, in_stride, isamp))) return rc;
  if (n1 < osamp && obuf && (rc = pull_host(obuf + n1 * nch_, out_stride, osamp - n1, &n2))) return rc;
  if (iused) *iused = isamp;
  if (ogen) *ogen = n1 + n2;
  return kOk;
}

int Engine::flow_device(const float *ibuf, size_t in_stride, float *obuf, size_t out_stride, size_t isamp, size_t osamp,
                        size_t *iused, size_t *ogen)
{
  size_t n1 = 0, n2 = 0;
  if (!ibuf) isamp = 0;
  if (isamp > plan_.isamp_max) isamp = plan_.isamp_max;
  if (S_ > 1 && obuf && out_stride < osamp) return kInvParam;
  int rc = pull_device(obuf, out_stride, osamp, &n1);
  if (rc) return rc;
  if (isamp) {
    // frames produced by this push land directly in the caller's buffer (no ring round trip)
    float *direct = obuf && n1 < osamp ? obuf + n1 * nch_ : nullptr;
    rc = feed(ibuf, S_ > 1 ? in_stride : isamp, isamp, direct, S_ > 1 ? out_stride : osamp, direct ? osamp - n1 : 0, &n2);
    if (rc) return rc;
    if (direct && !n2 && available() && n1 < osamp) { // the ring was not empty: fall back to a copy
      size_t n3 = 0;
      if ((rc = pull_device(direct, out_stride, osamp - n1, &n3))) return rc;
      n2 = n3;
    }
  }
  if (iused) *iused = isamp;
  if (ogen) *ogen = n1 + n2;
  return kOk;
}

// rate_base.h:454-468,662-672
int Engine::drain()
{
  if (poisoned_) return kInternal;
  const size_t target = size_t(double(book_.samples_in) / plan_.factor + .5);
  if (target <= book_.samples_out) return kOk;
  const size_t remaining = target - book_.samples_out;
  // how many 1024-frame blocks of silence the reference would feed: counters only
  Book trial = book_;
  size_t blocks = 0;
  const ExtIn no_in;
  const ExtOut no_out;
  while (size_t(trial.wr.back() - trial.rd.back()) < remaining) {
    int rc = advance(trial, 1024, false, no_in, no_out);
    if (rc) return rc;
    if (++blocks > (1u << 20)) return fail(kInternal);
  }
  if (blocks) {
    const size_t frames = blocks * 1024, floats = frames * size_t(nch_) * size_t(S_);
    float *zeros = nullptr;
    ALLOC_TRY(&zeros, floats * sizeof(float));
    HIP_TRY(hipMemsetAsync(zeros, 0, floats * sizeof(float), stream_));
    // feed them one reference block at a time so that the counter wrap in rate_input sees the same sequence
    for (size_t k = 0; k < blocks; ++k) {
      int rc = feed(zeros + k * 1024 * nch_, frames, 1024, nullptr, 0, 0, nullptr);
      if (rc) { garbage_.push_back(zeros); return fail(rc); }
    }
    garbage_.push_back(zeros);
  }
  book_.trimmed += book_.wr.back() - (book_.rd.back() + (long long)remaining);
  book_.wr.back() = book_.rd.back() + (long long)remaining; // fifo_trim_to
  if (mir_end_ > book_.wr.back()) mir_end_ = std::max(mir_begin_, book_.wr.back()); // (the trim can cut into frames the host mirror holds)
  book_.samples_in = book_.samples_out = 0;
  return kOk;
}

} // namespace rsmp
