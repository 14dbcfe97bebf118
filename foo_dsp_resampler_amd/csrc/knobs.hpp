// Run-time and build-time switches of the library, in ONE place.
//
// Run-time knobs (environment, read ONCE per process by knobs(); nothing on the launch path calls getenv).  Every one of
// them only selects between kernel variants that produce the same samples to the parity bar, or sizes a buffer:
//   RSMP_NO_FUSE / RSMP_NO_MFMA / RSMP_NO_POLYMF / RSMP_NO_FAST / RSMP_NO_SPLIT / RSMP_NO_SPLIT2 / RSMP_NO_DFTX / RSMP_NO_POLYI /
//   RSMP_NO_POLYCOOP /
//   RSMP_SPREAD_VECTOR    keep a chain off the named fast variant (the generic variant of the same stage runs instead)
//   RSMP_NO_SIDE          seam kernels on the main stream instead of the side stream
//   RSMP_NO_GRAPH         small pushes launch their kernels one by one instead of replaying a captured HIP graph
//   RSMP_SLAB_MB=n        fp64 fifo budget of a time slab (default 1536)
//   RSMP_SEAM_RING_MB=n   budget of a fused chain's seam ring (default 1280): bounds the blocks per launch (tests force many launches per push)
//   RSMP_STAMPS=1         per-phase cycle sums of the fused kernels (s_memtime), printed when the handle closes
//   RSMP_LDS_PAD=n / RSMP_OCC=1   occupancy experiments of fused_kernel (more LDS per workgroup / print blocks per CU)
//   RATELIB_AMD_DEVICES=all | i,j,...   RR_open / RRX_open_batch deal new handles round-robin over these devices
//                         (unset: a handle lives on the calling thread's current device)
//
// Build-time experiment switches (RSMP_EXP_*, RSMP_DFTX_SKIP, the RSMP_DBG ablation bits) produce WRONG results by design:
// they remove one suspect from a kernel to bound what it costs.  They exist only in builds made with -DRSMP_EXPERIMENTS
// (tools/build_variant.sh); the product Makefile never sets it, and without it every such macro is forced to 0 here.
#pragma once
#include <cstddef>

#ifndef RSMP_EXPERIMENTS
#if (defined(RSMP_EXP_TAB) && RSMP_EXP_TAB) || (defined(RSMP_EXP_LINEAR) && RSMP_EXP_LINEAR) ||       \
    (defined(RSMP_EXP_HALFMFMA) && RSMP_EXP_HALFMFMA) || (defined(RSMP_EXP_TWK0) && RSMP_EXP_TWK0) || \
    (defined(RSMP_EXP_TWLOAD) && RSMP_EXP_TWLOAD) || (defined(RSMP_EXP_NOBAR) && RSMP_EXP_NOBAR) ||   \
    (defined(RSMP_DFTX_SKIP) && RSMP_DFTX_SKIP) || (defined(RSMP_EXP_SKIP) && RSMP_EXP_SKIP)
#error "wrong-result experiment switches need -DRSMP_EXPERIMENTS (tools/build_variant.sh); the product build never sets them"
#endif
#undef RSMP_EXP_TAB
#undef RSMP_EXP_LINEAR
#undef RSMP_EXP_HALFMFMA
#undef RSMP_EXP_TWK0
#undef RSMP_EXP_TWLOAD
#undef RSMP_EXP_NOBAR
#undef RSMP_DFTX_SKIP
#undef RSMP_EXP_SKIP
#define RSMP_EXP_SKIP 0
#define RSMP_EXP_TAB 0
#define RSMP_EXP_LINEAR 0
#define RSMP_EXP_HALFMFMA 0
#define RSMP_EXP_TWK0 0
#define RSMP_EXP_TWLOAD 0
#define RSMP_EXP_NOBAR 0
#define RSMP_DFTX_SKIP 0
#else
#ifndef RSMP_EXP_TAB
#define RSMP_EXP_TAB 0
#endif
#ifndef RSMP_EXP_LINEAR
#define RSMP_EXP_LINEAR 0
#endif
#ifndef RSMP_EXP_HALFMFMA
#define RSMP_EXP_HALFMFMA 0
#endif
#ifndef RSMP_EXP_TWK0
#define RSMP_EXP_TWK0 0
#endif
#ifndef RSMP_EXP_TWLOAD
#define RSMP_EXP_TWLOAD 0
#endif
#ifndef RSMP_EXP_NOBAR
#define RSMP_EXP_NOBAR 0
#endif
#ifndef RSMP_DFTX_SKIP
#define RSMP_DFTX_SKIP 0
#endif
#ifndef RSMP_EXP_SKIP
#define RSMP_EXP_SKIP 0
#endif
#endif

namespace rsmp {

struct Knobs {
  bool no_fuse = false, no_mfma = false, no_polymf = false, no_fast = false, no_split = false, no_split2 = false, no_dftx = false, no_polyi = false,
       no_polycoop = false, spread_vector = false, no_side = false, no_graph = false, stamps = false, occ = false,
       test_hooks = false;
  double slab_mb = 1536.0, seam_ring_mb = 1280.0;
  size_t lds_pad = 0;
  int dbg = 0; // RSMP_DBG ablation bits: honoured in -DRSMP_EXPERIMENTS builds only
};
// the process's knobs: the environment is read at the first call (thread-safe), never again
const Knobs &knobs();

} // namespace rsmp
