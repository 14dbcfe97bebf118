// Device-side fp64 complex FFT for one workgroup (gfx950).
//
// Layout: an M-point transform is held by T = M/16 threads, 16 points per thread, in registers:
// thread `tid` owns x[tid + s*T], s = 0..15 ("slots"), both on entry and on exit (natural order,
// no bit reversal: Stockham autosort).  Passes are radix-R0 (R0 = 2,4,8 or 16, twiddle-free) followed
// by radix-16 passes; between passes the 16 register values are exchanged through LDS
// (ds_write_b128 / ds_read_b128, or two 8-byte half exchanges in SPLIT mode so that a 16384-point
// transform fits 128 KiB of LDS).  The last pass leaves its results in registers.
//
// Twiddles come from a per-size table in HBM/L2 laid out [pass][r-1][k] so that consecutive lanes read
// consecutive 16-byte entries; entries are exp(+2 pi i r k / (Ns*16)), conjugated on the fly for DIR<0.
#pragma once
#include <hip/hip_runtime.h>

#include "knobs.hpp" // RSMP_EXP_* (wrong-result timing experiments): 0 unless the build says -DRSMP_EXPERIMENTS

namespace rsmp {

struct c64 { double x, y; };

__device__ __forceinline__ c64 cadd(c64 a, c64 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c64 csub(c64 a, c64 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c64 cmul(c64 a, c64 w) { return {fma(a.x, w.x, -a.y * w.y), fma(a.x, w.y, a.y * w.x)}; }
__device__ __forceinline__ c64 cmulc(c64 a, c64 w) { return {fma(a.x, w.x, a.y * w.y), fma(a.y, w.x, -a.x * w.y)}; } // a*conj(w)
// multiply by exp(DIR * i pi/2)
template <int DIR> __device__ __forceinline__ c64 rot90(c64 a) { return DIR > 0 ? c64{-a.y, a.x} : c64{a.y, -a.x}; }
// multiply by the unit-modulus constant (c + DIR*i*s)
template <int DIR> __device__ __forceinline__ c64 crot(c64 a, double c, double s)
{
  return DIR > 0 ? c64{fma(a.x, c, -a.y * s), fma(a.y, c, a.x * s)} : c64{fma(a.x, c, a.y * s), fma(a.y, c, -a.x * s)};
}

template <int DIR> __device__ __forceinline__ void bfly2(c64 &a, c64 &b)
{
  c64 t = csub(a, b);
  a = cadd(a, b);
  b = t;
}

// in: natural order a0..a3, out: natural order X0..X3
template <int DIR> __device__ __forceinline__ void bfly4(c64 &a0, c64 &a1, c64 &a2, c64 &a3)
{
  c64 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = rot90<DIR>(csub(a1, a3));
  a0 = cadd(t0, t2);
  a2 = csub(t0, t2);
  a1 = cadd(t1, t3);
  a3 = csub(t1, t3);
}

template <int R, int DIR> struct Bfly;

template <int DIR> struct Bfly<2, DIR> {
  static __device__ __forceinline__ void run(c64 (&b)[2]) { bfly2<DIR>(b[0], b[1]); }
};
template <int DIR> struct Bfly<4, DIR> {
  static __device__ __forceinline__ void run(c64 (&b)[4]) { bfly4<DIR>(b[0], b[1], b[2], b[3]); }
};
template <int DIR> struct Bfly<8, DIR> {
  static __device__ __forceinline__ void run(c64 (&b)[8])
  {
    const double h = 0.70710678118654752440;
    // even / odd 4-point transforms
    bfly4<DIR>(b[0], b[2], b[4], b[6]);
    bfly4<DIR>(b[1], b[3], b[5], b[7]);
    c64 o1 = crot<DIR>(b[3], h, h);   // W8^1
    c64 o2 = rot90<DIR>(b[5]);        // W8^2
    c64 o3 = crot<DIR>(b[7], -h, h);  // W8^3
    c64 e0 = b[0], e1 = b[2], e2 = b[4], e3 = b[6], o0 = b[1];
    b[0] = cadd(e0, o0); b[4] = csub(e0, o0);
    b[1] = cadd(e1, o1); b[5] = csub(e1, o1);
    b[2] = cadd(e2, o2); b[6] = csub(e2, o2);
    b[3] = cadd(e3, o3); b[7] = csub(e3, o3);
  }
};
template <int DIR> struct Bfly<16, DIR> {
  static __device__ __forceinline__ void run(c64 (&b)[16])
  {
    const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173; // cos, sin(pi/8)
    const double h = 0.70710678118654752440;
    // step 1: for each n2, 4-point transform over n1 of x[4 n1 + n2]  -> A[n2][k1] stored at b[4 k1 + n2]
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) bfly4<DIR>(b[n2], b[4 + n2], b[8 + n2], b[12 + n2]);
    // step 2: twiddle A[n2][k1] by W16^(n2 k1)
    b[4 + 1] = crot<DIR>(b[4 + 1], c1, s1);   // k1=1,n2=1 : W^1
    b[4 + 2] = crot<DIR>(b[4 + 2], h, h);     //        n2=2 : W^2
    b[4 + 3] = crot<DIR>(b[4 + 3], s1, c1);   //        n2=3 : W^3
    b[8 + 1] = crot<DIR>(b[8 + 1], h, h);     // k1=2,n2=1 : W^2
    b[8 + 2] = rot90<DIR>(b[8 + 2]);          //        n2=2 : W^4
    b[8 + 3] = crot<DIR>(b[8 + 3], -h, h);    //        n2=3 : W^6
    b[12 + 1] = crot<DIR>(b[12 + 1], s1, c1);   // k1=3,n2=1 : W^3
    b[12 + 2] = crot<DIR>(b[12 + 2], -h, h);    //        n2=2 : W^6
    b[12 + 3] = crot<DIR>(b[12 + 3], -c1, -s1); //        n2=3 : W^9
    // step 3: for each k1, 4-point transform over n2 -> X[k1 + 4 k2] lands at b[4 k1 + k2]
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) bfly4<DIR>(b[4 * k1], b[4 * k1 + 1], b[4 * k1 + 2], b[4 * k1 + 3]);
    // reorder b[4 k1 + k2] -> X[k1 + 4 k2]  (a 4x4 transpose of register names)
    c64 t;
    t = b[1]; b[1] = b[4]; b[4] = t;
    t = b[2]; b[2] = b[8]; b[8] = t;
    t = b[3]; b[3] = b[12]; b[12] = t;
    t = b[6]; b[6] = b[9]; b[9] = t;
    t = b[7]; b[7] = b[13]; b[13] = t;
    t = b[11]; b[11] = b[14]; b[14] = t;
  }
};

// first-pass radix for an M-point transform held as 16 points/thread
constexpr int fft_first_radix(int log2m) { return (log2m & 3) ? (1 << (log2m & 3)) : 16; }
constexpr int fft_num_passes(int log2m) { return (log2m + 3) / 4; }
// number of twiddle entries (c64) of the table for size 2^log2m: sum over passes p>=1 of 15*Ns_p
constexpr int fft_twiddle_count(int log2m)
{
  int n = 0, ns = fft_first_radix(log2m);
  for (int p = 1; p < fft_num_passes(log2m); ++p) {
    n += 15 * ns;
    ns *= 16;
  }
  return n;
}

// LDS image used by the exchange after the FIRST pass: that pass scatters with a lane stride of R0
// elements (R0*16 bytes = a multiple of the 128-byte write bank row for R0 >= 8): an 8-way bank
// conflict on ds_write_b128.  One 16-byte pad per R0 elements makes the lane stride R0+1 elements.
template <int R0> __device__ __forceinline__ constexpr int lds_phys(int i) { return R0 >= 8 ? i + i / R0 : i; }
// same for an index known to be non-negative (a shift instead of a signed division when the compiler cannot see the sign)
template <int R0> __device__ __forceinline__ constexpr int lds_phys_u(unsigned i) { return R0 >= 8 ? int(i + i / unsigned(R0)) : int(i); }
// doubles of LDS an M-point transform needs (non-split) including that padding
constexpr int fft_lds_doubles(int log2m)
{
  const int m = 1 << log2m, r0 = (log2m & 3) ? (1 << (log2m & 3)) : 16;
  return 2 * (r0 >= 8 ? m + m / r0 : m);
}
// same for the two-half-rounds exchange (MODE 2)
constexpr int fft_lds_doubles_halves(int log2m) { return fft_lds_doubles(log2m) / 2; }

// Exchange the 16 register values through LDS: value in slot s goes to Stockham position pos[s];
// afterwards slot s holds element tid + s*T.  All threads of the workgroup must call this
// (barriers); only `active` threads move data.
//
// MODE 0: one round of 16-byte elements (M*16 bytes of LDS, plus padding).
// MODE 1 ("split"): real and imaginary halves in two rounds of 8-byte elements (M*8 bytes).
// MODE 2 ("halves"): destinations [0, M/2) first, then [M/2, M), 16-byte elements both times (M*8 bytes plus
//         padding): same LDS traffic as mode 0, two more barriers, half the footprint.
// RSMP_EXP_NOBAR (timing experiments only, results are WRONG): exchanges after the first one of a transform run without
// their workgroup barriers -- an upper bound for what wave-local exchanges could save.
// default of the TWGEN template parameter of fft_regs / fft8_regs (every kernel that does not say otherwise)
#ifndef RSMP_TWGEN_DEFAULT
#define RSMP_TWGEN_DEFAULT 1
#endif
#ifndef RSMP_TWGEN_SQ
#define RSMP_TWGEN_SQ 0
#endif
// RSMP_EXP_TWK0 (timing experiment only, WRONG results; knobs.hpp): every twiddle load reads entry k = 0 of its row (one
// line per row: the loads stay, their misses and most of their latency go) -- upper bound of what twiddle tables in LDS
// could buy.  RSMP_EXP_TWLOAD: only the twiddles r = 1, 2, 4, 8 are loaded.
template <bool BAR> __device__ __forceinline__ void rsmp_xbar()
{
  if (BAR) (__syncthreads)();
}
// BYTHREAD (MODE 2 only): all 16 destinations of a thread lie in the same half of the image -- the lower one for tid < T / 2 --
// which holds for every radix-16 pass whose butterfly stride NS is at most T / 2 (destinations of thread tid fill
// [16 NS q, 16 NS (q + 1)), q = tid / NS).  The two rounds are then one uniform branch per wave each instead of 16 per-element
// range tests with predicated stores (~60 vector instructions per exchange).
template <int T, int MODE, int PADR = 1, bool BAR = true, bool BYTHREAD = false>
__device__ __forceinline__ void lds_exchange(c64 (&v)[16], const int (&pos)[16], int tid, bool active, double *lds)
{
#define __syncthreads() rsmp_xbar<BAR>()
  if (MODE == 2 && BYTHREAD) {
    constexpr int H = 8 * T; // M / 2
    double2 *l2 = reinterpret_cast<double2 *>(lds);
    const bool lower = tid < T / 2;
    c64 lo[8];
    // reads: element tid + s T sits at phys(tid) + s (T + T / pad period) (T is a multiple of the pad period)
    static_assert(PADR < 8 || T % PADR == 0, "lds_exchange: pad period must divide T");
    constexpr int RS = T + (PADR >= 8 ? T / PADR : 0);
    int rb = lds_phys_u<PADR>(unsigned(tid));
    asm volatile("" : "+v"(rb));
    const double2 *const src = l2 + rb;
    // A thread's destinations are pos[0] + (compile-time offsets): with the padded image (first pass: pos[0] = 16 tid, offsets
    // r < 16 = the pad period) phys(pos[0] + d) = phys(pos[0]) + d, without padding phys is the identity -- one address per
    // thread and round, the rest are immediates of the stores.
    // (the base index is made opaque: folded into other address arithmetic it leaves the stores offsets that are negative or
    // do not fit the 16-bit immediate, i.e. one add per store again)
    if (active && lower) {
      int b0 = lds_phys_u<PADR>(unsigned(pos[0]));
      asm volatile("" : "+v"(b0));
      double2 *const dst = l2 + b0;
#pragma unroll
      for (int s = 0; s < 16; ++s) dst[pos[s] - pos[0]] = make_double2(v[s].x, v[s].y);
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        double2 q = src[s * RS];
        lo[s] = {q.x, q.y};
      }
    }
    __syncthreads();
    if (active && !lower) { // (pos >= H for these threads)
      int b1 = lds_phys_u<PADR>(unsigned(pos[0] - H));
      asm volatile("" : "+v"(b1));
      double2 *const dst = l2 + b1;
#pragma unroll
      for (int s = 0; s < 16; ++s) dst[pos[s] - pos[0]] = make_double2(v[s].x, v[s].y);
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        double2 q = src[s * RS];
        v[s + 8] = {q.x, q.y};
        v[s] = lo[s];
      }
    }
    __syncthreads();
  } else if (MODE == 2) {
    // a thread's 16 values may all belong to the second round, so the first round's reads need their own
    // registers until the second round's writes are out
    constexpr int H = 8 * T; // M / 2
    double2 *l2 = reinterpret_cast<double2 *>(lds);
    c64 lo[8];
    if (active) {
#pragma unroll
      for (int s = 0; s < 16; ++s)
        if (pos[s] < H) l2[lds_phys<PADR>(pos[s])] = make_double2(v[s].x, v[s].y);
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        double2 q = l2[lds_phys<PADR>(tid + s * T)];
        lo[s] = {q.x, q.y};
      }
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int s = 0; s < 16; ++s)
        if (pos[s] >= H) l2[lds_phys<PADR>(pos[s] - H)] = make_double2(v[s].x, v[s].y);
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        double2 q = l2[lds_phys<PADR>(tid + s * T)];
        v[s + 8] = {q.x, q.y};
        v[s] = lo[s];
      }
    }
    __syncthreads();
  } else if (MODE == 0) {
    double2 *l2 = reinterpret_cast<double2 *>(lds);
    if (active) {
#pragma unroll
      for (int s = 0; s < 16; ++s) l2[lds_phys<PADR>(pos[s])] = make_double2(v[s].x, v[s].y);
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        double2 q = l2[lds_phys<PADR>(tid + s * T)];
        v[s] = {q.x, q.y};
      }
    }
    __syncthreads();
  } else {
    if (active) {
#pragma unroll
      for (int s = 0; s < 16; ++s) lds[pos[s]] = v[s].x;
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int s = 0; s < 16; ++s) v[s].x = lds[tid + s * T];
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int s = 0; s < 16; ++s) lds[pos[s]] = v[s].y;
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int s = 0; s < 16; ++s) v[s].y = lds[tid + s * T];
    }
    __syncthreads();
  }
}

#undef __syncthreads
// One pass.  PF > 0 (twiddle prefetch): this pass's twiddles were loaded into `wcur` ahead of the previous
// exchange, and the first PF twiddles of the NEXT pass (table `twn`, butterfly stride NSN) are loaded into `wnext`
// before this pass's exchange, so their L2 round trip runs behind the LDS round trip instead of after it.
// TWGEN: only the twiddles w^1, w^2, w^4, w^8 of a butterfly are loaded; the other eleven are products of two of them
// (w^3 = w^1 w^2, w^5 = w^4 w^1, w^6 = w^4 w^2, w^7 = w^4 w^3, w^(8+m) = w^8 w^m): 11 complex multiplications instead of 11
// 16-byte loads per thread and pass.  The vector-memory path, not the fp64 pipe, is what the headline kernel runs out of
// (measured: dropping those loads -6.4 %, halving the kernel's MFMAs -3 %); a generated twiddle is off by 2-4e-16.
template <int LOG2M, int R, int NS, int DIR, int MODE, bool LAST, int PF = 0, int NSN = 1, bool TWGEN = false>
__device__ __forceinline__ void fft_pass(c64 (&v)[16], int tid, bool active, const double2 *__restrict__ tw, double *lds,
                                         const double2 (&wcur)[15], double2 (&wnext)[15], const double2 *__restrict__ twn)
{
  constexpr int T = (1 << LOG2M) / 16, NB = 16 / R;
  if (active) {
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      c64 b[R];
#pragma unroll
      for (int r = 0; r < R; ++r) b[r] = v[t + NB * r];
      if (NS > 1 && TWGEN && R == 16) {
        const int k = RSMP_EXP_TWK0 ? 0 : (tid + t * T) & (NS - 1);
        c64 w[16];
#pragma unroll
        for (int r = 1; r < 16; r <<= 1) {
          if (RSMP_TWGEN_SQ && r > 1) { // experiment: w^2, w^4, w^8 by squaring (one load per butterfly)
            w[r] = cmul(w[r >> 1], w[r >> 1]);
            continue;
          }
          const double2 q = (PF > 0 && NB == 1) ? wcur[r - 1] : tw[(r - 1) * NS + k];
          w[r] = {q.x, q.y};
        }
        // every twiddle is applied as soon as it exists, so that at most w^1..w^8 are live together (32 registers, not 60:
        // this is what removes most of the spills of the 128-VGPR kernels)
        auto apply = [&](int r, c64 wr) { b[r] = DIR > 0 ? cmul(b[r], wr) : cmulc(b[r], wr); };
        apply(1, w[1]);
        apply(2, w[2]);
        apply(4, w[4]);
        w[3] = cmul(w[1], w[2]);
        apply(3, w[3]);
        w[5] = cmul(w[4], w[1]);
        apply(5, w[5]);
        w[6] = cmul(w[4], w[2]);
        apply(6, w[6]);
        w[7] = cmul(w[4], w[3]);
        apply(7, w[7]);
        apply(8, w[8]);
#pragma unroll
        for (int m = 1; m < 8; ++m) apply(8 + m, cmul(w[8], w[m]));
      } else if (NS > 1) {
        const int k = RSMP_EXP_TWK0 ? 0 : (tid + t * T) & (NS - 1);
#pragma unroll
        for (int r = 1; r < R; ++r) {
          // RSMP_EXP_TWLOAD (timing experiment only, WRONG results): load the twiddles of r = 1, 2, 4, 8 only
          const int rr = (RSMP_EXP_TWLOAD && (r & (r - 1))) ? 1 : r;
          const double2 w = (PF > 0 && rr - 1 < PF && NB == 1) ? wcur[rr - 1] : tw[(rr - 1) * NS + k];
          b[r] = DIR > 0 ? cmul(b[r], c64{w.x, w.y}) : cmulc(b[r], c64{w.x, w.y});
        }
      }
      Bfly<R, DIR>::run(b);
#pragma unroll
      for (int r = 0; r < R; ++r) v[t + NB * r] = b[r];
    }
  }
  if (!LAST) {
    if (PF > 0 && active) {
      const int kn = RSMP_EXP_TWK0 ? 0 : tid & (NSN - 1);
#pragma unroll
      for (int r = 1; r <= (TWGEN ? 8 : PF); ++r)
        if (!((RSMP_EXP_TWLOAD || TWGEN) && (r & (r - 1))) && !(TWGEN && RSMP_TWGEN_SQ && r > 1)) wnext[r - 1] = twn[(r - 1) * NSN + kn];
    }
    int pos[16];
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const int j = tid + t * T, k = j & (NS - 1);
#pragma unroll
      for (int r = 0; r < R; ++r) pos[t + NB * r] = (j - k) * R + k + r * NS;
    }
    lds_exchange<T, MODE, (NS == 1 && MODE != 1) ? R : 1, !(RSMP_EXP_NOBAR && NS > 1), (MODE == 2 && R == 16 && 2 * NS <= T)>(v, pos, tid, active, lds);
  }
}

// Full transform.  `tw` points at this size's table (fft_twiddle_count(LOG2M) entries).
// PF = number of twiddles (of 15) per pass that are prefetched ahead of the preceding exchange (registers: 4 each).
template <int LOG2M, int DIR, int MODE, int PF = 0, bool TWGEN = (RSMP_TWGEN_DEFAULT != 0)>
__device__ __forceinline__ void fft_regs(c64 (&v)[16], int tid, bool active, const double2 *__restrict__ tw, double *lds)
{
  constexpr int R0 = fft_first_radix(LOG2M), NP = fft_num_passes(LOG2M);
  double2 wa[15], wb[15];
  fft_pass<LOG2M, R0, 1, DIR, MODE, NP == 1, NP >= 2 ? PF : 0, R0, TWGEN>(v, tid, active, tw, lds, wa, wa, tw);
  if constexpr (NP >= 2)
    fft_pass<LOG2M, 16, R0, DIR, MODE, NP == 2, NP >= 3 ? PF : 0, R0 * 16, TWGEN>(v, tid, active, tw, lds, wa, wb, tw + 15 * R0);
  if constexpr (NP >= 3)
    fft_pass<LOG2M, 16, R0 * 16, DIR, MODE, NP == 3, NP >= 4 ? PF : 0, R0 * 256, TWGEN>(v, tid, active, tw + 15 * R0, lds, wb, wa,
                                                                                         tw + 15 * R0 * 17);
  if constexpr (NP >= 4) fft_pass<LOG2M, 16, R0 * 256, DIR, MODE, NP == 4, 0, 1, TWGEN>(v, tid, active, tw + 15 * R0 * 17, lds, wa, wb, tw);
}

// ------------------------------------------------------------------------------------------------------------
// 8 points per thread: an M-point transform on T8 = M/8 threads (used where a transform has half the points of
// the workgroup's main one, so that all waves take part: thread `tid` owns x[tid + s*T8], s = 0..7, on entry and
// on exit).  Passes are radix 8 with a final radix 8, 4 or 2 (log2 M = 3a + b); twiddle table layout
// [pass >= 1][r-1][k] with entries exp(+2 pi i r k / (R Ns)), see Engine::twiddles8.
constexpr int fft8_num_passes(int log2m) { return (log2m + 2) / 3; }
constexpr int fft8_last_radix(int log2m) { return (log2m % 3) ? (1 << (log2m % 3)) : 8; }
constexpr int fft8_twiddle_count(int log2m)
{
  int n = 0, ns = 8;
  for (int p = 1; p < fft8_num_passes(log2m); ++p) {
    const int r = p + 1 == fft8_num_passes(log2m) ? fft8_last_radix(log2m) : 8;
    n += (r - 1) * ns;
    ns *= 8;
  }
  return n;
}
constexpr int fft8_lds_doubles(int log2m) { return 2 * ((1 << log2m) + (1 << log2m) / 8); } // first exchange padded

template <int LOG2M, int R, int NS, int DIR, bool LAST, bool TWGEN = false>
__device__ __forceinline__ void fft8_pass(c64 (&u)[8], int tid, const double2 *__restrict__ tw, double *lds, bool active = true)
{
  constexpr int T8 = (1 << LOG2M) / 8, NB = 8 / R;
  if (active) {
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    c64 b[R];
#pragma unroll
    for (int r = 0; r < R; ++r) b[r] = u[t + NB * r];
    if (NS > 1 && TWGEN && R >= 4) { // w^1, w^2 (, w^4) loaded, the rest multiplied up (see fft_pass)
      const int k = RSMP_EXP_TWK0 ? 0 : (tid + t * T8) & (NS - 1);
      c64 w[8];
#pragma unroll
      for (int r = 1; r < R; r <<= 1) {
        if (RSMP_TWGEN_SQ && r > 1) {
          w[r] = cmul(w[r >> 1], w[r >> 1]);
          continue;
        }
        const double2 q = tw[(r - 1) * NS + k];
        w[r] = {q.x, q.y};
      }
      w[3] = cmul(w[1], w[2]);
      if (R == 8) {
        w[5] = cmul(w[4], w[1]);
        w[6] = cmul(w[4], w[2]);
        w[7] = cmul(w[4], w[3]);
      }
#pragma unroll
      for (int r = 1; r < R; ++r) b[r] = DIR > 0 ? cmul(b[r], w[r]) : cmulc(b[r], w[r]);
    } else if (NS > 1) {
      const int k = RSMP_EXP_TWK0 ? 0 : (tid + t * T8) & (NS - 1);
#pragma unroll
      for (int r = 1; r < R; ++r) {
        const int rr = (RSMP_EXP_TWLOAD && (r & (r - 1))) ? 1 : r;
        const double2 w = tw[(rr - 1) * NS + k];
        b[r] = DIR > 0 ? cmul(b[r], c64{w.x, w.y}) : cmulc(b[r], c64{w.x, w.y});
      }
    }
    Bfly<R, DIR>::run(b);
#pragma unroll
    for (int r = 0; r < R; ++r) u[t + NB * r] = b[r];
  }
  }
  if (!LAST) {
    double2 *l2 = reinterpret_cast<double2 *>(lds);
    constexpr int PADR = NS == 1 ? 8 : 1; // the first pass scatters with a lane stride of 8 elements
    if (active) {
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const int j = tid + t * T8, k = j & (NS - 1);
#pragma unroll
        for (int r = 0; r < R; ++r) l2[lds_phys<PADR>((j - k) * R + k + r * NS)] = make_double2(u[t + NB * r].x, u[t + NB * r].y);
      }
    }
    if (!(RSMP_EXP_NOBAR >= 2 && NS > 1)) __syncthreads();
    if (active) { // element tid + s T8 sits at phys(tid) + s (T8 + T8 / pad period): one address, seven immediates
      constexpr int RS = T8 + (PADR >= 8 ? T8 / PADR : 0);
      int rb = lds_phys_u<PADR>(unsigned(tid));
      asm volatile("" : "+v"(rb));
      const double2 *const src = l2 + rb;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const double2 q = src[s * RS];
        u[s] = {q.x, q.y};
      }
    }
    if (!(RSMP_EXP_NOBAR >= 2 && NS > 1)) __syncthreads();
  }
}

// ---- the same transform with its twiddles PRELOADED: every table read of a thread (w^1, w^2 (, w^4) of each twiddled pass, the
// rest multiplied up as in the TWGEN form) is issued by fft8_tw_load in one go, typically next to the block's input loads, so
// the whole transform pays one memory round trip instead of one per pass (hipcc does not hoist a pass's table loads over the
// s_barrier of the exchange in front of it: in fft8_regs each pass exposes a full L2 latency).  Same arithmetic, same bits.
constexpr int fft8_tw_regs(int log2m)
{ // double2 registers per thread: passes 1 .. NP-2 are radix 8 (3 loads), the last one radix RL on 8 / RL butterflies
  const int np = fft8_num_passes(log2m), rl = fft8_last_radix(log2m);
  return 3 * (np - 2) + (8 / rl) * (rl == 8 ? 3 : rl == 4 ? 2 : 1);
}
template <int LOG2M> __device__ __forceinline__ void fft8_tw_load(double2 (&w)[fft8_tw_regs(LOG2M)], int tid, const double2 *__restrict__ tw)
{
  constexpr int NP = fft8_num_passes(LOG2M), RL = fft8_last_radix(LOG2M), T8 = (1 << LOG2M) / 8;
  int o = 0, ns = 8;
  const double2 *t = tw;
#pragma unroll
  for (int p = 1; p < NP; ++p) {
    const int R = p + 1 == NP ? RL : 8, NB = 8 / R;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int k = (tid + b * T8) & (ns - 1);
#pragma unroll
      for (int r = 1; r < R; r <<= 1) w[o++] = t[(r - 1) * ns + k];
    }
    t += (R - 1) * ns;
    ns *= 8;
  }
}
// one twiddled pass on preloaded twiddles `w` (this pass's registers start at w[O])
template <int LOG2M, int R, int NS, int DIR, bool LAST, int O, int NW>
__device__ __forceinline__ void fft8_pass_pre(c64 (&u)[8], int tid, const double2 (&wr)[NW], double *lds)
{
  constexpr int T8 = (1 << LOG2M) / 8, NB = 8 / R, PER = R == 8 ? 3 : R == 4 ? 2 : 1;
#pragma unroll
  for (int t = 0; t < NB; ++t) {
    c64 b[R];
#pragma unroll
    for (int r = 0; r < R; ++r) b[r] = u[t + NB * r];
    c64 w[8];
#pragma unroll
    for (int r = 1, i = 0; r < R; r <<= 1, ++i) w[r] = {wr[O + t * PER + i].x, wr[O + t * PER + i].y};
    if (R >= 4) w[3] = cmul(w[1], w[2]);
    if (R == 8) {
      w[5] = cmul(w[4], w[1]);
      w[6] = cmul(w[4], w[2]);
      w[7] = cmul(w[4], w[3]);
    }
#pragma unroll
    for (int r = 1; r < R; ++r) b[r] = DIR > 0 ? cmul(b[r], w[r]) : cmulc(b[r], w[r]);
    Bfly<R, DIR>::run(b);
#pragma unroll
    for (int r = 0; r < R; ++r) u[t + NB * r] = b[r];
  }
  if (!LAST) {
    double2 *l2 = reinterpret_cast<double2 *>(lds);
#pragma unroll
    for (int t = 0; t < NB; ++t) {
      const int j = tid + t * T8, k = j & (NS - 1);
#pragma unroll
      for (int r = 0; r < R; ++r) l2[(j - k) * R + k + r * NS] = make_double2(u[t + NB * r].x, u[t + NB * r].y);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const double2 q = l2[tid + s * T8];
      u[s] = {q.x, q.y};
    }
    __syncthreads();
  }
}
// all threads of the workgroup hold points (M / 8 threads); `between(p)` runs after pass p and its exchange (p = 0 .. NP-2):
// the caller's hook for issuing further loads into registers that only just became free
template <int LOG2M, int DIR, typename Hook>
__device__ __forceinline__ void fft8_regs_pre(c64 (&u)[8], int tid, const double2 (&w)[fft8_tw_regs(LOG2M)], double *lds, Hook between)
{
  constexpr int NP = fft8_num_passes(LOG2M), RL = fft8_last_radix(LOG2M), NW = fft8_tw_regs(LOG2M);
  static_assert(NP >= 3 && NP <= 4, "fft8_regs_pre: 512 <= M <= 4096");
  fft8_pass<LOG2M, 8, 1, DIR, false, true>(u, tid, nullptr, lds, true); // twiddle-free first pass (padded exchange)
  between(0);
  if constexpr (NP == 3) {
    fft8_pass_pre<LOG2M, 8, 8, DIR, false, 0, NW>(u, tid, w, lds);
    between(1);
    fft8_pass_pre<LOG2M, RL, 64, DIR, true, 3, NW>(u, tid, w, lds);
  } else {
    fft8_pass_pre<LOG2M, 8, 8, DIR, false, 0, NW>(u, tid, w, lds);
    between(1);
    fft8_pass_pre<LOG2M, 8, 64, DIR, false, 3, NW>(u, tid, w, lds);
    between(2);
    fft8_pass_pre<LOG2M, RL, 512, DIR, true, 6, NW>(u, tid, w, lds);
  }
}

// `active`: threads that hold points (tid < M/8); every thread of the workgroup must call (barriers)
template <int LOG2M, int DIR, bool TWGEN = (RSMP_TWGEN_DEFAULT != 0)>
__device__ __forceinline__ void fft8_regs_masked(c64 (&u)[8], int tid, bool active, const double2 *__restrict__ tw, double *lds)
{
  constexpr int NP = fft8_num_passes(LOG2M), RL = fft8_last_radix(LOG2M);
  static_assert(NP >= 2 && NP <= 5, "fft8_regs: 64 <= M <= 8192");
  fft8_pass<LOG2M, 8, 1, DIR, false, TWGEN>(u, tid, tw, lds, active);
  if constexpr (NP == 2) fft8_pass<LOG2M, RL, 8, DIR, true, TWGEN>(u, tid, tw, lds, active);
  if constexpr (NP >= 3) fft8_pass<LOG2M, 8, 8, DIR, false, TWGEN>(u, tid, tw, lds, active);
  if constexpr (NP == 3) fft8_pass<LOG2M, RL, 64, DIR, true, TWGEN>(u, tid, tw + 7 * 8, lds, active);
  if constexpr (NP >= 4) fft8_pass<LOG2M, 8, 64, DIR, false, TWGEN>(u, tid, tw + 7 * 8, lds, active);
  if constexpr (NP == 4) fft8_pass<LOG2M, RL, 512, DIR, true, TWGEN>(u, tid, tw + 7 * 8 + 7 * 64, lds, active);
  if constexpr (NP == 5) {
    fft8_pass<LOG2M, 8, 512, DIR, false, TWGEN>(u, tid, tw + 7 * 8 + 7 * 64, lds, active);
    fft8_pass<LOG2M, RL, 4096, DIR, true, TWGEN>(u, tid, tw + 7 * 8 + 7 * 64 + 7 * 512, lds, active);
  }
}
template <int LOG2M, int DIR, bool TWGEN = (RSMP_TWGEN_DEFAULT != 0)>
__device__ __forceinline__ void fft8_regs(c64 (&u)[8], int tid, const double2 *__restrict__ tw, double *lds)
{
  fft8_regs_masked<LOG2M, DIR, TWGEN>(u, tid, true, tw, lds);
}

} // namespace rsmp
