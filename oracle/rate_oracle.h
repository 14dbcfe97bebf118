/* rate_oracle.h -- CPU restatement of the reference's "Best"/"Normal" double-precision
 * SoX-rate path (rate/rate_base.h + rate/dft_filter.h + rate/rate_filters_generic.h +
 * rate/effects_i_dsp.c + rate/prepare_coefs.h of VSF1/foo_dsp_resampler).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (foo_dsp_resampler_amd/, the C-ABI library)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported baseline.
 *
 * PARITY UNPINNED at the sample level: the reference is MSVC-only code (rate/sox_i.h:19 needs
 * <intrin.h>, rate/xmalloc.c:60 / fft-double/fft4g_dbl.c:100 need the MSVC CRT _aligned_*), so it
 * cannot be compiled in this image without writing stand-in headers, which this build's rules
 * forbid; the reference ships no tests, fixtures or golden vectors either.
 * What pins this restatement is the set of reference-run facts the survey recorded
 * (SURVEY.md sections 0, 8a', 8c [probe]): stage plans, design-call arguments, tap counts, DFT
 * sizes, preloads, output frame counts, block accounting, pullable-frame counts and impulse
 * position.  Those live in tests/golden/survey_probe_facts.json and are checked by
 * tests/test_oracle_facts.py.
 */
#ifndef RATE_ORACLE_H
#define RATE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors RR_config (rate/ratelib.h:53-63); quality 0 = RR_best, 1 = RR_norm */
typedef struct {
  size_t in_rate, out_rate;
  double phase, bandwidth;
  int allow_aliasing;
  int quality;
} orc_config;

/* mirrors enum RR_error (rate/ratelib.h:25-34) */
enum { ORC_OK = 0, ORC_ENOMEM, ORC_INTERNAL, ORC_NULLHANDLE, ORC_RATEERROR, ORC_EXTUNINIT, ORC_INVPARAM };

typedef struct orc_handle orc_handle;

/* rate/rate_base.h:517-543,725-741 (RR_init_x / RR_ctor_double). Unlike the reference ctor a bad
 * factor is reported (ORC_INVPARAM) instead of returning a half-built handle. */
int orc_open(const orc_config *cfg, int nchannels, orc_handle **out);
int orc_push(orc_handle *h, const float *ibuf, size_t isamp);                 /* rate_base.h:616-636 */
int orc_pull(orc_handle *h, float *obuf, size_t osamp, size_t *ogen);         /* rate_base.h:638-660 */
int orc_flow(orc_handle *h, const float *ibuf, float *obuf, size_t isamp, size_t osamp,
             size_t *iused, size_t *ogen);                                     /* rate_base.h:571-614 */
int orc_drain(orc_handle *h);                                                  /* rate_base.h:662-672 */
void orc_close(orc_handle **h);                                                /* rate_base.h:545-557 */
size_t orc_isamp_max(const orc_handle *h);                                     /* rate_base.h:531 */

/* ---- introspection for tests ---- */
enum { ORC_STAGE_HALF = 0, ORC_STAGE_DFT = 1, ORC_STAGE_POLY = 2 };

typedef struct {
  int kind;
  int L;                 /* dft: up factor; poly: number of phases when rational (arbL) */
  int step_int;          /* dft: +M (time-domain decimation) or -m (F-domain /2^m); poly: integer part of step */
  int64_t at, step;      /* poly: 32.32 fixed-point clock start / increment */
  int n;                 /* poly: taps per phase; half-band: number of distinct coefs */
  int interp_order;      /* poly: 0..3 */
  int phase_bits;        /* poly, interpolated variants */
  int pre, pre_post, preload;
  int remL;              /* dft: initial zero-stuffing offset */
  int num_taps, dft_length, post_peak; /* dft */
  double out_in_ratio;
} orc_stage_info;

int orc_num_stages(const orc_handle *h);
int orc_stage_info_get(const orc_handle *h, int idx, orc_stage_info *info);
/* Time-domain taps of dft filter `which` (0 = pre, 1 = post) after any phase conversion,
 * BEFORE the 2L/N scaling; and the packed spectrum actually used (Ooura packing). */
const double *orc_dft_taps(const orc_handle *h, int which, int *len);
const double *orc_dft_spectrum(const orc_handle *h, int which, int *len);
/* polyphase table [phase][tap][order+1] (rate/prepare_coefs.h:20-46) */
const double *orc_poly_table(const orc_handle *h, int *len);
/* current contents of the input fifo of stage `stage` (stage == num_stages: output fifo) */
const double *orc_stage_fifo(const orc_handle *h, int channel, int stage, int *len);

/* every lsx_design_lpf-equivalent call made while opening (SURVEY.md 8a') */
typedef struct { double Fp, Fs, Fn, att; int k; int num_taps; double beta; } orc_design_call;
int orc_design_trace(const orc_handle *h, orc_design_call *calls, int max_calls);

/* ---- design primitives (rate/effects_i_dsp.c) ---- */
double orc_bessel_I0(double x);                                   /* :46-55 */
int orc_dft_length(int num_taps);                                 /* :64-73 */
double orc_kaiser_beta(double att, double tr_bw);                 /* :83-108 */
double *orc_design_lpf(double Fp, double Fs, double Fn, double att, int *num_taps, int k,
                       double beta);                              /* :137-171 (malloc'd; caller frees with orc_free) */
void orc_fir_to_phase(double **h, int *len, int *post_len, double phase); /* :181-278, long double inside (see rate_oracle.c) */
void orc_fir_to_phase_ref64(double **h, int *len, int *post_len, double phase); /* :181-278 in the reference's own fp64 */
void orc_set_phase_arith(int ref64); /* 1: orc_fir_to_phase runs the fp64 statement (process-wide; default 0) */
void orc_free(void *p);

/* real FFT with the reference's packing/scaling conventions (fft-double/fft4g_dbl.c:26-62):
 * isgn=+1: a[0]=X0, a[1]=X[n/2], a[2k]=Re X[k], a[2k+1]=Im X[k], X[k]=sum x[j] e^{+2 pi i jk/n};
 * isgn=-1: inverse, unnormalised (returns n/2 times the signal). n = power of two >= 4. */
void orc_rdft(int n, int isgn, double *a);

/* ---- caller-side harness (plugin_harness.c): the plugin's chunk logic over a table of ABI entry points ---- */
typedef struct {
  int (*open)(const orc_config *cfg, int nchannels, void **handle);  /* RR_open, rate/ratelib.h:74 (RR_config has orc_config's layout) */
  int (*push)(void *handle, const float *ibuf, size_t isamp);        /* RR_push, :76 */
  int (*pull)(void *handle, float *obuf, size_t osamp, size_t *ogen); /* RR_pull, :77 */
  int (*drain)(void *handle);                                         /* RR_drain, :78 */
  void (*close)(void **handle);                                       /* RR_close, :79 */
} orc_rr_api;

typedef struct orc_dsp orc_dsp;
/* api == NULL: the CPU oracle's orc_* functions */
orc_dsp *orc_dsp_create_on(const orc_rr_api *api, int out_rate, int quality, int allow_aliasing, int passband10, int phase);
orc_dsp *orc_dsp_create(int out_rate, int quality, int allow_aliasing, int passband10, int phase);
void orc_dsp_destroy(orc_dsp *d);
int orc_dsp_on_chunk(orc_dsp *d, const float *cur, size_t count, unsigned ch, unsigned rate, unsigned chmask); /* 1 = pass-through */
void orc_dsp_end_of_track(orc_dsp *d);
void orc_dsp_flush(orc_dsp *d);
double orc_dsp_latency(const orc_dsp *d);
int orc_dsp_last_error(const orc_dsp *d);
void orc_lpc_extrapolate(float *data, size_t data_len, int nch, int order, size_t extra_bkwd, size_t extra_fwd);

#ifdef __cplusplus
}
#endif
#endif
