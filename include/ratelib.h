/* ratelib.h -- drop-in C ABI of the MI355X SoX-rate engine.
 *
 * This header declares exactly the interface that the reference plugin binds
 * (/root/reference/rate/ratelib.h:25-81, called from chain.h:36-40 and main.c:25), so that
 * foo_dsp_rate.cpp / chain.h compile and link against libratelib_amd.so unchanged.  Names, argument
 * order, units and return codes are the reference's; everything behind them is new code running on
 * the GPU.  Device-pointer and multi-stream entry points are in ratelib_amd.h.
 *
 * Units: isamp / osamp / ogen / iused count FRAMES (one sample per channel).  Buffers hold
 * interleaved float32, nchannels samples per frame.  No pointer is retained across calls.
 */
#ifndef RATELIB_H
#define RATELIB_H

#include <stddef.h>

/* return codes of every RR_* function; replaces rate/ratelib.h:25-34 (same values) */
enum RR_error {
    RR_OK         = 0, /* success                                                          */
    RR_ENOMEM     = 1, /* host or device allocation failed (the alloc handler ran first)   */
    RR_INTERNAL   = 2, /* a HIP call failed                                                */
    RR_NULLHANDLE = 3, /* handle argument was NULL                                         */
    RR_RATEERROR  = 4, /* kept for ABI compatibility; not produced                         */
    RR_EXTUNINIT  = 5, /* init_ratelib() has not succeeded, or no HIP device is present    */
    RR_INVPARAM   = 6  /* bad argument, or a rate ratio outside [1/5644.8, 5644.8]         */
};

/* replaces rate/ratelib.h:36-42.  Both qualities run the fp64 chain on the GPU; RR_norm only
 * changes the filter specification (20-bit accuracy, small roll-off) as in rate_base.h:692-695. */
enum RR_quality { RR_best = 0, RR_norm = 1 };

/* replaces rate/ratelib.h:44-49: filter phase response, any value in [0,100] */
enum RR_phase { RR_minimum = 0, RR_linear = 50, RR_maximum = 100 };

typedef float fb_sample_t; /* rate/ratelib.h:51; foobar2000's audio_sample */

/* replaces rate/ratelib.h:53-63, field for field */
typedef struct RR_config_tag {
    size_t in_rate, out_rate;  /* Hz                                                              */
    double phase;              /* 0 = minimum ... 50 = linear ... 100 = maximum                   */
    double bandwidth;          /* -3 dB pass-band end, percent of Nyquist (90 ... 99)             */
    int allow_aliasing;        /* non-zero: let the transition band alias above `bandwidth`       */
    enum RR_quality quality;   /* RR_best / RR_norm                                               */
} RR_config;

typedef struct RR_handle_tag RR_handle; /* opaque; rate/ratelib.h:65 */

#ifdef __cplusplus
extern "C" {
#endif

/* Replaces rate/ratelib.h:72 (rate_uni.c:210-223).  Call once before RR_open; not thread-safe.
 * `alloc_error_handler` is invoked (from host C++ frames, never from a HIP callback) when an
 * allocation fails; the plugin's handler throws std::bad_alloc.  Returns 0, or -1 when the handler
 * is NULL or no usable HIP device exists. */
int init_ratelib(void (*alloc_error_handler)(void));

/* Replaces rate/ratelib.h:74 (rate_uni.c:27-57).  Designs the filters on the host, uploads them and
 * allocates the device fifos.  Unlike the reference, an unsupported ratio returns RR_INVPARAM
 * instead of a half-built handle (rate_base.h:738 ignores the failure). */
int RR_open(const RR_config *config, int nchannels,
            RR_handle **const handle);

/* Replaces rate/ratelib.h:75 (rate_base.h:571-614): deliver up to osamp ready frames, take isamp
 * input frames, deliver again.  iused / ogen may be NULL. */
int RR_flow(RR_handle *h, const fb_sample_t *ibuf, fb_sample_t *obuf,
            size_t isamp, size_t osamp, size_t *iused, size_t *ogen);

/* Replaces rate/ratelib.h:76 (rate_base.h:616-636).  ibuf == NULL or isamp == 0 is a no-op; more
 * than isamp_max = 1048576 * min(1, in/out) frames are silently truncated, as in the reference.
 * Everything computable from the pushed input is pullable as soon as this returns. */
int RR_push(RR_handle *h, const fb_sample_t *ibuf, size_t isamp);

/* Replaces rate/ratelib.h:77 (rate_base.h:638-660).  Copies min(osamp, available) frames to obuf and
 * stores that count in *ogen (ogen may be NULL; obuf == NULL or osamp == 0 yields 0). */
int RR_pull(RR_handle *h, fb_sample_t *obuf, size_t osamp, size_t *ogen);

/* Replaces rate/ratelib.h:78 (rate_base.h:454-468,662-672): end of stream.  Afterwards pulling until
 * 0 yields exactly round(total_in * out_rate / in_rate) frames in total. */
int RR_drain(RR_handle *h);

/* Replaces rate/ratelib.h:79 (rate_uni.c:83-90): frees everything, sets *h = NULL, tolerates NULL. */
void RR_close(RR_handle **h);

/* Replaces rate/ratelib.h:81 (rate_uni.c:92-111): static string for an RR_error value. */
const char* RR_strerror(int error);

/* Defined by the reference but not declared in its header (rate_uni.c:225). */
void close_ratelib(void);

#ifdef __cplusplus
}
#endif

#endif
