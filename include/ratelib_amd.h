/* ratelib_amd.h -- additive extensions of the ratelib.h C ABI for GPU-resident data and for many
 * independent streams per handle.  Nothing here exists in the reference; the closest reference
 * interfaces are cited so a maintainer can see what each call generalises.
 *
 * A "batch" handle holds `nstreams` independent streams of `nchannels` channels each, all with the
 * same RR_config and all pushed in lock step (the reference would use nstreams separate handles:
 * rate/rate_base.h:533-540 makes every channel an independent rate_t already).  Buffers are laid out
 * [stream][frame][channel]; `*_stride` is the distance between consecutive streams in FRAMES.
 */
#ifndef RATELIB_AMD_H
#define RATELIB_AMD_H

#include "ratelib.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Generalises RR_open (rate/ratelib.h:74) to nstreams lock-stepped streams. RR_push/RR_pull/RR_flow
 * on such a handle use packed host buffers (stride = the frame count of the call).
 * Every stream's output is bit-identical to what a handle of its own produces, for even and odd channel counts
 * alike: the channel pairs that share a complex transform never straddle two streams (with an odd count the last
 * channel of each stream rides alone, exactly as in a one-stream handle). */
int RRX_open_batch(const RR_config *config, int nchannels, int nstreams, RR_handle **const handle);

/* Devices.  A handle lives on ONE HIP device: all of its memory, streams and launches.  Every entry point of both headers
 * selects that device for the duration of the call and restores the caller's afterwards, so a handle may be used from any
 * thread whatever device that thread has current (one thread at a time per handle, as in the reference: SURVEY 8b).
 *   RR_open / RRX_open_batch place the handle on the calling thread's current device, or -- when the environment variable
 *   RATELIB_AMD_DEVICES ("all", or a comma list of device indices) was set at init_ratelib -- deal new handles round-robin
 *   over those devices: the unchanged plugin, one process with one handle per converter thread (chain.h:36), then spreads
 *   over the GPUs of a node by itself (channels / streams are independent, rate/rate_base.h:533-540: no collective).
 *   RRX_open_batch_on names the device explicitly: RR_INVPARAM for an index the process does not have (or < 0),
 *   RR_EXTUNINIT for a device that is not gfx950.
 * Device pointers handed to RRX_*_device must be accessible from the handle's device; RRX_device reports it (-1: NULL). */
int RRX_open_batch_on(const RR_config *config, int nchannels, int nstreams, int device, RR_handle **const handle);
int RRX_device(const RR_handle *h);

/* Device-pointer forms of RR_push / RR_pull / RR_flow (rate/ratelib.h:75-77).  Pointers are HBM
 * addresses valid on the handle's HIP stream; calls only enqueue work (no host synchronisation),
 * the frame counts they return are exact because availability never depends on sample values.
 * RRX_flow_device consumes the input in place and writes new output straight into d_obuf.
 * Ordering is the caller's: whatever filled d_ibuf (or last wrote d_obuf, e.g. a zero fill) on ANOTHER stream must have
 * completed, or be ordered before the handle's stream by an event, when the call is made; work on the handle's own
 * stream (RRX_set_stream) is ordered by the stream itself. */
int RRX_push_device(RR_handle *h, const fb_sample_t *d_ibuf, size_t in_stride, size_t isamp);
int RRX_pull_device(RR_handle *h, fb_sample_t *d_obuf, size_t out_stride, size_t osamp, size_t *ogen);
int RRX_flow_device(RR_handle *h, const fb_sample_t *d_ibuf, size_t in_stride, fb_sample_t *d_obuf, size_t out_stride,
                    size_t isamp, size_t osamp, size_t *iused, size_t *ogen);

/* Host-pointer forms with an explicit stream stride (batch handles). */
int RRX_push_strided(RR_handle *h, const fb_sample_t *ibuf, size_t in_stride, size_t isamp);
int RRX_pull_strided(RR_handle *h, fb_sample_t *obuf, size_t out_stride, size_t osamp, size_t *ogen);

/* Use the caller's hipStream_t (passed as void*) for all work of this handle from now on.  NULL is the device's default
 * stream, as in every HIP call (it is also what PyTorch's default stream is); RRX_STREAM_OWN restores the stream the
 * handle created for itself at RR_open, which is what a handle uses until this is called.  The caller keeps ownership of
 * its stream: RR_close never destroys it, but waits on it, so a caller-owned stream must outlive RR_close.  Work already
 * queued on the previous stream is ordered before the work queued after the switch.  RRX_sync blocks until everything
 * enqueued so far has finished. */
#define RRX_STREAM_OWN ((void *)(~(size_t)0))
int RRX_set_stream(RR_handle *h, void *hip_stream);
int RRX_sync(RR_handle *h);

/* Per-kernel timing for benchmarks: while enabled, every stage launch is bracketed by HIP events on the
 * handle's stream.  RRX_profile_read synchronises, returns the summed duration and launch count of the
 * chain's dominant kernel ("hot": the fused dft->polyphase kernel, or the dft stage) and of all other
 * stage kernels, and clears the records. */
int RRX_profile(RR_handle *h, int enable);
int RRX_profile_read(RR_handle *h, double *hot_ms, long long *hot_launches, double *other_ms, long long *other_launches);
/* The same records per kernel instance, as JSON text: [{"kernel": "rsmp::fused_kernel<12, 11, 2, 7, true>", "hot": 1,
 * "launches": n, "ms": t}, ...] with the names rocprofv3 prints, so a benchmark line can say which variant the
 * engine's dispatch picked.  Synchronises and clears the records.  Returns the length written (text is truncated
 * to cap-1 bytes) or the negated RR_error. */
int RRX_profile_report(RR_handle *h, char *buf, size_t cap);

/* Test hook (fault injection), inert unless the process was started with RSMP_TEST_HOOKS set in its environment (read once,
 * at init_ratelib): the nth device allocation from now on, counted process-wide, fails as if the GPU were
 * out of memory (RR_ENOMEM + the init_ratelib handler, rate/xmalloc.c:38-43); 0 disarms.  A failure in the middle of a
 * push or drain poisons the handle: every later data call returns RR_INTERNAL until it is closed (its counters no
 * longer describe the device fifos; the reference has no recovery path either, chain.h:26-29 tears the chain down). */
void RRX_debug_fail_alloc(int nth);

/* Introspection: isamp_max of rate_base.h:531, frames currently pullable (fifo_occupancy of the last
 * fifo, rate_base.h:447-448), shape of the handle. */
size_t RRX_isamp_max(const RR_handle *h);
size_t RRX_available(const RR_handle *h);
int RRX_channels(const RR_handle *h);
int RRX_streams(const RR_handle *h);

/* Host-only (no GPU needed): JSON description of the stage chain the planner builds for `config`
 * (what rate_init decides, rate/rate_base.h:247-423).  Returns the length written (without the
 * terminator), or the negated RR_error on failure; the text is truncated to cap-1 bytes. */
int RRX_describe_plan(const RR_config *config, char *buf, size_t cap);

/* Host-only (no GPU needed): which kernel form the first stage pair of `config` gets on handles of `nchannels` channels per
 * stream, as JSON: {"sub_blocked": false} or {"sub_blocked": true, "two_round": .., "nsub": .., "Vs": .., "V": .., "taps": ..,
 * "N": .., "Pref": .., "sub_blocks": [{"off", "len", "win", "shift"} ...]} -- the geometry of the sub-blocked fused kernels
 * (DESIGN.md 4), decided from the plan alone.  Nothing in the reference corresponds to it (its blocks are one transform
 * each, rate/dft_filter.h:60-190); it exists so that the decision can be tested without a device.  Returns as
 * RRX_describe_plan does. */
int RRX_describe_dispatch(const RR_config *config, int nchannels, char *buf, size_t cap);

/* Host-only: copy a designed table for `config` into out[0..cap): which = 0 / 1 -> taps of the first /
 * second DFT-stage filter (after phase conversion, before the 2L/N scaling of rate_base.h:175),
 * which = 2 -> polyphase table [phase][tap][order+1] (rate/prepare_coefs.h:20-46).  *count receives
 * the full length.  Returns RR_OK or RR_INVPARAM. */
int RRX_plan_table(const RR_config *config, int which, double *out, size_t cap, size_t *count);

#ifdef __cplusplus
}
#endif

#endif
