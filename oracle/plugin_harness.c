/* plugin_harness.c -- the caller side of the ratelib.h ABI as ONE test harness: the chunk logic of
 * dsp_rate::on_chunk / flushwrite / get_latency (foo_dsp_rate.cpp:84-322) with the LPC edge extrapolation of
 * lpc/lpc.cpp, written once over a table of {open, push, pull, drain, close} function pointers (orc_rr_api).
 * The tests run the SAME harness over the CPU oracle (orc_*, the default table) and over the product's
 * RR_* entry points (libratelib_amd.so, table filled in by tests/oracle_binding.py), so what is compared is the
 * resampler behind the ABI under the plugin's real call pattern (small chunks, pre-roll drop, three flush
 * cases), not two copies of the chunk logic.
 * TEST INFRASTRUCTURE ONLY, like rate_oracle.c; nothing of it is linked into the product library.
 * The chunk logic itself is "parity unpinned" (foo_dsp_rate.cpp needs the foobar2000 SDK, the reference ships
 * no fixture for it); the LPC arithmetic IS pinned on the reference's own lpc/lpc.cpp
 * (tests/golden/lpc_reference_vectors.npz, tests/test_lpc_reference.py).
 */
#include "rate_oracle.h"

#include <stdlib.h>
#include <string.h>

#define LPC_ORDER 32 /* lpc/lpc.h:25 */

/* ------------------------------------------------------------------ lpc/lpc.cpp */
static void lpc_window(float *x, size_t n) /* :85-93 Welch */
{
  const float n2 = (n + 1) / 2.0f;
  size_t i;
  for (i = 0; i < n; ++i) {
    float k = ((int)i + 1 - n2) / n2;
    x[i] *= 1.0f - k * k;
  }
}

static void lpc_autocorr(const float *x, size_t n, double *r, int m) /* :96-110 */
{
  int j;
  for (j = m; j >= 0; --j) {
    double d = 0;
    size_t i;
    for (i = (size_t)j; i < n; ++i) d += (double)x[i] * x[i - (size_t)j];
    r[j] = d;
  }
}

static int lpc_solve(const double *r, double *a, int order) /* :112-165 */
{
  int i, j, used = order;
  double err = r[0] * (1. + 1e-10), eps = 1e-9 * r[0] + 1e-10, damp;
  for (i = 0; i < order; ++i) {
    double k;
    if (err < eps) {
      memset(&a[i], 0, (size_t)(order - i) * sizeof(a[0]));
      used = i;
      break;
    }
    k = -r[i + 1];
    for (j = 0; j < i; ++j) k -= a[j] * r[i - j];
    k /= err;
    a[i] = k;
    for (j = 0; j < i / 2; ++j) {
      double t = a[j];
      a[j] += k * a[i - 1 - j];
      a[i - 1 - j] += k * t;
    }
    if (i & 1) a[j] += a[j] * k;
    err *= 1.0 - k * k;
  }
  for (j = 0, damp = 0.999; j < used; ++j, damp *= 0.999) a[j] *= damp;
  if (used == 0) { used = 1; a[0] = -1; }
  return used;
}

/* :25-68 and :167-195; data -> frame 0 of data_len interleaved frames */
static void lpc_extrapolate(float *data, size_t data_len, int nch, int order, size_t bk, size_t fw)
{
  float *line = (float *)malloc(sizeof(float) * (bk + data_len + fw)), *x = line + bk;
  double *r = (double *)malloc(sizeof(double) * ((size_t)order + 1)), *a = (double *)malloc(sizeof(double) * (size_t)order);
  int c;
  for (c = 0; c < nch; ++c) {
    long i;
    int j, used;
    memset(line, 0, sizeof(float) * (bk + data_len + fw));
    for (i = 0; i < (long)data_len; ++i) x[i] = data[i * nch + c];
    lpc_window(x, data_len);
    lpc_autocorr(x, data_len, r, order);
    used = lpc_solve(r, a, order);
    for (i = 0; i < (long)data_len; ++i) x[i] = data[i * nch + c];
    if (fw) {
      float *p = x + data_len - used;
      for (i = 0; i < (long)fw; ++i) {
        float s = 0;
        for (j = 0; j < used; ++j) s -= p[i + j] * (float)a[used - 1 - j];
        if (s > 10.f) s = 10.f; else if (s < -10.f) s = -10.f;
        p[used + i] = s;
      }
      for (i = (long)data_len; i < (long)(data_len + fw); ++i) data[i * nch + c] = x[i];
    }
    if (bk) {
      float *p = x - 1 + used;
      for (i = 0; i < (long)bk; ++i) {
        float s = 0;
        for (j = 0; j < used; ++j) s -= p[-i - j] * (float)a[used - 1 - j];
        if (s > 10.f) s = 10.f; else if (s < -10.f) s = -10.f;
        p[-used - i] = s;
      }
      for (i = -(long)bk; i < 0; ++i) data[i * nch + c] = x[i];
    }
  }
  free(a); free(r); free(line);
}

/* exported for tests: same signature as lpc_extrapolate2 of lpc/lpc.h:25 */
void orc_lpc_extrapolate(float *data, size_t data_len, int nch, int order, size_t extra_bkwd, size_t extra_fwd)
{
  lpc_extrapolate(data, data_len, nch, order, extra_bkwd, extra_fwd);
}

/* ------------------------------------------------------------------ util.h:24-55 */
static unsigned gcd_u(unsigned a, unsigned b)
{
  unsigned c;
  if (!a || !b) return 0;
  c = a % b;
  while (c) { a = b; b = c; c = a % b; }
  return b;
}

static void samples_len(unsigned *r1, unsigned *r2, unsigned N, unsigned M)
{
  unsigned v = gcd_u(*r1, *r2), n, z;
  if (!v) return;
  *r1 /= v; *r2 /= v;
  n = (v + N - 1) / N;
  z = *r1 > *r2 ? *r1 : *r2;
  if (z * n > M) n = M / z;
  if (n < 1) n = 1;
  *r1 *= n; *r2 *= n;
}

static unsigned next_pow2(unsigned n, unsigned p) { while (p < n) p *= 2; return p; }
static unsigned umin(unsigned a, unsigned b) { return a < b ? a : b; }
static unsigned umax(unsigned a, unsigned b) { return a > b ? a : b; }
static size_t zmin(size_t a, size_t b) { return a < b ? a : b; }

/* ------------------------------------------------------------------ foo_dsp_rate.{h,cpp} */
typedef struct {
  float *data;
  size_t frames;
  unsigned ch, rate;
} out_chunk;

/* the five ratelib.h entry points the plugin binds (chain.h:36-40); handles are opaque */
static int api_open(const orc_config *c, int nch, void **h) { return orc_open(c, nch, (orc_handle **)h); }
static int api_push(void *h, const float *i, size_t n) { return orc_push((orc_handle *)h, i, n); }
static int api_pull(void *h, float *o, size_t n, size_t *g) { return orc_pull((orc_handle *)h, o, n, g); }
static int api_drain(void *h) { return orc_drain((orc_handle *)h); }
static void api_close(void **h) { orc_close((orc_handle **)h); }
static const orc_rr_api oracle_api = {api_open, api_push, api_pull, api_drain, api_close};

typedef struct orc_dsp {
  int out_rate_cfg, quality, allow_aliasing, passband10, phase; /* RateConfig, dsp_config.h:71-111 */
  orc_rr_api api;
  int err;   /* first non-zero RR_error of the current entry-point call (chain.h:26-29 raises per call) */
  void *h;
  size_t in_accum, out_accum;
  float *in0, *in, *outb;
  size_t inbuf0, inbuf, outbuf, prime;
  unsigned buf_ch;
  unsigned out_rate, rate, ch, chmask;
  size_t n_add, n_drop, held, dropped;
  int pre;
  out_chunk *out;
  size_t nout, capout;
} orc_dsp;

static unsigned realrate(const orc_dsp *d, unsigned in) /* dsp_config.h:80-95 */
{
  if (d->out_rate_cfg > 0) return (unsigned)d->out_rate_cfg;
  switch (d->out_rate_cfg) {
    case -2: return in * 2;
    case -5: return in * 4;
    case -3: return in / 2;
    case -4: return in / 4;
  }
  return in;
}

static void emit(orc_dsp *d, const float *frames, size_t n)
{
  out_chunk *c;
  if (d->nout == d->capout) {
    d->capout = d->capout ? d->capout * 2 : 16;
    d->out = (out_chunk *)realloc(d->out, d->capout * sizeof(out_chunk));
  }
  c = &d->out[d->nout++];
  c->data = (float *)malloc(sizeof(float) * n * d->ch);
  memcpy(c->data, frames, sizeof(float) * n * d->ch);
  c->frames = n; c->ch = d->ch; c->rate = d->out_rate;
}

static void note(orc_dsp *d, int rc) { if (rc && !d->err) d->err = rc; }

static void dsp_close(orc_dsp *d) /* :123-128 */
{
  d->api.close(&d->h);
  d->in_accum = d->out_accum = 0;
}

static void reinit(orc_dsp *d, unsigned rate, unsigned ch, unsigned chmask) /* :84-121 */
{
  orc_config c;
  unsigned a, b;
  size_t need_in, need_out;
  d->out_rate = realrate(d, rate);
  c.in_rate = rate; c.out_rate = d->out_rate; c.phase = (double)d->phase;
  c.bandwidth = (double)d->passband10 / 10.0; c.allow_aliasing = d->allow_aliasing ? 1 : 0; c.quality = d->quality;
  note(d, d->api.open(&c, (int)ch, &d->h));
  d->ch = ch; d->chmask = chmask; d->rate = rate;
  d->in_accum = d->out_accum = 0;
  d->held = 0; d->dropped = 0; d->pre = 0;
  a = rate; b = d->out_rate;
  samples_len(&a, &b, 20, 8192u);
  d->n_add = a; d->n_drop = b;
  d->inbuf = umin(umax(rate / 10, 2048u), 65536u);
  d->prime = umax(umin(umax(rate / 20, 1024u), 16384u), 2 * LPC_ORDER + 1);
  need_in = d->n_add + d->inbuf + d->n_add;
  need_out = umin(next_pow2(d->out_rate / 10, 8192u), 65536u) + d->n_drop;
  if (need_in < d->inbuf0) need_in = d->inbuf0;
  if (need_out < d->outbuf) need_out = d->outbuf;
  if (d->buf_ch < ch || d->inbuf0 < need_in || d->outbuf < need_out) {
    unsigned bc = d->buf_ch > ch ? d->buf_ch : ch;
    free(d->in0); free(d->outb);
    d->in0 = (float *)calloc(bc * need_in, sizeof(float));
    d->outb = (float *)calloc(bc * need_out, sizeof(float));
    d->inbuf0 = need_in; d->outbuf = need_out; d->buf_ch = bc;
  }
  d->in = d->in0 + d->n_add * d->ch;
}

static void flushwrite(orc_dsp *d) /* :218-313 */
{
  size_t got, avail;
  if (!d->h) return;
  if (!d->pre && !(d->held > 2 * LPC_ORDER)) { /* too short to extrapolate */
    note(d, d->api.push(d->h, d->in, d->held));
    note(d, d->api.drain(d->h));
    for (;;) {
      got = 0; note(d, d->api.pull(d->h, d->outb, d->outbuf, &got));
      if (!got) break;
      d->out_accum += got;
      emit(d, d->outb, got);
    }
    dsp_close(d);
    return;
  }
  if (!d->pre) { /* one short buffer: extrapolate both ways */
    size_t prime = zmin(d->held, d->prime);
    lpc_extrapolate(d->in, prime, (int)d->ch, LPC_ORDER, d->n_add, 0);
    lpc_extrapolate(d->in + (d->held - prime) * d->ch, prime, (int)d->ch, LPC_ORDER, 0, d->n_add);
    d->pre = 1;
    note(d, d->api.push(d->h, d->in0, d->n_add + d->held + d->n_add));
    note(d, d->api.drain(d->h));
    d->dropped = 0;
    d->held = 0;
    for (;;) {
      size_t drop;
      got = 0; note(d, d->api.pull(d->h, d->outb + d->held * d->ch, d->outbuf - d->held, &got));
      if (!got) break;
      drop = zmin(d->n_drop - d->dropped, got);
      if (drop) {
        got -= drop;
        d->dropped += drop;
        memmove(d->outb, d->outb + drop * d->ch, got * d->ch * sizeof(float));
      }
      d->held += got;
      avail = d->held - zmin(d->held, d->n_drop);
      if (avail) {
        d->out_accum += avail;
        emit(d, d->outb, avail);
        d->held -= avail;
        memmove(d->outb, d->outb + avail * d->ch, d->held * d->ch * sizeof(float));
      }
    }
    dsp_close(d);
    return;
  }
  /* steady state reached earlier: extrapolate forward from the retained tail */
  lpc_extrapolate(d->in + (d->inbuf - d->prime) * d->ch, d->prime, (int)d->ch, LPC_ORDER, 0, d->n_add);
  note(d, d->api.push(d->h, d->in + d->inbuf * d->ch, d->n_add));
  note(d, d->api.drain(d->h));
  d->held = 0;
  for (;;) {
    got = 0; note(d, d->api.pull(d->h, d->outb + d->held * d->ch, d->outbuf - d->held, &got));
    if (!got) break;
    d->held += got;
    avail = d->held - zmin(d->held, d->n_drop);
    if (avail) {
      d->out_accum += avail;
      emit(d, d->outb, avail);
      d->held -= avail;
      memmove(d->outb, d->outb + avail * d->ch, d->held * d->ch * sizeof(float));
    }
  }
  dsp_close(d);
}

orc_dsp *orc_dsp_create_on(const orc_rr_api *api, int out_rate, int quality, int allow_aliasing, int passband10, int phase)
{
  orc_dsp *d = (orc_dsp *)calloc(1, sizeof(*d));
  d->api = api ? *api : oracle_api;
  d->out_rate_cfg = out_rate; d->quality = quality; d->allow_aliasing = allow_aliasing;
  d->passband10 = passband10; d->phase = phase;
  return d;
}

static void clear_out(orc_dsp *d)
{
  size_t i;
  for (i = 0; i < d->nout; ++i) free(d->out[i].data);
  d->nout = 0;
}

orc_dsp *orc_dsp_create(int out_rate, int quality, int allow_aliasing, int passband10, int phase)
{
  return orc_dsp_create_on(NULL, out_rate, quality, allow_aliasing, passband10, phase);
}

/* RR_error raised during the last on_chunk / end_of_track call (0 = none) */
int orc_dsp_last_error(const orc_dsp *d) { return d->err; }

void orc_dsp_destroy(orc_dsp *d)
{
  if (!d) return;
  if (d->h) dsp_close(d);
  clear_out(d);
  free(d->out); free(d->in0); free(d->outb); free(d);
}

/* :130-210; returns 1 for pass-through */
int orc_dsp_on_chunk(orc_dsp *d, const float *cur, size_t count, unsigned ch, unsigned rate, unsigned chmask)
{
  size_t got;
  d->err = 0;
  if (!d->h) {
    if ((int)rate == d->out_rate_cfg) return 1;
    reinit(d, rate, ch, chmask);
  } else if (d->ch != ch || d->chmask != chmask || d->rate != rate) {
    flushwrite(d);
    if ((int)rate == d->out_rate_cfg) return 1;
    reinit(d, rate, ch, chmask);
  }
  do {
    size_t drop;
    if (!d->pre) {
      size_t take = zmin(count, d->inbuf - d->held);
      memcpy(d->in + d->held * d->ch, cur, take * d->ch * sizeof(float));
      d->held += take; cur += take * d->ch; count -= take; d->in_accum += take;
      if (d->held == d->inbuf) {
        lpc_extrapolate(d->in, d->prime, (int)d->ch, LPC_ORDER, d->n_add, 0);
        d->pre = 1;
        note(d, d->api.push(d->h, d->in0, d->n_add + d->inbuf));
      }
    }
    if (d->pre && count) {
      if (count < d->inbuf) {
        memmove(d->in, d->in + count * d->ch, (d->inbuf - count) * d->ch * sizeof(float));
        memcpy(d->in + (d->inbuf - count) * d->ch, cur, count * d->ch * sizeof(float));
      } else
        memcpy(d->in, cur + (count - d->inbuf) * d->ch, d->inbuf * d->ch * sizeof(float));
      note(d, d->api.push(d->h, cur, count));
      cur += count * d->ch; d->in_accum += count; count = 0;
    }
    got = 0; note(d, d->api.pull(d->h, d->outb, d->outbuf, &got));
    drop = d->n_drop - d->dropped;
    if (drop) {
      drop = zmin(drop, got);
      got -= drop;
      d->dropped += drop;
    }
    if (got) {
      d->out_accum += got;
      emit(d, d->outb + drop * d->ch, got);
    }
  } while (count || got);
  while (d->in_accum > d->rate && d->out_accum > d->out_rate) {
    d->in_accum -= d->rate;
    d->out_accum -= d->out_rate;
  }
  return 0;
}

void orc_dsp_end_of_track(orc_dsp *d) { d->err = 0; flushwrite(d); }
void orc_dsp_flush(orc_dsp *d) { if (d->h) dsp_close(d); } /* :212-216 */

double orc_dsp_latency(const orc_dsp *d) /* :315-322 */
{
  if (d->rate && d->out_rate) return (double)d->in_accum / (double)d->rate - (double)d->out_accum / (double)d->out_rate;
  return 0;
}

size_t orc_dsp_out_count(const orc_dsp *d) { return d->nout; }
size_t orc_dsp_out_frames(const orc_dsp *d, size_t i) { return d->out[i].frames; }
unsigned orc_dsp_out_channels(const orc_dsp *d, size_t i) { return d->out[i].ch; }
unsigned orc_dsp_out_rate(const orc_dsp *d, size_t i) { return d->out[i].rate; }
const float *orc_dsp_out_data(const orc_dsp *d, size_t i) { return d->out[i].data; }
void orc_dsp_out_clear(orc_dsp *d) { clear_out(d); }
