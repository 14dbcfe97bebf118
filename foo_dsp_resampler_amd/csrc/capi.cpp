// C ABI: the reference's ratelib.h entry points (rate/rate_uni.c:27-111,210-231) plus the
// device / batch extensions of include/ratelib_amd.h, all thin shims over rsmp::Engine.
#include "../../include/ratelib_amd.h"

#include "engine.hpp"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

struct RR_handle_tag {
  rsmp::Engine *eng;
};

namespace {

void (*g_alloc_handler)(void) = nullptr;
int g_initialized = 0;
// RATELIB_AMD_DEVICES (read by init_ratelib): devices RR_open / RRX_open_batch deal new handles over, round-robin.
// Empty = a handle lives on the calling thread's current device.
std::vector<int> g_devices;
std::atomic<unsigned> g_next_device{0};

// "all" or a comma list of device indices; false when it names a device the process does not have or that is not gfx950
bool parse_devices(const char *spec, int count, std::vector<int> &out)
{
  out.clear();
  if (!spec || !*spec) return true;
  if (std::strcmp(spec, "all") == 0) {
    for (int d = 0; d < count; ++d) out.push_back(d);
  } else {
    for (const char *p = spec; *p;) {
      char *end = nullptr;
      const long d = std::strtol(p, &end, 10);
      if (end == p || d < 0 || d >= count) return false;
      out.push_back(int(d));
      p = *end == ',' ? end + 1 : end;
      if (*end && *end != ',') return false;
    }
  }
  for (int d : out)
    if (!rsmp::device_is_gfx950(d)) return false;
  return true;
}

rsmp::Config to_config(const RR_config *c)
{
  rsmp::Config k;
  k.in_rate = c->in_rate;
  k.out_rate = c->out_rate;
  k.phase = c->phase;
  k.bandwidth = c->bandwidth;
  k.allow_aliasing = c->allow_aliasing;
  k.quality = c->quality == RR_best ? 0 : 1;
  return k;
}

// engine codes are numerically the RR_error values; an allocation failure additionally runs the
// registered handler (xmalloc.c:38-43) from this plain host frame, where unwinding is safe
int finish(int rc)
{
  if (rc == RR_ENOMEM && g_alloc_handler) g_alloc_handler();
  return rc;
}

// No C++ exception of ours may cross the C ABI: host-side containers can throw std::bad_alloc, which is
// reported like any other allocation failure (the handler itself may throw, as the plugin's does; that is the
// caller's contract, rate/xmalloc.c:38-43).
template <class Fn> int guarded(RR_handle *h, Fn fn)
{
  int rc;
  rsmp::DeviceScope on(h->eng->device()); // the handle's device, whatever the calling thread has selected
  if (!on.ok()) return finish(RR_INTERNAL);
  try {
    rc = fn();
  } catch (const std::bad_alloc &) {
    rc = RR_ENOMEM;
  } catch (...) {
    rc = RR_INTERNAL;
  }
  return finish(rc);
}

// device: -1 = the default placement (RATELIB_AMD_DEVICES round-robin, else the calling thread's current device)
int open_common(const RR_config *config, int nchannels, int nstreams, int device, RR_handle **const handle)
{
  if (handle == nullptr) return RR_INVPARAM;
  *handle = nullptr;
  if (!g_initialized) return RR_EXTUNINIT;
  if (config == nullptr) return RR_INVPARAM;
  RR_handle *h = new (std::nothrow) RR_handle_tag();
  if (!h) return finish(RR_ENOMEM);
  h->eng = nullptr;
  if (device < 0 && !g_devices.empty()) device = g_devices[g_next_device.fetch_add(1) % g_devices.size()];
  int rc;
  try {
    rc = rsmp::Engine::create(to_config(config), nchannels, nstreams, device, &h->eng);
  } catch (const std::bad_alloc &) {
    rc = RR_ENOMEM;
  } catch (...) {
    rc = RR_INTERNAL;
  }
  if (rc != RR_OK) {
    delete h;
    return finish(rc);
  }
  *handle = h;
  return RR_OK;
}

} // namespace

extern "C" {

int init_ratelib(void (*alloc_error_handler)(void))
{
  g_initialized = 0;
  if (alloc_error_handler == nullptr) return -1;
  g_alloc_handler = alloc_error_handler;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return -1; // no GPU: refuse loudly, there is no CPU path
  if (!parse_devices(std::getenv("RATELIB_AMD_DEVICES"), n, g_devices)) return -1; // names a device we cannot run on
  if (g_devices.empty() && !rsmp::device_is_gfx950()) return -1; // the code object is gfx950-only: refuse here, not at the first launch
  (void)rsmp::knobs(); // the environment is read here, once
  g_initialized = 1;
  return 0;
}

void close_ratelib(void) { g_initialized = 0; }

int RR_open(const RR_config *config, int nchannels, RR_handle **const handle) { return open_common(config, nchannels, 1, -1, handle); }

int RRX_open_batch(const RR_config *config, int nchannels, int nstreams, RR_handle **const handle)
{
  return open_common(config, nchannels, nstreams, -1, handle);
}

int RRX_open_batch_on(const RR_config *config, int nchannels, int nstreams, int device, RR_handle **const handle)
{
  if (device < 0) {
    if (handle) *handle = nullptr;
    return RR_INVPARAM;
  }
  return open_common(config, nchannels, nstreams, device, handle);
}

int RRX_device(const RR_handle *h) { return h ? h->eng->device() : -1; }

int RR_push(RR_handle *h, const fb_sample_t *ibuf, size_t isamp)
{
  if (!h) return RR_NULLHANDLE;
  return guarded(h, [&] { return h->eng->push_host(ibuf, isamp, isamp); });
}

int RR_pull(RR_handle *h, fb_sample_t *obuf, size_t osamp, size_t *ogen)
{
  if (!h) return RR_NULLHANDLE;
  size_t n = osamp < h->eng->available() ? osamp : h->eng->available();
  return guarded(h, [&] { return h->eng->pull_host(obuf, n, osamp, ogen); });
}

int RR_flow(RR_handle *h, const fb_sample_t *ibuf, fb_sample_t *obuf, size_t isamp, size_t osamp, size_t *iused, size_t *ogen)
{
  if (!h) return RR_NULLHANDLE;
  if (h->eng->nstreams() != 1) return RR_INVPARAM; // packed layout of a batch is ambiguous here
  return guarded(h, [&] { return h->eng->flow_host(ibuf, isamp, obuf, osamp, isamp, osamp, iused, ogen); });
}

int RR_drain(RR_handle *h)
{
  if (!h) return RR_NULLHANDLE;
  return guarded(h, [&] { return h->eng->drain(); });
}

void RR_close(RR_handle **h)
{
  if (h == nullptr || *h == nullptr) return;
  delete (*h)->eng; // (~Engine selects the handle's device for its frees itself)
  delete *h;
  *h = nullptr;
}

const char *RR_strerror(int error)
{ // same strings as rate_uni.c:92-111
  switch (error) {
    case RR_OK: return "OK";
    case RR_ENOMEM: return "Not enough memory";
    case RR_INTERNAL: return "Internal error";
    case RR_NULLHANDLE: return "NULL handle";
    case RR_RATEERROR: return "Error in rate() functions";
    case RR_EXTUNINIT: return "Externals not initialized";
    default: return "Other error";
  }
}

int RRX_push_device(RR_handle *h, const fb_sample_t *d_ibuf, size_t in_stride, size_t isamp)
{
  if (!h) return RR_NULLHANDLE;
  return guarded(h, [&] { return h->eng->push_device(d_ibuf, in_stride, isamp); });
}

int RRX_pull_device(RR_handle *h, fb_sample_t *d_obuf, size_t out_stride, size_t osamp, size_t *ogen)
{
  if (!h) return RR_NULLHANDLE;
  return guarded(h, [&] { return h->eng->pull_device(d_obuf, out_stride, osamp, ogen); });
}

int RRX_flow_device(RR_handle *h, const fb_sample_t *d_ibuf, size_t in_stride, fb_sample_t *d_obuf, size_t out_stride,
                    size_t isamp, size_t osamp, size_t *iused, size_t *ogen)
{
  if (!h) return RR_NULLHANDLE;
  return guarded(h, [&] { return h->eng->flow_device(d_ibuf, in_stride, d_obuf, out_stride, isamp, osamp, iused, ogen); });
}

int RRX_push_strided(RR_handle *h, const fb_sample_t *ibuf, size_t in_stride, size_t isamp)
{
  if (!h) return RR_NULLHANDLE;
  return guarded(h, [&] { return h->eng->push_host(ibuf, in_stride, isamp); });
}

int RRX_pull_strided(RR_handle *h, fb_sample_t *obuf, size_t out_stride, size_t osamp, size_t *ogen)
{
  if (!h) return RR_NULLHANDLE;
  return guarded(h, [&] { return h->eng->pull_host(obuf, out_stride, osamp, ogen); });
}

int RRX_set_stream(RR_handle *h, void *hip_stream)
{
  if (!h) return RR_NULLHANDLE;
  const bool own = hip_stream == RRX_STREAM_OWN;
  return guarded(h, [&] { return h->eng->set_stream(own ? nullptr : static_cast<hipStream_t>(hip_stream), own); });
}

int RRX_sync(RR_handle *h)
{
  if (!h) return RR_NULLHANDLE;
  return guarded(h, [&] { return h->eng->sync(); });
}

int RRX_profile(RR_handle *h, int enable)
{
  if (!h) return RR_NULLHANDLE;
  return guarded(h, [&] { h->eng->set_profiling(enable != 0); return int(RR_OK); });
}

int RRX_profile_read(RR_handle *h, double *hot_ms, long long *hot_launches, double *other_ms, long long *other_launches)
{
  if (!h) return RR_NULLHANDLE;
  return guarded(h, [&] { return h->eng->read_profile(hot_ms, hot_launches, other_ms, other_launches); });
}

int RRX_profile_report(RR_handle *h, char *buf, size_t cap)
{
  if (!h) return -RR_NULLHANDLE;
  if (!buf || !cap) return -RR_INVPARAM;
  std::string s;
  int rc = guarded(h, [&] { return h->eng->read_profile_json(s); });
  if (rc) return -rc;
  size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
  std::memcpy(buf, s.data(), n);
  buf[n] = 0;
  return int(n);
}

void RRX_debug_fail_alloc(int nth) { rsmp::Engine::fail_alloc_after(nth); }

size_t RRX_isamp_max(const RR_handle *h) { return h ? h->eng->isamp_max() : 0; }
size_t RRX_available(const RR_handle *h) { return h ? h->eng->available() : 0; }
int RRX_channels(const RR_handle *h) { return h ? h->eng->nch() : 0; }
int RRX_streams(const RR_handle *h) { return h ? h->eng->nstreams() : 0; }

int RRX_describe_plan(const RR_config *config, char *buf, size_t cap)
{
  if (!config || !buf || !cap) return -RR_INVPARAM;
  rsmp::ChainPlan plan;
  int rc = rsmp::make_plan(to_config(config), plan);
  if (rc) return -rc;
  std::string s = plan.describe();
  size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
  std::memcpy(buf, s.data(), n);
  buf[n] = 0;
  return int(n);
}

int RRX_describe_dispatch(const RR_config *config, int nchannels, char *buf, size_t cap)
{
  if (!config || !buf || !cap || nchannels < 1) return -RR_INVPARAM;
  rsmp::ChainPlan plan;
  int rc = rsmp::make_plan(to_config(config), plan);
  if (rc) return -rc;
  int nsub = 0, vs = 0;
  const bool sub = !plan.stages.empty() && rsmp::split_geometry(plan, nchannels, 0, nsub, vs);
  std::string s = "{\"sub_blocked\": ";
  s += sub ? "true" : "false";
  if (sub) {
    const rsmp::DftFilter &f = plan.dft[plan.stages[0].filt];
    const int V = f.N - (f.num_taps - 1), Pref = f.N / 2;
    const bool two = nsub == 1 && vs > rsmp::kSplitVsMax;
    char tmp[256];
    snprintf(tmp, sizeof tmp, ", \"two_round\": %s, \"nsub\": %d, \"Vs\": %d, \"V\": %d, \"taps\": %d, \"N\": %d, \"Pref\": %d, \"sub_blocks\": [",
             two ? "true" : "false", nsub, vs, V, f.num_taps, f.N, Pref);
    s += tmp;
    for (int i = 0; i < nsub; ++i) {
      const rsmp::SubBlock sb = rsmp::sub_block(i, V, vs, Pref);
      snprintf(tmp, sizeof tmp, "%s{\"off\": %d, \"len\": %d, \"win\": %d, \"shift\": %d}", i ? ", " : "", sb.off, sb.len, sb.win, sb.shift);
      s += tmp;
    }
    s += "]";
  }
  s += "}";
  size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
  std::memcpy(buf, s.data(), n);
  buf[n] = 0;
  return int(n);
}

int RRX_plan_table(const RR_config *config, int which, double *out, size_t cap, size_t *count)
{
  if (!config || which < 0 || which > 2) return RR_INVPARAM;
  rsmp::ChainPlan plan;
  int rc = rsmp::make_plan(to_config(config), plan);
  if (rc) return rc;
  const std::vector<double> &t = which == 2 ? plan.poly_table : plan.dft[which].taps;
  if (count) *count = t.size();
  if (out)
    for (size_t i = 0; i < t.size() && i < cap; ++i) out[i] = t[i];
  return RR_OK;
}

} // extern "C"
