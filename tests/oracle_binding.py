"""ctypes binding of the CPU oracle (oracle/rate_oracle.c).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "librate_oracle.so")


class OrcConfig(C.Structure):
    _fields_ = [("in_rate", C.c_size_t), ("out_rate", C.c_size_t), ("phase", C.c_double),
                ("bandwidth", C.c_double), ("allow_aliasing", C.c_int), ("quality", C.c_int)]


class OrcStageInfo(C.Structure):
    _fields_ = [("kind", C.c_int), ("L", C.c_int), ("step_int", C.c_int), ("at", C.c_int64),
                ("step", C.c_int64), ("n", C.c_int), ("interp_order", C.c_int), ("phase_bits", C.c_int),
                ("pre", C.c_int), ("pre_post", C.c_int), ("preload", C.c_int), ("remL", C.c_int),
                ("num_taps", C.c_int), ("dft_length", C.c_int), ("post_peak", C.c_int),
                ("out_in_ratio", C.c_double)]


class OrcDesignCall(C.Structure):
    _fields_ = [("Fp", C.c_double), ("Fs", C.c_double), ("Fn", C.c_double), ("att", C.c_double),
                ("k", C.c_int), ("num_taps", C.c_int), ("beta", C.c_double)]


def build(force=False):
    srcs = [os.path.join(_ROOT, "oracle", f) for f in ("rate_oracle.c", "plugin_harness.c", "rate_oracle.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        P = C.POINTER
        L.orc_open.argtypes = [P(OrcConfig), C.c_int, P(C.c_void_p)]
        L.orc_push.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_pull.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, P(C.c_size_t)]
        L.orc_flow.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                               P(C.c_size_t), P(C.c_size_t)]
        L.orc_drain.argtypes = [C.c_void_p]
        L.orc_close.argtypes = [P(C.c_void_p)]
        L.orc_close.restype = None
        L.orc_isamp_max.argtypes = [C.c_void_p]
        L.orc_isamp_max.restype = C.c_size_t
        L.orc_num_stages.argtypes = [C.c_void_p]
        L.orc_stage_info_get.argtypes = [C.c_void_p, C.c_int, P(OrcStageInfo)]
        for name in ("orc_dft_taps", "orc_dft_spectrum"):
            f = getattr(L, name)
            f.argtypes = [C.c_void_p, C.c_int, P(C.c_int)]
            f.restype = P(C.c_double)
        L.orc_poly_table.argtypes = [C.c_void_p, P(C.c_int)]
        L.orc_poly_table.restype = P(C.c_double)
        L.orc_stage_fifo.argtypes = [C.c_void_p, C.c_int, C.c_int, P(C.c_int)]
        L.orc_stage_fifo.restype = P(C.c_double)
        L.orc_design_trace.argtypes = [C.c_void_p, P(OrcDesignCall), C.c_int]
        L.orc_rdft.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.orc_rdft.restype = None
        L.orc_bessel_I0.argtypes = [C.c_double]
        L.orc_bessel_I0.restype = C.c_double
        L.orc_dft_length.argtypes = [C.c_int]
        L.orc_kaiser_beta.argtypes = [C.c_double, C.c_double]
        L.orc_kaiser_beta.restype = C.c_double
        L.orc_dsp_create.argtypes = [C.c_int] * 5
        L.orc_dsp_create.restype = C.c_void_p
        L.orc_dsp_create_on.argtypes = [C.c_void_p] + [C.c_int] * 5
        L.orc_dsp_create_on.restype = C.c_void_p
        L.orc_dsp_last_error.argtypes = [C.c_void_p]
        L.orc_dsp_destroy.argtypes = [C.c_void_p]
        L.orc_dsp_destroy.restype = None
        L.orc_dsp_on_chunk.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_uint]
        L.orc_dsp_end_of_track.argtypes = [C.c_void_p]
        L.orc_dsp_end_of_track.restype = None
        L.orc_dsp_flush.argtypes = [C.c_void_p]
        L.orc_dsp_flush.restype = None
        L.orc_dsp_latency.argtypes = [C.c_void_p]
        L.orc_dsp_latency.restype = C.c_double
        L.orc_dsp_out_count.argtypes = [C.c_void_p]
        L.orc_dsp_out_count.restype = C.c_size_t
        L.orc_dsp_out_frames.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_dsp_out_frames.restype = C.c_size_t
        L.orc_dsp_out_channels.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_dsp_out_channels.restype = C.c_uint
        L.orc_dsp_out_rate.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_dsp_out_rate.restype = C.c_uint
        L.orc_dsp_out_data.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_dsp_out_data.restype = P(C.c_float)
        L.orc_dsp_out_clear.argtypes = [C.c_void_p]
        L.orc_dsp_out_clear.restype = None
        _lib = L
    return _lib


KIND_NAMES = {0: "half", 1: "dft", 2: "poly"}


class Oracle:
    """One open oracle handle; same call shapes as the RR_* API (frames, interleaved float32)."""

    def __init__(self, in_rate, out_rate, nch=2, phase=50.0, bandwidth=95.0, allow_aliasing=0, quality=0):
        self.L = lib()
        self.nch = nch
        self.cfg = OrcConfig(in_rate, out_rate, phase, bandwidth, allow_aliasing, quality)
        self.h = C.c_void_p()
        rc = self.L.orc_open(C.byref(self.cfg), nch, C.byref(self.h))
        if rc:
            raise ValueError("orc_open failed: %d" % rc)

    def close(self):
        if self.h:
            self.L.orc_close(C.byref(self.h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def isamp_max(self):
        return self.L.orc_isamp_max(self.h)

    def push(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.nch)
        return self.L.orc_push(self.h, x.ctypes.data, x.shape[0])

    def pull(self, max_frames):
        out = np.empty((max_frames, self.nch), dtype=np.float32)
        n = C.c_size_t(0)
        self.L.orc_pull(self.h, out.ctypes.data, max_frames, C.byref(n))
        return out[: n.value]

    def pull_all(self, chunk=1 << 16):
        parts = []
        while True:
            p = self.pull(chunk)
            if p.shape[0] == 0:
                break
            parts.append(p.copy())
        return np.concatenate(parts) if parts else np.empty((0, self.nch), np.float32)

    def flow(self, x, max_out):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.nch)
        out = np.empty((max_out, self.nch), dtype=np.float32)
        iu, og = C.c_size_t(0), C.c_size_t(0)
        self.L.orc_flow(self.h, x.ctypes.data if x.shape[0] else None, out.ctypes.data, x.shape[0], max_out,
                        C.byref(iu), C.byref(og))
        return iu.value, out[: og.value]

    def drain(self):
        return self.L.orc_drain(self.h)

    def process(self, x, chunk=None):
        """push everything (in `chunk`-frame pushes, default isamp_max), drain, return all output."""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.nch)
        chunk = chunk or self.isamp_max
        parts = []
        for s in range(0, x.shape[0], chunk):
            self.push(x[s:s + chunk])
            parts.append(self.pull_all())
        self.drain()
        parts.append(self.pull_all())
        return np.concatenate(parts)

    # ---- introspection ----
    def plan(self):
        out = []
        for i in range(self.L.orc_num_stages(self.h)):
            s = OrcStageInfo()
            self.L.orc_stage_info_get(self.h, i, C.byref(s))
            d = {f[0]: getattr(s, f[0]) for f in OrcStageInfo._fields_}
            d["kind"] = KIND_NAMES[d["kind"]]
            out.append(d)
        return out

    def dft_taps(self, which):
        n = C.c_int(0)
        p = self.L.orc_dft_taps(self.h, which, C.byref(n))
        return np.ctypeslib.as_array(p, (n.value,)).copy() if n.value else np.empty(0)

    def dft_spectrum(self, which):
        n = C.c_int(0)
        p = self.L.orc_dft_spectrum(self.h, which, C.byref(n))
        return np.ctypeslib.as_array(p, (n.value,)).copy() if n.value else np.empty(0)

    def poly_table(self):
        n = C.c_int(0)
        p = self.L.orc_poly_table(self.h, C.byref(n))
        return np.ctypeslib.as_array(p, (n.value,)).copy() if n.value else np.empty(0)

    def stage_fifo(self, channel, stage):
        n = C.c_int(0)
        p = self.L.orc_stage_fifo(self.h, channel, stage, C.byref(n))
        return np.ctypeslib.as_array(p, (n.value,)).copy() if n.value else np.empty(0)

    def design_trace(self):
        arr = (OrcDesignCall * 16)()
        n = min(self.L.orc_design_trace(self.h, arr, 16), 16)
        return [{f[0]: getattr(arr[i], f[0]) for f in OrcDesignCall._fields_} for i in range(n)]


class RrApi(C.Structure):
    """orc_rr_api: the five ratelib.h entry points the plugin binds (chain.h:36-40)."""
    _fields_ = [("open", C.c_void_p), ("push", C.c_void_p), ("pull", C.c_void_p), ("drain", C.c_void_p),
                ("close", C.c_void_p)]


def product_api():
    """Function table over the PRODUCT's C ABI (libratelib_amd.so: RR_open / RR_push / RR_pull / RR_drain /
    RR_close).  Needs a GPU once anything is opened through it."""
    import foo_dsp_resampler_amd.ratelib as R
    R._ensure_init()
    L = R.lib()
    return RrApi(*[C.cast(getattr(L, n), C.c_void_p).value for n in ("RR_open", "RR_push", "RR_pull", "RR_drain", "RR_close")])


class OracleDsp:
    """The plugin's dsp_rate object restated as a test harness (oracle/plugin_harness.c) over a table of ABI entry
    points: `api=None` drives the CPU oracle, `api=product_api()` drives libratelib_amd.so."""

    def __init__(self, out_rate, quality=0, allow_aliasing=0, passband10=950, phase=50, api=None):
        self.L = lib()
        self._api = api  # keep the table alive (the harness copies it, but the library handle must outlive us)
        self.h = self.L.orc_dsp_create_on(C.byref(api) if api is not None else None, out_rate, quality, allow_aliasing,
                                          passband10, phase)

    def close(self):
        if self.h:
            self.L.orc_dsp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _take(self):
        out = []
        for i in range(self.L.orc_dsp_out_count(self.h)):
            n, ch = self.L.orc_dsp_out_frames(self.h, i), self.L.orc_dsp_out_channels(self.h, i)
            p = self.L.orc_dsp_out_data(self.h, i)
            out.append((np.ctypeslib.as_array(p, (n * ch,)).reshape(n, ch).copy(), self.L.orc_dsp_out_rate(self.h, i)))
        self.L.orc_dsp_out_clear(self.h)
        return out

    def on_chunk(self, x, sample_rate, channel_config=3):
        x = np.ascontiguousarray(x, dtype=np.float32)
        x = x.reshape(x.shape[0], -1)
        p = self.L.orc_dsp_on_chunk(self.h, x.ctypes.data, x.shape[0], x.shape[1], sample_rate, channel_config)
        return bool(p), self._take()

    def end_of_track(self):
        self.L.orc_dsp_end_of_track(self.h)
        return self._take()

    def flush(self):
        self.L.orc_dsp_flush(self.h)

    @property
    def latency(self):
        return self.L.orc_dsp_latency(self.h)

    @property
    def last_error(self):
        return self.L.orc_dsp_last_error(self.h)


def PluginOnGpu(out_rate, **kw):
    """The same harness over the product's RR_* entry points."""
    return OracleDsp(out_rate, api=product_api(), **kw)


def lcg_noise(n_frames, nch, seed):
    """SURVEY.md 8(d) synthetic input: s = s*1664525 + 1013904223; sample = ((s>>8) - 2^23)/2^23 * 0.5."""
    n = n_frames * nch
    out = np.empty(n, dtype=np.float32)
    # vectorised LCG via jump-ahead in blocks
    s = np.uint64(seed & 0xFFFFFFFF)
    a, c, mask = 1664525, 1013904223, 0xFFFFFFFF
    # closed form in chunks: generate sequentially in python ints per chunk of 1<<16 using numpy cumulative trick
    state = int(s)
    B = 1 << 16
    # precompute a^k and c_k for k=1..B
    ak = np.empty(B, dtype=np.uint64)
    ck = np.empty(B, dtype=np.uint64)
    aa, cc = 1, 0
    for k in range(B):
        aa = (aa * a) & mask
        cc = (cc * a + c) & mask
        ak[k] = aa
        ck[k] = cc
    pos = 0
    while pos < n:
        m = min(B, n - pos)
        vals = (ak[:m] * np.uint64(state) + ck[:m]) & np.uint64(mask)
        out[pos:pos + m] = ((vals >> np.uint64(8)).astype(np.float64) - 8388608.0) / 8388608.0 * 0.5
        state = int(vals[m - 1])
        pos += m
    return out.reshape(n_frames, nch)
