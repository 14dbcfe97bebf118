"""ctypes mirror of include/ratelib.h + include/ratelib_amd.h (same names, argument meaning and
error behaviour as the reference's rate/ratelib.h:25-81)."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
RR_BEST, RR_NORM = 0, 1

# every symbol the two public headers declare
EXPECTED_SYMBOLS = [
    "init_ratelib", "close_ratelib", "RR_open", "RR_flow", "RR_push", "RR_pull", "RR_drain", "RR_close", "RR_strerror",
    "RRX_open_batch", "RRX_open_batch_on", "RRX_device", "RRX_push_device", "RRX_pull_device", "RRX_flow_device", "RRX_push_strided", "RRX_pull_strided",
    "RRX_set_stream", "RRX_sync", "RRX_profile", "RRX_profile_read", "RRX_profile_report", "RRX_debug_fail_alloc", "RRX_isamp_max", "RRX_available", "RRX_channels", "RRX_streams",
    "RRX_describe_plan", "RRX_describe_dispatch", "RRX_plan_table",
]


class RRConfig(C.Structure):
    """RR_config, rate/ratelib.h:53-63."""
    _fields_ = [("in_rate", C.c_size_t), ("out_rate", C.c_size_t), ("phase", C.c_double),
                ("bandwidth", C.c_double), ("allow_aliasing", C.c_int), ("quality", C.c_int)]


class RRError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__("%s failed: %d (%s)" % (what, code, lib().RR_strerror(code).decode()))


def lib_path():
    # RATELIB_AMD_SO: another in-tree build of the same library (kernel experiments: tools/build_variant.sh)
    return os.environ.get("RATELIB_AMD_SO") or os.path.join(HERE, "libratelib_amd.so")


_lib = None
_ALLOC_CB = C.CFUNCTYPE(None)


alloc_handler_calls = 0  # how often the library ran the registered allocation-failure handler (xmalloc.c:38-43)


def _alloc_failed():
    # The plugin's handler throws std::bad_alloc through the C frames; a Python callback cannot unwind C, so this one
    # only counts -- the failing call still returns RR_ENOMEM, which _check() raises as RRError.
    global alloc_handler_calls
    alloc_handler_calls += 1


_alloc_cb = _ALLOC_CB(_alloc_failed)
_inited = False


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so / libhsa-runtime64.so and load them by file
    name; a process that also loads /opt/rocm's copy ends up with two HSA runtimes and the second one
    finds no device.  Loading torch's copy first (same SONAME, libamdhip64.so.7) makes the dynamic
    linker bind libratelib_amd.so to it, so torch tensors and this engine share one runtime.  Without
    torch installed the system runtime is used."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return None
        p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(p):
            return C.CDLL(p, mode=C.RTLD_GLOBAL)
    except Exception:
        pass
    return None


_hip_rt = None


def lib():
    """Load the shared library (raises if it has not been built: there is no fallback)."""
    global _lib, _hip_rt
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RuntimeError("libratelib_amd.so is not built; run `python -m foo_dsp_resampler_amd.build`")
        _hip_rt = _share_hip_runtime_with_torch()
        L = C.CDLL(p)
        P = C.POINTER
        vp, sz = C.c_void_p, C.c_size_t
        L.init_ratelib.argtypes = [_ALLOC_CB]
        L.RR_open.argtypes = [P(RRConfig), C.c_int, P(vp)]
        L.RRX_open_batch.argtypes = [P(RRConfig), C.c_int, C.c_int, P(vp)]
        L.RRX_open_batch_on.argtypes = [P(RRConfig), C.c_int, C.c_int, C.c_int, P(vp)]
        L.RRX_device.argtypes = [vp]
        L.RR_push.argtypes = [vp, vp, sz]
        L.RR_pull.argtypes = [vp, vp, sz, P(sz)]
        L.RR_flow.argtypes = [vp, vp, vp, sz, sz, P(sz), P(sz)]
        L.RR_drain.argtypes = [vp]
        L.RR_close.argtypes = [P(vp)]
        L.RR_close.restype = None
        L.RR_strerror.argtypes = [C.c_int]
        L.RR_strerror.restype = C.c_char_p
        L.RRX_push_device.argtypes = [vp, vp, sz, sz]
        L.RRX_pull_device.argtypes = [vp, vp, sz, sz, P(sz)]
        L.RRX_flow_device.argtypes = [vp, vp, sz, vp, sz, sz, sz, P(sz), P(sz)]
        L.RRX_push_strided.argtypes = [vp, vp, sz, sz]
        L.RRX_pull_strided.argtypes = [vp, vp, sz, sz, P(sz)]
        L.RRX_set_stream.argtypes = [vp, vp]
        L.RRX_sync.argtypes = [vp]
        L.RRX_profile.argtypes = [vp, C.c_int]
        L.RRX_profile_read.argtypes = [vp, P(C.c_double), P(C.c_longlong), P(C.c_double), P(C.c_longlong)]
        L.RRX_profile_report.argtypes = [vp, C.c_char_p, sz]
        L.RRX_debug_fail_alloc.argtypes = [C.c_int]
        L.RRX_debug_fail_alloc.restype = None
        for n in ("RRX_isamp_max", "RRX_available"):
            getattr(L, n).argtypes = [vp]
            getattr(L, n).restype = sz
        L.RRX_channels.argtypes = [vp]
        L.RRX_streams.argtypes = [vp]
        L.RRX_describe_plan.argtypes = [P(RRConfig), C.c_char_p, sz]
        L.RRX_describe_dispatch.argtypes = [P(RRConfig), C.c_int, C.c_char_p, sz]
        L.RRX_plan_table.argtypes = [P(RRConfig), C.c_int, vp, sz, P(sz)]
        _lib = L
    return _lib


def available_symbols():
    L = lib()
    return [s for s in EXPECTED_SYMBOLS if hasattr(L, s)]


def _config(in_rate, out_rate, phase=50.0, bandwidth=95.0, allow_aliasing=0, quality=RR_BEST):
    return RRConfig(int(in_rate), int(out_rate), float(phase), float(bandwidth), int(allow_aliasing), int(quality))


def describe_plan(in_rate, out_rate, **kw):
    """Host-only: the stage chain the planner builds (dict).  Needs no GPU."""
    cfg = _config(in_rate, out_rate, **kw)
    buf = C.create_string_buffer(1 << 16)
    n = lib().RRX_describe_plan(C.byref(cfg), buf, len(buf))
    if n < 0:
        raise RRError(-n, "RRX_describe_plan")
    return json.loads(buf.value.decode())


def describe_dispatch(in_rate, out_rate, nch, **kw):
    """Host-only: the kernel form of the first stage pair on `nch`-channel handles (dict; RRX_describe_dispatch).  Needs no GPU."""
    cfg = _config(in_rate, out_rate, **kw)
    buf = C.create_string_buffer(1 << 14)
    n = lib().RRX_describe_dispatch(C.byref(cfg), int(nch), buf, len(buf))
    if n < 0:
        raise RRError(-n, "RRX_describe_dispatch")
    return json.loads(buf.value.decode())


def plan_table(which, in_rate, out_rate, **kw):
    """Host-only: designed table (0/1: DFT-stage taps, 2: polyphase table) as float64 array."""
    cfg = _config(in_rate, out_rate, **kw)
    n = C.c_size_t(0)
    rc = lib().RRX_plan_table(C.byref(cfg), which, None, 0, C.byref(n))
    if rc:
        raise RRError(rc, "RRX_plan_table")
    out = np.empty(n.value, dtype=np.float64)
    if n.value:
        lib().RRX_plan_table(C.byref(cfg), which, out.ctypes.data, n.value, C.byref(n))
    return out


def _ensure_init():
    global _inited
    if not _inited:
        if lib().init_ratelib(_alloc_cb) != 0:
            raise RuntimeError("init_ratelib failed: no usable HIP device (the engine has no CPU path)")
        _inited = True


def _check(rc, what):
    if rc:
        raise RRError(rc, what)


class Resampler:
    """One RR_handle (optionally a batch of lock-stepped streams).

    Host arrays are numpy float32 shaped [frames, nch] (one stream) or [streams, frames, nch].
    Device buffers are anything with `data_ptr()` (torch CUDA tensors) of the same shapes.
    """

    def __init__(self, in_rate, out_rate, nch=2, nstreams=1, device=None, **kw):
        _ensure_init()
        self.L = lib()
        self.nch, self.nstreams = nch, nstreams
        self.cfg = _config(in_rate, out_rate, **kw)
        self.h = C.c_void_p()
        if device is not None:  # explicit HIP device index (RRX_open_batch_on)
            _check(self.L.RRX_open_batch_on(C.byref(self.cfg), nch, nstreams, int(device), C.byref(self.h)), "RRX_open_batch_on")
        elif nstreams == 1:
            _check(self.L.RR_open(C.byref(self.cfg), nch, C.byref(self.h)), "RR_open")
        else:
            _check(self.L.RRX_open_batch(C.byref(self.cfg), nch, nstreams, C.byref(self.h)), "RRX_open_batch")

    # -- lifecycle
    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.RR_close(C.byref(self.h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device(self):
        return self.L.RRX_device(self.h)

    @property
    def isamp_max(self):
        return self.L.RRX_isamp_max(self.h)

    @property
    def available(self):
        return self.L.RRX_available(self.h)

    def set_stream(self, hip_stream_ptr):
        """hipStream_t as an integer (torch: stream.cuda_stream); 0 / None = the default stream."""
        _check(self.L.RRX_set_stream(self.h, C.c_void_p(hip_stream_ptr or 0)), "RRX_set_stream")

    def use_own_stream(self):
        """Back to the stream the handle created for itself (RRX_STREAM_OWN)."""
        _check(self.L.RRX_set_stream(self.h, C.c_void_p(C.c_size_t(-1).value)), "RRX_set_stream")

    def sync(self):
        _check(self.L.RRX_sync(self.h), "RRX_sync")

    def profile(self, enable=True):
        _check(self.L.RRX_profile(self.h, 1 if enable else 0), "RRX_profile")

    def profile_read(self):
        hm, om = C.c_double(0), C.c_double(0)
        hn, on = C.c_longlong(0), C.c_longlong(0)
        _check(self.L.RRX_profile_read(self.h, C.byref(hm), C.byref(hn), C.byref(om), C.byref(on)), "RRX_profile_read")
        return {"hot_ms": hm.value, "hot_launches": hn.value, "other_ms": om.value, "other_launches": on.value}

    def profile_report(self):
        """Per-kernel records since the last read: [{"kernel", "hot", "launches", "ms"}, ...]."""
        buf = C.create_string_buffer(1 << 14)
        n = self.L.RRX_profile_report(self.h, buf, len(buf))
        if n < 0:
            raise RRError(-n, "RRX_profile_report")
        return json.loads(buf.value.decode())

    # -- host API (RR_push / RR_pull / RR_flow / RR_drain)
    def _host_in(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        if self.nstreams == 1:
            x = x.reshape(-1, self.nch)
            return x, x.shape[0]
        x = x.reshape(self.nstreams, -1, self.nch)
        return x, x.shape[1]

    def push(self, x):
        x, n = self._host_in(x)
        if n == 0:
            return
        if self.nstreams == 1:
            _check(self.L.RR_push(self.h, x.ctypes.data, n), "RR_push")
        else:
            _check(self.L.RRX_push_strided(self.h, x.ctypes.data, n, n), "RRX_push_strided")

    def pull(self, max_frames):
        shape = (max_frames, self.nch) if self.nstreams == 1 else (self.nstreams, max_frames, self.nch)
        out = np.empty(shape, dtype=np.float32)
        n = C.c_size_t(0)
        if self.nstreams == 1:
            _check(self.L.RR_pull(self.h, out.ctypes.data, max_frames, C.byref(n)), "RR_pull")
            return out[: n.value]
        _check(self.L.RRX_pull_strided(self.h, out.ctypes.data, max_frames, max_frames, C.byref(n)), "RRX_pull_strided")
        return out[:, : n.value]

    def pull_all(self, chunk=1 << 16):
        parts = []
        ax = 0 if self.nstreams == 1 else 1
        while True:
            p = self.pull(chunk)
            if p.shape[ax] == 0:
                break
            parts.append(p.copy())
        if parts:
            return np.concatenate(parts, axis=ax)
        return np.empty((0, self.nch) if self.nstreams == 1 else (self.nstreams, 0, self.nch), np.float32)

    def flow(self, x, max_out):
        assert self.nstreams == 1
        x, n = self._host_in(x)
        out = np.empty((max_out, self.nch), dtype=np.float32)
        iu, og = C.c_size_t(0), C.c_size_t(0)
        _check(self.L.RR_flow(self.h, x.ctypes.data if n else None, out.ctypes.data, n, max_out, C.byref(iu), C.byref(og)),
               "RR_flow")
        return iu.value, out[: og.value]

    def drain(self):
        _check(self.L.RR_drain(self.h), "RR_drain")

    def process(self, x, chunk=None):
        """push everything in `chunk`-frame pushes (default isamp_max), drain, return all output."""
        x, n = self._host_in(x)
        chunk = chunk or self.isamp_max
        ax = 0 if self.nstreams == 1 else 1
        parts = []
        for s in range(0, n, chunk):
            self.push(x[s:s + chunk] if self.nstreams == 1 else x[:, s:s + chunk])
            parts.append(self.pull_all())
        self.drain()
        parts.append(self.pull_all())
        return np.concatenate(parts, axis=ax)

    # -- device API (buffers expose data_ptr(); strides in frames)
    def push_device(self, t, frames, stride=None):
        _check(self.L.RRX_push_device(self.h, C.c_void_p(t.data_ptr()), stride or frames, frames), "RRX_push_device")

    def pull_device(self, t, max_frames, stride=None):
        n = C.c_size_t(0)
        _check(self.L.RRX_pull_device(self.h, C.c_void_p(t.data_ptr()), stride or max_frames, max_frames, C.byref(n)),
               "RRX_pull_device")
        return n.value

    def flow_device(self, tin, in_frames, tout, out_cap, in_stride=None, out_stride=None):
        iu, og = C.c_size_t(0), C.c_size_t(0)
        _check(self.L.RRX_flow_device(self.h, C.c_void_p(tin.data_ptr()) if tin is not None else None,
                                      in_stride or in_frames, C.c_void_p(tout.data_ptr()), out_stride or out_cap,
                                      in_frames, out_cap, C.byref(iu), C.byref(og)), "RRX_flow_device")
        return iu.value, og.value
