"""Plugin call pattern (dsp_rate::on_chunk / flushwrite / get_latency + LPC edge extrapolation).

ONE chunk-logic harness (oracle/plugin_harness.c, test infrastructure) parameterised by a table of
{open, push, pull, drain, close} entry points.  CPU part: the harness over the oracle's orc_* functions
(lengths, latency accounting, edges) -- there is no reference fixture for this path, so these are consistency
checks.  GPU part: the same harness over the product's RR_* entry points (libratelib_amd.so) against the run
over the oracle, chunk for chunk: same chunking, same frame counts, same latency values, samples within the
parity bar -- i.e. the resampler behind the ABI behaves identically under the plugin's real call pattern."""
import numpy as np
import pytest

from oracle_binding import OracleDsp, PluginOnGpu, lcg_noise


def music_like(n, nch, fs, seed):
    """band-limited, strongly predictable signal (LPC extrapolation is meant for audio, not white noise)"""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    x = np.zeros((n, nch))
    for c in range(nch):
        for k in range(6):
            f = rng.uniform(80, 5000)
            x[:, c] += rng.uniform(.05, .15) * np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28))
    return x.astype(np.float32)


def run_track(dsp, x, fs, chunk_sizes):
    outs, pos, k, lat = [], 0, 0, []
    while pos < x.shape[0]:
        n = chunk_sizes[k % len(chunk_sizes)]
        k += 1
        passthrough, chunks = dsp.on_chunk(x[pos:pos + n], fs)
        assert not passthrough
        pos += n
        outs += chunks
        lat.append(dsp.latency)
    outs += dsp.end_of_track()
    return outs, lat


@pytest.mark.parametrize("fs,fo,n", [(44100, 96000, 120000), (96000, 44100, 150000), (44100, 48000, 30000)])
def test_oracle_plugin_track_length_and_latency(fs, fo, n):
    x = music_like(n, 2, fs, 1)
    outs, lat = run_track(OracleDsp(fo), x, fs, [4096, 1000, 3333])
    total = sum(c.shape[0] for c, _ in outs)
    assert all(r == fo for _, r in outs)
    # what comes out is the resampled track itself: both extrapolated edges are cut away again
    assert abs(total - n * fo / fs) <= 2
    assert all(-1e-9 <= v < 0.5 for v in lat)          # bounded: staging + filter delay, never negative
    y = np.concatenate([c for c, _ in outs])
    # interior of the track equals a plain resampling of the interior (edges only differ by the extrapolation)
    assert np.isfinite(y).all() and np.max(np.abs(y)) < 1.5


def test_oracle_plugin_short_tracks_and_passthrough():
    d = OracleDsp(48000)
    p, chunks = d.on_chunk(np.zeros((100, 2), np.float32), 48000)
    assert p and not chunks                              # same rate: pass through, nothing opened
    x = music_like(50, 2, 44100, 2)                      # <= 2*LPC_ORDER frames: no extrapolation (case a)
    p, chunks = d.on_chunk(x, 44100)
    assert not p and not chunks
    tail = d.end_of_track()
    assert sum(c.shape[0] for c, _ in tail) == int(50 * 48000 / 44100 + .5)
    d2 = OracleDsp(48000)                                # one short buffer: both edges extrapolated (case b)
    x = music_like(1500, 2, 44100, 3)
    d2.on_chunk(x, 44100)
    tail = d2.end_of_track()
    assert abs(sum(c.shape[0] for c, _ in tail) - 1500 * 48000 / 44100) <= 2


def test_oracle_plugin_format_change_flushes():
    d = OracleDsp(48000)
    a = music_like(30000, 2, 44100, 4)
    _, c1 = d.on_chunk(a, 44100)
    b = music_like(20000, 1, 32000, 5)
    _, c2 = d.on_chunk(b, 32000)                         # new format: old stream is finished first
    n1 = sum(c.shape[0] for c, r in c1 + c2 if c.shape[1] == 2)
    assert abs(n1 - 30000 * 48000 / 44100) <= 2
    tail = d.end_of_track()
    n2 = sum(c.shape[0] for c, r in c2 + tail if c.shape[1] == 1)
    assert abs(n2 - 20000 * 48000 / 32000) <= 2


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("fs,fo,nch,n,sizes", [
    (44100, 96000, 2, 140000, [4096, 1000, 3333]),       # steady state, case (c) at the end
    (96000, 44100, 2, 100000, [8192]),
    (44100, 48000, 2, 1500, [700]),                      # one short buffer: case (b)
    (44100, 48000, 2, 40, [40]),                         # too short to extrapolate: case (a)
    (48000, 44100, 6, 60000, [2048, 4095]),              # 5.1
])
def test_gpu_plugin_layer_matches_oracle(fs, fo, nch, n, sizes):
    from parity import assert_parity
    x = music_like(n, nch, fs, 7)
    gpu = PluginOnGpu(fo)
    got, lat_g = run_track(gpu, x, fs, sizes)
    assert gpu.last_error == 0
    ref, lat_r = run_track(OracleDsp(fo), x, fs, sizes)
    assert [c.shape for c, _ in got] == [c.shape for c, _ in ref]      # same chunking, same frame counts
    assert [r for _, r in got] == [r for _, r in ref]
    assert lat_g == lat_r                                               # latency accounting is integer-exact
    yg, yr = np.concatenate([c for c, _ in got]), np.concatenate([c for c, _ in ref])
    assert_parity(yg, yr)


@pytest.mark.gpu
def test_gpu_plugin_layer_format_change_and_passthrough():
    from parity import assert_parity
    g, o = PluginOnGpu(48000), OracleDsp(48000)
    seq = [(music_like(30000, 2, 44100, 4), 44100), (np.zeros((64, 2), np.float32), 48000),
           (music_like(20000, 1, 32000, 5), 32000)]
    for x, fs in seq:
        pg, cg = g.on_chunk(x, fs)
        po, co = o.on_chunk(x, fs)
        assert pg == po and [c.shape for c, _ in cg] == [c.shape for c, _ in co]
        for (a, _), (b, _) in zip(cg, co):
            assert_parity(a, b)
    tg, to = g.end_of_track(), o.end_of_track()
    assert [c.shape for c, _ in tg] == [c.shape for c, _ in to]
    for (a, _), (b, _) in zip(tg, to):
        assert_parity(a, b)


def test_harness_reports_errors_per_call():
    """An ABI that fails: the harness records the first RR_error of a call and forgets it on the next call
    (chain.h:26-29 raises per call)."""
    import ctypes as C
    from oracle_binding import OrcConfig, RrApi, lib
    calls = {"open": 0}
    OPEN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p))
    PUSH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)
    PULL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t))
    DRAIN = C.CFUNCTYPE(C.c_int, C.c_void_p)
    CLOSE = C.CFUNCTYPE(None, C.POINTER(C.c_void_p))
    L = lib()

    def f_open(cfg, nch, out):
        calls["open"] += 1
        if calls["open"] == 1:
            out[0] = None
            return 1  # RR_ENOMEM on the first open only
        return L.orc_open(C.cast(cfg, C.POINTER(OrcConfig)), nch, out)

    fo, fp = OPEN(f_open), PUSH(lambda h, p, n: L.orc_push(h, p, n) if h else 3)

    def f_pull(h, p, n, g):
        if not h:
            if g:
                g[0] = 0
            return 3
        return L.orc_pull(h, p, n, g)

    fl, fd = PULL(f_pull), DRAIN(lambda h: L.orc_drain(h) if h else 3)
    fc = CLOSE(lambda hp: L.orc_close(hp))
    api = RrApi(*[C.cast(f, C.c_void_p).value for f in (fo, fp, fl, fd, fc)])
    d = OracleDsp(48000, api=api)
    x = music_like(3000, 2, 44100, 9)
    d.on_chunk(x, 44100)
    assert d.last_error == 1           # the failed open is what this call reports
    d.on_chunk(x, 44100)               # handle still NULL -> reinit succeeds now
    assert d.last_error == 0
    keep = (fo, fp, fl, fd, fc)        # noqa: F841  (callbacks must outlive the harness object)
    d.close()
