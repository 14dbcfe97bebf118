"""GPU tests added in round 2 (pytest -m gpu): the per-GPU shard of BASELINE configs[4], error paths behind the C ABI,
stream ownership, chains whose DFT stage upsamples by 8 or more, and the measured parity of non-linear-phase filters.
Everything goes through the C ABI (foo_dsp_resampler_amd.ratelib) and is compared with the CPU oracle."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import foo_dsp_resampler_amd as F
import foo_dsp_resampler_amd.ratelib as R
from oracle_binding import Oracle, lcg_noise
from devbuf import dev_zeros
from parity import assert_parity, compare_f32

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cfg4_per_gpu_shard_against_oracle():
    """BASELINE configs[4] on one of 8 GPUs: 128 independent stereo 44.1k->48k streams in ONE batch handle
    (each stream is what the reference runs as a handle of its own, rate_base.h:533-540), device-resident, two
    pushes + drain.  Three streams sample by sample against the oracle, all by frame count and distinct checksums."""
    torch = pytest.importorskip("torch")
    from foo_dsp_resampler_amd.sharding import shard_range
    first, S = shard_range(1024, 8, 3)          # rank 3 of 8: streams [384, 512)
    assert (first, S) == (384, 128)
    nch, fi, fo, n = 2, 44100, 48000, 60000
    x = torch.stack([torch.from_numpy(lcg_noise(n, nch, 12345 + first + s)) for s in range(S)]).cuda()
    r = F.Resampler(fi, fo, nch=nch, nstreams=S)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    cap = int(n * fo / fi) + 4096
    y = dev_zeros((S, cap, nch))
    got = 0
    for lo, hi in ((0, 41000), (41000, n)):
        seg = x[:, lo:hi].contiguous()
        iu, og = r.flow_device(seg, hi - lo, y[:, got:], cap - got, out_stride=cap)
        assert iu == hi - lo
        got += og
    r.drain()
    og2 = r.pull_device(y[:, got:], cap - got, stride=cap)
    r.sync()
    total = got + og2
    assert total == int(n * fo / fi + .5)
    for s in (0, 77, 127):
        ref = Oracle(fi, fo, nch).process(x[s].cpu().numpy())
        assert_parity(y[s, :total].cpu().numpy(), ref)
    sums = y[:, :total].double().abs().sum(dim=(1, 2)).cpu().numpy()
    assert np.all(np.isfinite(sums)) and len(np.unique(sums)) == S


def test_enomem_runs_the_handler_once_and_leaves_no_handle():
    """A request no device could hold fifos for (4 M channels) is answered by the parameter guard in Engine::create before
    any allocation: RR_ENOMEM, the init_ratelib handler runs exactly once from the calling host frame (xmalloc.c:38-43,
    rate_uni.c:31-53), *handle stays NULL, and the library keeps working.  (A REAL device allocation failing in the middle of a
    handle's construction, and the teardown of the half-built engine, is tests/test_gpu_round3.py::
    test_allocation_failure_in_the_middle_of_open.)"""
    L = F.lib()
    R._ensure_init()
    before = R.alloc_handler_calls
    cfg = F.RRConfig(44100, 96000, 50.0, 95.0, 0, 0)
    h = C.c_void_p(0x1234)
    rc = L.RRX_open_batch(C.byref(cfg), 2, 1 << 21, C.byref(h))   # 4 M channels: refused by the guard, nothing is allocated
    assert rc == 1 and not h.value                                # RR_ENOMEM
    assert R.alloc_handler_calls == before + 1
    assert L.RR_strerror(rc).decode() == "Not enough memory"
    x = lcg_noise(5000, 2, 1)
    assert_parity(F.Resampler(44100, 96000, 2).process(x), Oracle(44100, 96000, 2).process(x))


def test_failed_push_poisons_the_handle():
    """A device allocation that fails in the middle of a push (ring growth) returns RR_ENOMEM, runs the handler once,
    and the handle then answers RR_INTERNAL to every data call instead of continuing on skewed counters; closing it and
    opening a new one works."""
    L = F.lib()
    x = lcg_noise(200000, 2, 2)
    r = F.Resampler(44100, 96000, 2)
    r.push(x[:100000])                        # sizes the host staging buffer; its output stays in the ring (no pull)
    before = R.alloc_handler_calls
    L.RRX_debug_fail_alloc(1)                 # the next device allocation fails: the output ring must double for this push
    rc = L.RR_push(r.h, x[100000:].ctypes.data, 100000)
    L.RRX_debug_fail_alloc(0)
    assert rc == 1, rc
    assert R.alloc_handler_calls == before + 1
    n = C.c_size_t(99)
    buf = np.zeros((16, 2), np.float32)
    assert L.RR_push(r.h, x.ctypes.data, 100) == 2            # RR_INTERNAL from now on
    assert L.RR_pull(r.h, buf.ctypes.data, 16, C.byref(n)) == 2 and n.value == 0
    assert L.RR_drain(r.h) == 2
    assert R.alloc_handler_calls == before + 1                 # RR_INTERNAL does not run the allocation handler
    r.close()
    y = F.Resampler(44100, 96000, 2).process(x[:20000])
    assert_parity(y, Oracle(44100, 96000, 2).process(x[:20000]))


def test_set_stream_does_not_take_ownership():
    """RRX_set_stream with a non-default stream: results equal the default-stream run, closing the handle leaves the
    caller's stream usable, RRX_STREAM_OWN restores the handle's own stream, and work is ordered across the switch."""
    torch = pytest.importorskip("torch")
    fi, fo, nch, n = 44100, 96000, 2, 50000
    x = torch.from_numpy(lcg_noise(n, nch, 5)).cuda()
    cap = int(n * fo / fi) + 4096
    ref = F.Resampler(fi, fo, nch).process(x.cpu().numpy())
    side = torch.cuda.Stream()
    r = F.Resampler(fi, fo, nch)
    y = dev_zeros((cap, nch))
    torch.cuda.synchronize()
    r.set_stream(side.cuda_stream)
    iu, og = r.flow_device(x[:30000].contiguous(), 30000, y, cap)
    r.use_own_stream()                                   # back to the handle's own stream, ordered after the work above
    iu2, og2 = r.flow_device(x[30000:].contiguous(), n - 30000, y[og:], cap - og)
    r.set_stream(side.cuda_stream)
    r.drain()
    og3 = r.pull_device(y[og + og2:], cap - og - og2)
    r.sync()
    assert og + og2 + og3 == ref.shape[0]
    assert np.array_equal(y[: ref.shape[0]].cpu().numpy(), ref)
    r.close()                                            # must not destroy `side`
    with torch.cuda.stream(side):
        z = (x * 2).sum()
    side.synchronize()
    assert np.isfinite(float(z))


@pytest.mark.parametrize("fi,fo", [(8000, 352800), (8000, 705600), (11025, 384000)])
def test_dft_stage_upsampling_by_8_or_more(fi, fo):
    """Chains whose last DFT stage has a power-of-two L >= 8 (x32 and more overall; reachable through the plugin's
    free-form target rate, dsp_config.cpp:79): served by the time-domain zero-stuffing branch, same bar."""
    plan = F.describe_plan(fi, fo)
    assert any(s["kind"] == "dft" and s["L"] >= 8 for s in plan["stages"]), plan
    x = lcg_noise(6000, 2, 17)
    got = F.Resampler(fi, fo, 2).process(x, chunk=2500)
    ref = Oracle(fi, fo, 2).process(x, chunk=2500)
    assert got.shape == ref.shape
    assert_parity(got, ref)


def test_non_linear_phase_measured_parity():
    """phase != 50 goes through the cepstral minimum-phase construction (effects_i_dsp.c:181-278).  Round 2 measured 5.5e-7 ...
    4.2e-6 relative RMS against the oracle here (profiles/r02_phase_parity.json): two fp64 FFTs, two filters.  With the
    construction's first transform in binary128 and the rest in long double on both sides every case meets the normal bar; the values go to
    gpurun_out/phase_parity.json (tracked copy: profiles/r03_phase_parity.json)."""
    out = {}
    for phase in (0.0, 25.0, 75.0, 100.0):
        for fi, fo in ((44100, 48000), (44100, 96000), (96000, 44100)):
            x = lcg_noise(30000, 2, 12345)
            got = F.Resampler(fi, fo, 2, phase=phase).process(x, chunk=8192)
            ref = Oracle(fi, fo, 2, phase=phase).process(x, chunk=8192)
            assert got.shape == ref.shape
            rep = compare_f32(got, ref)
            out["%g:%d->%d" % (phase, fi, fo)] = {"rel_rms": rep["rel_rms"], "max_ulp": rep["max_ulp"], "max_abs": rep["max_abs"]}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "phase_parity.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, v in out.items():  # the one parity bar
        assert v["max_ulp"] <= 1.0 and v["rel_rms"] <= 1e-7, (k, v)


def test_long_blocks_many_items_per_push():
    """One push that spans more (block, pair) items than one round of the four-step workspaces holds (dftbig.hip:
    ws_items), so launch_dft_big runs several rounds: 32768-point blocks, stereo, 330 000 frames in a single push."""
    fi, fo, bw = 22050, 8000, 99.0
    plan = F.describe_plan(fi, fo, bandwidth=bw)
    assert any(s["kind"] == "dft" and s["dft_length"] == 32768 for s in plan["stages"]), plan
    x = lcg_noise(330000, 2, 51)
    got = F.Resampler(fi, fo, 2, bandwidth=bw).process(x)           # one push of isamp_max or less
    ref = Oracle(fi, fo, 2, bandwidth=bw).process(x)
    assert got.shape == ref.shape
    assert_parity(got, ref)


def test_profile_report_names_the_kernels_the_dispatch_picked():
    """RRX_profile_report: per-kernel HIP-event records with the instance names rocprofv3 would print; the lean kernel
    serves a device-resident stereo push, the generic one a host push (its output goes to the ring)."""
    torch = pytest.importorskip("torch")
    r = F.Resampler(44100, 96000, nch=2, nstreams=4)
    x = torch.rand((4, 60000, 2), device="cuda") - 0.5
    y = torch.empty((4, 140000, 2), device="cuda")
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    r.profile(True)
    r.flow_device(x, 60000, y, 140000)
    rep = r.profile_report()
    names = {k["kernel"] for k in rep}
    assert "rsmp::fused_fast_kernel<11, 7, false>" in names and "rsmp::seam_kernel" in names, names
    assert all(k["launches"] >= 1 and k["ms"] > 0 for k in rep)
    r.push(x.cpu().numpy())                                         # host path: output into the ring -> generic kernel
    names = {k["kernel"] for k in r.profile_report()}
    assert "rsmp::fused_kernel<12, 11, 2, 7, true>" in names, names
    r.profile(False)


@pytest.mark.parametrize("fi,fo,kw", [(44100, 48001, {}), (96000, 44101, {}), (44100, 48001, {"quality": 1})])
def test_interpolated_polyphase_rows_shared_across_channel_groups(fi, fo, kw):
    """Irrational ratios (vpoly1-3, rate_filters_generic.h:311-504): polyi_kernel computes the interpolated coefficient
    rows of a tile once and applies them to 16 channels at a time; 7 streams x 5 channels = 35 channels make two full
    groups and a partial one.  Every stream against the oracle, several pushes."""
    S, nch, n = 7, 5, 21000
    xs = np.stack([lcg_noise(n, nch, 700 + s) for s in range(S)])
    got = F.Resampler(fi, fo, nch=nch, nstreams=S, **kw).process(xs, chunk=6500)
    for s in range(S):
        ref = Oracle(fi, fo, nch, **kw).process(xs[s], chunk=6500)
        assert got[s].shape == ref.shape
        assert_parity(got[s], ref)


@pytest.mark.parametrize("fi,fo,kw", [
    (44100, 48000, {}),                       # dft x2 -> vpoly0, fused
    (96000, 44100, {}),                       # dft -> vpoly0 147/320, fused (vector variant)
    (48000, 192000, {}),                      # dft x4 as four component transforms (dftx_kernel)
    (44100, 192000, {"bandwidth": 99.0}),     # 16384-point stage -> polymf_kernel -> dftx_kernel
    (48000, 192000, {"bandwidth": 99.0}),     # 32768-point blocks (four-step transform)
    (44100, 48001, {}),                       # vpoly3 (shared interpolated rows)
    (192000, 44100, {}),                      # half-band -> dft -> vpoly0
])
def test_odd_channel_batch_gives_every_stream_its_own_bits(fi, fo, kw):
    """with an odd channel count the channel pairs of a batch handle used to straddle streams, so a
    stream's bits depended on its neighbours (within 1 ulp).  Pairs now stay inside a stream (pair_channels), and every
    stream of the batch must equal -- bit for bit -- what a one-stream handle gives for the same samples."""
    S, nch, n = 3, 3, 30000
    xs = np.stack([lcg_noise(n, nch, 4321 + s) for s in range(S)])
    got = F.Resampler(fi, fo, nch=nch, nstreams=S, **kw).process(xs, chunk=7000)
    for s in range(S):
        ref = F.Resampler(fi, fo, nch=nch, **kw).process(xs[s], chunk=7000)
        assert got[s].shape == ref.shape, (s, got[s].shape, ref.shape)
        assert np.array_equal(got[s].view(np.uint32), ref.view(np.uint32)), (fi, fo, kw, s)
    ora = Oracle(fi, fo, nch, **kw).process(xs[1], chunk=7000)
    assert got[1].shape == ora.shape
    assert_parity(got[1], ora)


@pytest.mark.parametrize("fi,fo,nch,kw", [
    (48000, 192000, 2, {}),                       # one x4 stage, float frames on both sides
    (44100, 176400, 8, {}),                       # the same on 8-channel frames (four pair-workgroups share every frame)
    (22050, 88200, 1, {}),                        # a single channel: the pair's second half is empty
    (44100, 192000, 8, {"bandwidth": 99.0}),      # last stage of the 3-stage chain: fp64 ring in, float frames out
])
def test_x4_stage_as_component_transforms(fi, fo, nch, kw):
    """dftx_kernel (csrc/dftx.hip): x4 upsampling on 8192-point blocks computed as the four 2048-point transforms of the
    filter's polyphase components instead of the reference's one 8192-point inverse (dft_filter.h:86-104,118-156).  Same
    parity bar against the oracle; bit-identical whatever the push sizes are (the lean instance takes the blocks with
    contiguous spans, the generic one a block at a ring wrap or across two pushes -- same arithmetic)."""
    n = 50000
    x = lcg_noise(n, nch, 777)
    ref = Oracle(fi, fo, nch, **kw).process(x)
    one = F.Resampler(fi, fo, nch=nch, **kw).process(x)
    assert one.shape == ref.shape
    assert_parity(one, ref)
    for chunk in (1777, 9000, 30011):
        got = F.Resampler(fi, fo, nch=nch, **kw).process(x, chunk=chunk)
        assert got.shape == one.shape and np.array_equal(got.view(np.uint32), one.view(np.uint32)), (fi, fo, nch, chunk)
    torch = pytest.importorskip("torch")
    r = F.Resampler(fi, fo, nch=nch, **kw)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    xd = torch.rand((1, 40000, nch), device="cuda") - 0.5
    yd = torch.empty((1, int(40000 * fo / fi) + 4096, nch), device="cuda")
    r.profile(True)
    r.flow_device(xd, 40000, yd, yd.shape[1])
    names = {k["kernel"] for k in r.profile_report()}
    r.profile(False)
    assert "rsmp::dftx_kernel<4>" in names, names
