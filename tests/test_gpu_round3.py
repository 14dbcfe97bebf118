"""Round-3 GPU tests: device-aware handles, the self-launching multi-rank bench, the bench shapes of BASELINE configs[2] and
configs[3] against the oracle, RR_flow against the oracle's orc_flow, allocation failures inside RR_open.
Everything goes through the C ABI of libratelib_amd.so (ctypes mirror in foo_dsp_resampler_amd/ratelib.py)."""
import ctypes as C
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import foo_dsp_resampler_amd as F
from foo_dsp_resampler_amd import ratelib as R
from oracle_binding import Oracle, lcg_noise
from devbuf import dev_zeros
from parity import assert_parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------ devices (VERDICT r2 #1b)
def test_handle_remembers_its_device_and_works_from_another_thread():
    """A handle opened on device 0 by one thread is driven by another thread (the plugin's converter threads, chain.h:36):
    every ABI entry selects the handle's device itself.  Output equals the oracle's."""
    x = lcg_noise(30000, 2, 5)
    r = F.Resampler(44100, 96000, nch=2, device=0)
    assert r.device == 0
    out = {}

    def work():
        try:
            out["y"] = r.process(x, chunk=7000)
        except Exception as e:  # surfaced below
            out["err"] = e

    t = threading.Thread(target=work)
    t.start()
    t.join()
    assert "err" not in out, out.get("err")
    assert_parity(out["y"], Oracle(44100, 96000, 2).process(x, chunk=7000))
    # the default placement is the calling thread's current device
    assert F.Resampler(44100, 48000, nch=1).device == 0


def test_open_on_a_device_the_process_does_not_have_fails_cleanly():
    torch = pytest.importorskip("torch")
    L = F.lib()
    R._ensure_init()
    n = torch.cuda.device_count()
    cfg = F.RRConfig(44100, 96000, 50.0, 95.0, 0, 0)
    for dev in (n, n + 5, -1, -7):
        h = C.c_void_p(0x1234)
        rc = L.RRX_open_batch_on(C.byref(cfg), 2, 4, dev, C.byref(h))
        assert rc == 6 and not h.value, (dev, rc, h.value)  # RR_INVPARAM, *handle NULL
    assert L.RRX_device(None) == -1
    # ... and the library still works afterwards
    x = lcg_noise(5000, 2, 1)
    assert_parity(F.Resampler(44100, 48000, 2).process(x), Oracle(44100, 48000, 2).process(x))
    if n > 1:  # a second GPU, when the box has one: the same handle API on device 1
        r = F.Resampler(44100, 96000, nch=2, nstreams=3, device=1)
        assert r.device == 1
        xs = np.stack([lcg_noise(20000, 2, 40 + k) for k in range(3)])
        ys = r.process(xs)
        for k in range(3):
            assert_parity(ys[k], Oracle(44100, 96000, 2).process(xs[k]))


def test_device_round_robin_from_the_environment():
    """RATELIB_AMD_DEVICES=all (read at init_ratelib): RR_open deals handles over the devices.  Own process, because the
    variable is read once."""
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import foo_dsp_resampler_amd as F, torch\n"
        "from oracle_binding import Oracle, lcg_noise\n"
        "from parity import assert_parity\n"
        "n = torch.cuda.device_count()\n"
        "hs = [F.Resampler(44100, 48000, nch=2) for _ in range(2 * n + 1)]\n"
        "assert [h.device for h in hs] == [k %% n for k in range(2 * n + 1)], [h.device for h in hs]\n"
        "x = lcg_noise(9000, 2, 3)\n"
        "ref = Oracle(44100, 48000, 2).process(x)\n"
        "for h in hs: assert_parity(h.process(x), ref)\n"
        "print('ok', n)\n" % (ROOT, os.path.join(ROOT, "tests")))
    env = dict(os.environ, RATELIB_AMD_DEVICES="all")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.startswith("ok"), p.stderr[-2000:]
    bad = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\nimport foo_dsp_resampler_amd as F\nF.Resampler(44100, 48000)" % ROOT],
                         env=dict(os.environ, RATELIB_AMD_DEVICES="0,99"), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "init_ratelib failed" in bad.stderr  # names a device the process does not have


# ------------------------------------------------------------------------------------------------ multi-rank launch (VERDICT r2 #1a)
def test_bench_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher: two fresh child processes (one rank each, both on this box's one GPU,
    gloo for the barrier / MAX), one JSON line from rank 0 with n_gpus 2, whole-job units = both ranks', stream 0 of the
    last timed step checked against the oracle."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(BENCH_SHARE_GPU="1", BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--streams", "16",
                        "--frames", "100000", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["checked"] is True, out
    per_rank_units = 16 * 100000 * 2
    assert abs(out["value"] * 1e6 * out["ms_per_step"] / 1e3 - 2 * per_rank_units) < 1e-3 * per_rank_units  # whole-job aggregate


# ------------------------------------------------------------------------------------------------ bench shapes
def _bench_shape_case(fi, fo, nch, S, kw, check_streams, expect_kernel=None):
    """bench.py's workload for one BASELINE config: S streams x isamp_max frames per push, device-resident in and out
    through RRX_flow_device, two steps (the second one starts from the first one's fifo state, slab cuts included), then a
    drain.  `check_streams` sample by sample against the oracle, every stream by frame count and a distinct checksum."""
    torch = pytest.importorskip("torch")
    import bench
    r = F.Resampler(fi, fo, nch=nch, nstreams=S, **kw)
    n = r.isamp_max
    x = bench.lcg_noise_device(torch, S, n, nch, 12345, "cuda")
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    cap = int(n * fo / fi) + 65536  # a push's output varies by a block or two of the last stage around the mean
    ys, ogs = [], []
    r.profile(True)
    for _ in range(2):
        y = dev_zeros((S, cap, nch))
        iu, og = r.flow_device(x, n, y, cap)
        assert iu == n
        ys.append(y)
        ogs.append(og)
    names = {rec["kernel"] for rec in r.profile_report()}
    r.profile(False)
    if expect_kernel and not os.environ.get("RSMP_NO_SPLIT") and not os.environ.get("RSMP_NO_FAST"):
        assert expect_kernel in names, names  # the kernel the bench line of this config is filed under is the one that ran
    r.drain()
    tail = dev_zeros((S, 16384, nch))
    og2 = r.pull_device(tail, 16384)
    r.sync()
    assert sum(ogs) + og2 == int(2 * n * fo / fi + .5)
    for s in check_streams:
        xs = x[s].cpu().numpy()
        assert np.array_equal(xs, lcg_noise(n, nch, 12345 + s).reshape(n, nch))  # the generator bench.py uses IS SURVEY 8(d)'s LCG
        o = Oracle(fi, fo, nch, **kw)
        for k in range(2):
            o.push(xs)
            ref = o.pull_all(1 << 20)
            assert ref.shape[0] == ogs[k], (k, ref.shape, ogs[k])  # same frames pullable after every push
            assert_parity(ys[k][s, :ogs[k]].cpu().numpy(), ref)
        o.drain()
        assert_parity(tail[s, :og2].cpu().numpy(), o.pull_all())
    for k in range(2):
        sums = ys[k][:, :ogs[k]].double().abs().sum(dim=(1, 2)).cpu().numpy()
        assert np.all(np.isfinite(sums)) and len(np.unique(sums)) == S


def test_cfg2_bench_shape_against_oracle():
    """BASELINE configs[2]: 44.1k->192k, 8 ch, passband 99 % (16384-point dft -> vpoly0 -> x4 dft), 32 streams x 240 844 frames."""
    _bench_shape_case(44100, 192000, 8, 32, {"bandwidth": 99.0}, (0, 17, 31), "rsmp::fused_split_kernel<9, 1>")


def test_cfg3_bench_shape_against_oracle():
    """BASELINE configs[3]: 96k->44.1k, 32 ch, aliasing off, linear phase, 16 streams x 1 048 576 frames (item_map with 16
    pairs per frame)."""
    _bench_shape_case(96000, 44100, 32, 16, {"allow_aliasing": 0, "phase": 50.0}, (0, 9, 15), "rsmp::fused_fast_kernel<12, 8, false>")


def test_cfg0_bench_shape_against_oracle():
    """BASELINE configs[0] as bench.py --config 0 runs it: 44.1k->48k stereo, 256 streams x 963 379 frames."""
    _bench_shape_case(44100, 48000, 2, 256, {}, (0, 100, 255), "rsmp::fused_fast_kernel<11, 8, false>")


# ------------------------------------------------------------------------------------------------ RR_flow vs orc_flow
@pytest.mark.parametrize("fi,fo,nch", [(44100, 96000, 2), (96000, 44100, 3), (44100, 48001, 1)])
def test_flow_against_the_oracles_flow(fi, fo, nch):
    """RR_flow (rate_base.h:571-614) call by call against the oracle's restatement of the same function: small osamp, so
    output piles up in the last fifo between calls; input-less calls; osamp 0; identical iused / ogen and samples."""
    x = lcg_noise(60000, nch, 11).reshape(-1, nch)
    r, o = F.Resampler(fi, fo, nch), Oracle(fi, fo, nch)
    pos = 0
    plan = [(5000, 100), (3000, 0), (0, 4000), (12000, 50000), (1, 7), (0, 0), (20000, 900), (0, 100000), (7000, 3), (12999, 100000)]
    for isamp, osamp in plan:
        seg = x[pos:pos + isamp]
        iu_g, y_g = r.flow(seg, osamp)
        iu_o, y_o = o.flow(seg, osamp)
        assert iu_g == iu_o == isamp, (isamp, osamp, iu_g, iu_o)
        assert y_g.shape == y_o.shape, (isamp, osamp, y_g.shape, y_o.shape)
        if y_o.size:
            assert_parity(y_g, y_o)
        pos += isamp
    r.drain(); o.drain()
    a, b = r.pull_all(), o.pull_all()
    assert a.shape == b.shape
    assert_parity(a, b)


# ------------------------------------------------------------------------------------------------ allocation failures inside RR_open
def test_allocation_failure_in_the_middle_of_open():
    """RRX_debug_fail_alloc(k) armed BEFORE RR_open: the k-th device allocation of the handle's construction fails.
    RR_ENOMEM, *handle NULL, the init_ratelib handler runs exactly once (xmalloc.c:38-43), the half-built engine is torn
    down, and the next open / process still matches the oracle."""
    L = F.lib()
    R._ensure_init()
    x = lcg_noise(12000, 2, 9)
    ref = Oracle(44100, 96000, 2).process(x)
    cfg = F.RRConfig(44100, 96000, 50.0, 95.0, 0, 0)
    hit = 0
    for k in (1, 2, 3, 5, 8, 12, 40):
        before = R.alloc_handler_calls
        L.RRX_debug_fail_alloc(k)
        h = C.c_void_p(0x1234)
        rc = L.RR_open(C.byref(cfg), 2, C.byref(h))
        L.RRX_debug_fail_alloc(0)
        if rc == 0:  # the construction makes fewer than k allocations: nothing failed
            L.RR_close(C.byref(h))
            assert R.alloc_handler_calls == before
            continue
        hit += 1
        assert rc == 1 and not h.value, (k, rc, h.value)  # RR_ENOMEM, NULL handle
        assert R.alloc_handler_calls == before + 1, k
        assert_parity(F.Resampler(44100, 96000, 2).process(x), ref)
    assert hit >= 5


# ------------------------------------------------------------------------------------------------ plugin-sized host pushes
@pytest.mark.parametrize("fi,fo,nch,S", [(44100, 96000, 2, 1), (96000, 44100, 3, 1), (44100, 48000, 2, 3), (44100, 192000, 2, 1)])
def test_host_mirror_against_oracle_call_by_call(fi, fo, nch, S):
    """RR_push / RR_pull at plugin chunk sizes run without copy commands: the kernels read the push out of a page-locked slot
    and write what it produces into a page-locked mirror that RR_pull copies from.  Random push sizes and PARTIAL pulls (so that
    frames stay behind in the mirror while the next push arrives, a later pull crosses from the mirror into the device ring,
    a device pull and a drain find frames in the mirror): every call's frame count and samples against the oracle."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(fi + fo + nch + S)
    total = 120000
    xs = [lcg_noise(total, nch, 70 + s).reshape(total, nch) for s in range(S)]
    r = F.Resampler(fi, fo, nch=nch, nstreams=S)
    os_ = [Oracle(fi, fo, nch) for _ in range(S)]
    pos = 0
    step = 0
    while pos < total - 9000:
        n = int(rng.choice([1, 7, 256, 1024, 1024, 2048, 4096, 8192, 3000, 9000]))
        seg = [x[pos:pos + n] for x in xs]
        r.push(seg[0] if S == 1 else np.stack(seg))
        for o, sg in zip(os_, seg):
            o.push(sg)
        pos += n
        step += 1
        mode = step % 5
        want = int(rng.integers(0, int(n * fo / fi) + 300))
        if mode == 0:
            continue  # leave everything where it is: the next push finds frames in the mirror
        if mode == 3 and S == 1:  # device pull of part of what is there
            t = dev_zeros((max(want, 1), nch))
            got_n = r.pull_device(t, want) if want else 0
            r.sync()
            ref = os_[0].pull(want) if want else np.empty((0, nch), np.float32)
            assert got_n == ref.shape[0], (step, got_n, ref.shape)
            if got_n:
                assert_parity(t[:got_n].cpu().numpy(), ref)
            continue
        if mode == 4:  # pull until empty, as the plugin does
            a = r.pull_all(5000)
            for k, o in enumerate(os_):
                b = o.pull_all(5000)
                ak = a if S == 1 else a[k]
                assert ak.shape == b.shape, (step, ak.shape, b.shape)
                if b.size:
                    assert_parity(ak, b)
            continue
        a = r.pull(want) if want else None  # partial pull
        for k, o in enumerate(os_):
            b = o.pull(want) if want else np.empty((0, nch), np.float32)
            ak = np.empty((0, nch), np.float32) if a is None else (a if S == 1 else a[k])
            assert ak.shape == b.shape, (step, ak.shape, b.shape)
            if b.size:
                assert_parity(ak, b)
    r.drain()
    a = r.pull_all()
    for k, o in enumerate(os_):
        o.drain()
        b = o.pull_all()
        ak = a if S == 1 else a[k]
        assert ak.shape == b.shape
        assert_parity(ak, b)


@pytest.mark.parametrize("fo,nch,S,kw", [(48000, 2, 64, "{}"), (48000, 2, 64, "{'bandwidth': 99.0}"), (192000, 4, 24, "{'bandwidth': 99.0}")])
def test_many_launches_per_push(fo, nch, S, kw):
    """[The second and third case: the sub-blocked kernel (16384-point blocks as three sub-blocks each), float frames out and
    fp64 ring out -- its table counts sub-blocks, so the same squeeze cuts a push into as many launches.]
    A seam ring squeezed to 4 MB (RSMP_SEAM_RING_MB, read once per process: own process) cuts one push of 64 stereo
    streams x 300 000 frames into 64-block launches, ~3 per push: seam kernels on the side stream beside the next launch, block
    table halves and seam-ring slots reused within the push -- the situation in which round 2 lost seam outputs.
    Three streams against the oracle over two pushes and a drain."""
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np, torch, foo_dsp_resampler_amd as F, bench\n"
        "from oracle_binding import Oracle\n"
        "from parity import assert_parity\n"
        "S, n, nch, fi, fo, kw = %d, 300000, %d, 44100, %d, %s\n"
        "r = F.Resampler(fi, fo, nch=nch, nstreams=S, **kw)\n"
        "x = bench.lcg_noise_device(torch, S, n, nch, 999, 'cuda')\n"
        "r.set_stream(torch.cuda.current_stream().cuda_stream)\n"
        "r.profile(True)\n"
        "cap = int(n * fo / fi) + 65536\n"
        "ys, ogs = [], []\n"
        "for _ in range(2):\n"
        "    y = torch.zeros((S, cap, nch), device='cuda'); torch.cuda.synchronize(); iu, og = r.flow_device(x, n, y, cap); ys.append(y); ogs.append(og)\n"
        "launches = max([k['launches'] for k in r.profile_report() if 'fused' in k['kernel'] and 'prep' not in k['kernel']] or [0])\n"
        "r.profile(False)\n"
        "for _ in range(2):\n"
        "    y = torch.zeros((S, cap, nch), device='cuda'); torch.cuda.synchronize(); iu, og = r.flow_device(x, n, y, cap); ys.append(y); ogs.append(og)\n"
        "r.drain(); tail = torch.zeros((S, 65536, nch), device='cuda'); torch.cuda.synchronize(); og2 = r.pull_device(tail, 65536); r.sync()\n"
        "for s in (0, S // 2, S - 1):\n"
        "    o = Oracle(fi, fo, nch, **kw); xs = x[s].cpu().numpy()\n"
        "    for k in range(4):\n"
        "        o.push(xs); ref = o.pull_all(1 << 22)\n"
        "        assert ref.shape[0] == ogs[k]; assert_parity(ys[k][s, :ogs[k]].cpu().numpy(), ref)\n"
        "    o.drain(); assert_parity(tail[s, :og2].cpu().numpy(), o.pull_all())\n"
        "print('ok', launches)\n" % (ROOT, os.path.join(ROOT, "tests"), S, nch, fo, kw))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RSMP_SEAM_RING_MB="4"), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and p.stdout.startswith("ok"), (p.stdout[-500:], p.stderr[-2000:])
    # at least three launches per push in the profiled pushes (main-stream seams) ... (two with the x4 stage's rings in the slab budget)
    if not (kw != "{}" and os.environ.get("RSMP_NO_SPLIT")):  # (with the sub-blocked kernel switched off these chains have no fused launch)
        assert int(p.stdout.split()[1]) >= (6 if fo == 48000 else 4), p.stdout
