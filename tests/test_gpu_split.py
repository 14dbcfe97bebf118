"""GPU tests of the sub-blocked fused path (fused_split_kernel<KS, OMODE>): x2 stages with 8192- or 16384-point blocks
-> vpoly0 (-> a further stage), each block of the reference computed as several 4096-point component transforms
(DESIGN.md 4 "Sub-blocked fused launch").  Everything against the CPU oracle through the C ABI, at the one parity bar."""
import os
import subprocess
import sys

import numpy as np
import pytest

import foo_dsp_resampler_amd as F
from oracle_binding import Oracle, lcg_noise
from parity import assert_parity, compare_f32

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPLIT_KERNELS = tuple("rsmp::fused_split%s_kernel<%d, %d>" % (two, ks, om) for two in ("", "2") for ks in (7, 8, 9) for om in (0, 1, 2))


def _plan_is_split_shaped(fi, fo, **kw):
    st = F.describe_plan(fi, fo, **kw)["stages"]
    return (len(st) >= 2 and st[0]["kind"] == "dft" and st[0]["L"] == 2 and st[0]["dft_length"] in (8192, 16384, 32768)
            and st[1]["kind"] == "poly" and st[1]["interp_order"] == 0)


def _kernels_of(r, x, chunk):
    r.profile(True)
    y = r.process(x, chunk=chunk)
    names = {rec["kernel"] for rec in r.profile_report()}
    return y, names


@pytest.mark.parametrize("fi,fo,nch,kw", [
    (44100, 192000, 2, {"bandwidth": 99.0}),       # BASELINE configs[2]'s chain: 16384-point blocks, three sub-blocks
    (44100, 192000, 8, {"bandwidth": 99.0}),       # ... with its 8-channel frames (four pairs per frame, item_map)
    (44100, 192000, 6, {"bandwidth": 99.0}),
    (44100, 192000, 2, {"bandwidth": 97.0}),       # 8192-point blocks: one workgroup per block, polyphase stage in two rounds
    (44100, 192000, 4, {"bandwidth": 98.0}),       # ... with a longer filter (1425 taps)
    (22050, 96000, 2, {"bandwidth": 99.0}),
    # the polyphase stage LAST (float frames out): what a steep passband makes of the common conversions
    (44100, 48000, 2, {"bandwidth": 99.0}),
    (44100, 96000, 2, {"bandwidth": 98.0}),
    (44100, 96000, 6, {"bandwidth": 99.0}),
    (48000, 44100, 2, {"bandwidth": 99.0}),
    (48000, 44100, 4, {"bandwidth": 97.0}),
    (22050, 64000, 2, {"bandwidth": 99.0}),        # 640 phases (more residue pairs than the workgroup has threads: crashed RR_open once)
    (8000, 44100, 2, {"bandwidth": 99.0}),         # 441 phases: the last 16-residue group is partial
    (44100, 96000, 2, {"bandwidth": 99.5}),        # 32768-point blocks (5493 taps): eleven sub-blocks of 2480 samples
    (44100, 192000, 4, {"bandwidth": 99.5}),
])
def test_sub_blocked_chain_matches_the_oracle(fi, fo, nch, kw):
    if not _plan_is_split_shaped(fi, fo, **kw):
        pytest.skip("this configuration does not plan to x2 dft (8192/16384) -> vpoly0 -> stage: %r" % (F.describe_plan(fi, fo, **kw)["stages"],))
    x = lcg_noise(150000, nch, 77)
    r = F.Resampler(fi, fo, nch=nch, **kw)
    y, names = _kernels_of(r, x, 61000)
    assert names & set(SPLIT_KERNELS), names  # the path under test is the one that ran
    if F.describe_plan(fi, fo, **kw)["stages"][0]["dft_length"] == 8192:  # ... in its whole-block, two-round form
        assert any("fused_split2_kernel" in n for n in names), names
    assert "rsmp::seam_kernel" in names, names
    assert_parity(y, Oracle(fi, fo, nch, **kw).process(x, chunk=61000))


def test_sub_blocked_chain_call_by_call_with_awkward_chunks():
    """Pushes much shorter than a block (the first sub-block's window then starts in fifo 0's ring, several pushes back), one
    frame at a time, a push that completes many blocks, pulls in between: the same availability and samples as the oracle
    after every call."""
    fi, fo, nch, kw = 44100, 192000, 2, {"bandwidth": 99.0}
    x = lcg_noise(90000, nch, 3).reshape(-1, nch)
    r, o = F.Resampler(fi, fo, nch=nch, **kw), Oracle(fi, fo, nch, **kw)
    pos = 0
    for n in [1000, 1, 3000, 2769, 1, 1, 6770, 6771, 40000, 5, 13540, 9000, 2000]:
        seg = x[pos:pos + n]
        pos += n
        r.push(seg); o.push(seg)
        a, b = r.pull_all(), o.pull_all()
        assert a.shape == b.shape, (n, a.shape, b.shape)
        if b.size:
            assert_parity(a, b)
    r.drain(); o.drain()
    a, b = r.pull_all(), o.pull_all()
    assert a.shape == b.shape
    assert_parity(a, b)


def test_sub_blocked_batch_of_streams_and_unaligned_input():
    """A batch handle (every stream its own data) fed from a device buffer that is only 4-byte aligned: the sub-blocked form
    has no generic kernel to fall back to and reads the channels one float at a time."""
    torch = pytest.importorskip("torch")
    fi, fo, nch, S, kw = 44100, 192000, 2, 5, {"bandwidth": 99.0}
    P = 70000
    xs = np.stack([lcg_noise(P, nch, 200 + k).reshape(P, nch) for k in range(S)])
    r = F.Resampler(fi, fo, nch=nch, nstreams=S, **kw)
    raw = torch.zeros(S * P * nch + 1, dtype=torch.float32, device="cuda")
    xin = raw[1:].view(S, P, nch)  # base address = allocation + 4 bytes
    assert xin.data_ptr() % 8 == 4
    xin.copy_(torch.from_numpy(xs))
    torch.cuda.synchronize()  # torch's stream filled the input; the handle works on its own stream
    cap = int(P * fo / fi) + 65536
    y = torch.empty((S, cap, nch), dtype=torch.float32, device="cuda")
    iu, og = r.flow_device(xin, P, y, cap)
    assert iu == P
    torch.cuda.synchronize()
    got = y[:, :og].cpu().numpy()
    for k in (0, 2, 4):
        o = Oracle(fi, fo, nch, **kw)
        o.push(xs[k])
        ref = o.pull_all()
        assert ref.shape[0] == og, (ref.shape, og)
        assert_parity(got[k], ref)


def test_sub_blocked_and_unfused_paths_agree():
    """RSMP_NO_SPLIT=1 (own process: the environment is read once) runs the same chain as dft_kernel<14,..> + polymf_kernel:
    both within the parity bar of the oracle, and of each other."""
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np, foo_dsp_resampler_amd as F\n"
        "from oracle_binding import lcg_noise\n"
        "x = lcg_noise(120000, 2, 12)\n"
        "r = F.Resampler(44100, 192000, nch=2, bandwidth=99.0)\n"
        "r.profile(True)\n"
        "y = r.process(x, chunk=50000)\n"
        "names = sorted({k['kernel'] for k in r.profile_report()})\n"
        "assert not any('fused_split' in n for n in names), names\n"
        "np.save(sys.argv[1], y)\n" % (ROOT, os.path.join(ROOT, "tests")))
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "split_ref_%d.npy" % os.getpid())
    env = dict(os.environ, RSMP_NO_SPLIT="1")
    p = subprocess.run([sys.executable, "-c", code, out], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    y_old = np.load(out)
    os.remove(out)
    x = lcg_noise(120000, 2, 12)
    y_new = F.Resampler(44100, 192000, nch=2, bandwidth=99.0).process(x, chunk=50000)
    ref = Oracle(44100, 192000, 2, bandwidth=99.0).process(x, chunk=50000)
    assert_parity(y_new, ref)
    assert_parity(y_old, ref)
    rep = compare_f32(y_new, y_old)
    assert rep["max_ulp"] <= 1.0 and rep["rel_rms"] <= 1e-7, rep


@pytest.mark.parametrize("fi,fo,nch,kw", [(44100, 48000, 2, {"bandwidth": 99.0}), (44100, 96000, 4, {"bandwidth": 98.0})])
def test_sub_blocked_float_output_in_every_place_the_fifo_can_have_it(fi, fo, nch, kw):
    """The polyphase stage is the last one: outputs go straight into the caller's buffer (flow with room for everything:
    OMODE 0), into the output fifo's ring (push without a destination: OMODE 2), or partly into each (a flow whose buffer is
    smaller than what the push produces; a buffer that is only 4-byte aligned)."""
    torch = pytest.importorskip("torch")
    P, S = 100000, 3
    xs = np.stack([lcg_noise(P, nch, 900 + k).reshape(P, nch) for k in range(S)])
    refs = []
    for k in range(S):
        o = Oracle(fi, fo, nch, **kw)
        o.push(xs[k])
        refs.append(o.pull_all())
    total = refs[0].shape[0]
    xin = torch.from_numpy(xs).cuda()
    torch.cuda.synchronize()

    def names_of(r):
        return {rec["kernel"] for rec in r.profile_report()}

    # (a) room for everything
    r = F.Resampler(fi, fo, nch=nch, nstreams=S, **kw)
    r.profile(True)
    cap = total + 4096
    y = torch.empty((S, cap, nch), dtype=torch.float32, device="cuda")
    iu, og = r.flow_device(xin, P, y, cap)
    torch.cuda.synchronize()
    assert (iu, og) == (P, total)
    assert any(n.endswith(", 0>") and "fused_split" in n for n in names_of(r)), names_of(r)
    for k in range(S):
        assert_parity(y[k, :og].cpu().numpy(), refs[k])
    # (b) push without a destination, pull afterwards
    r = F.Resampler(fi, fo, nch=nch, nstreams=S, **kw)
    r.profile(True)
    r.push_device(xin, P)
    y2 = torch.empty((S, cap, nch), dtype=torch.float32, device="cuda")
    n = r.pull_device(y2, cap)
    torch.cuda.synchronize()
    assert n == total
    assert any(n_.endswith(", 2>") and "fused_split" in n_ for n_ in names_of(r)), names_of(r)
    for k in range(S):
        assert_parity(y2[k, :n].cpu().numpy(), refs[k])
    # (c) a destination that takes a third of it, only 4-byte aligned; the rest is pulled later
    r = F.Resampler(fi, fo, nch=nch, nstreams=S, **kw)
    small = total // 3
    raw = torch.zeros(S * small * nch + 1, dtype=torch.float32, device="cuda")
    y3 = raw[1:].view(S, small, nch)
    assert y3.data_ptr() % 8 == 4
    torch.cuda.synchronize()  # (the zero fill of `raw` runs on torch's stream)
    iu, og = r.flow_device(xin, P, y3, small)
    assert (iu, og) == (P, small)
    rest = torch.empty((S, cap, nch), dtype=torch.float32, device="cuda")
    n = r.pull_device(rest, cap)
    torch.cuda.synchronize()
    assert small + n == total
    for k in range(S):
        assert_parity(np.concatenate([y3[k].cpu().numpy(), rest[k, :n].cpu().numpy()]), refs[k])
