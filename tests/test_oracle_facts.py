"""Pins the CPU oracle (oracle/rate_oracle.c) on every reference-run fact recorded in SURVEY.md
([probe] items), held in tests/golden/survey_probe_facts.json.  CPU only."""
import numpy as np
import pytest

from oracle_binding import Oracle, lcg_noise, lib

CFG_KEYS = ["cfg1_44k1_48k", "cfg2_44k1_96k", "cfg3_44k1_192k", "cfg4_96k_44k1"]


def open_cfg(facts, key, nch=2):
    c = facts["configs"][key]
    return Oracle(c["in_rate"], c["out_rate"], nch, phase=c["phase"], bandwidth=c["bandwidth"],
                  allow_aliasing=c["allow_aliasing"], quality=c["quality"])


def check_plan(plan, expect):
    assert len(plan) == len(expect), (plan, expect)
    for got, want in zip(plan, expect):
        for k, v in want.items():
            if k.startswith("_"):
                continue
            if k == "step_fraction_u32_as_printed":
                assert "%.8f" % ((got["step"] & 0xFFFFFFFF) / 1e10) == "0." + v
            else:
                assert got[k] == v, (k, got, want)


@pytest.mark.parametrize("key", ["cfg1_44k1_48k", "cfg2_44k1_96k", "cfg3_44k1_192k", "cfg4_96k_44k1",
                                 "phase0_44k1_48k", "down2_88k2_44k1", "down_192k_44k1",
                                 "irr_44100_48001", "norm_44k1_48k"])
def test_stage_plans(facts, key):
    check_plan(open_cfg(facts, key).plan(), facts["plans"][key])


@pytest.mark.parametrize("key", CFG_KEYS)
def test_design_call_arguments(facts, key):
    trace = [t for t in open_cfg(facts, key, nch=1).design_trace() if t["Fn"] > 0]
    want = facts["design_calls"][key]
    assert len(trace) == len(want)
    # the oracle designs pre, arb, post in that order, as the table lists them
    for got, w in zip(trace, want):
        assert got["k"] == w["k"] and got["num_taps"] == w["num_taps"]
        assert abs(got["Fp"] - w["Fp"]) < 1e-10
        assert abs(got["Fs"] - w["Fs"]) < 1e-10
        assert abs(got["Fn"] - w["Fn"]) < 1e-9
        assert abs(got["att"] - w["att"]) < 1e-5


def test_scalars(facts):
    s = facts["scalars"]
    assert open_cfg(facts, "cfg2_44k1_96k").isamp_max == s["isamp_max_cfg2"]
    # H[0] of the dft filter = DC gain * 2L/N
    assert abs(open_cfg(facts, "cfg1_44k1_48k").dft_spectrum(0)[0] - s["H0_cfg1"]) < 1e-9
    assert abs(open_cfg(facts, "cfg4_96k_44k1").dft_spectrum(0)[0] - s["H0_cfg4"]) < 1e-9


@pytest.mark.parametrize("key", CFG_KEYS)
def test_frame_counts_and_rechunk_invariance(facts, key):
    fc = facts["frame_counts"]
    x = lcg_noise(fc["input_frames"], fc["nch"], facts["lcg"]["seed"])
    ref = open_cfg(facts, key).process(x)
    assert ref.shape[0] == fc[key]
    for chunk in fc["rechunk_sizes"]:
        y = open_cfg(facts, key).process(x, chunk=chunk)
        assert y.shape == ref.shape and np.array_equal(y, ref)


def test_cfg2_block_accounting(facts):
    a = facts["cfg2_accounting"]
    x = lcg_noise(a["push_frames"], 2, facts["lcg"]["seed"])
    o = open_cfg(facts, "cfg2_44k1_96k")
    o.push(x)
    assert len(o.stage_fifo(0, 0)) == a["stage0_left_buffered"]
    assert len(o.stage_fifo(0, 2)) == a["stage1_outputs"]
    produced = a["stage0_blocks"] * a["stage0_produce_per_block"]
    # stage-1 fifo keeps its 11 preload zeros + everything stage 0 produced minus what vpoly0 consumed
    consumed = (a["stage1_outputs"] * 147) // 160
    assert (a["stage1_outputs"] * 147) % 160 == a["stage1_at_carry"]
    assert len(o.stage_fifo(0, 1)) == 11 + produced - consumed

    o = open_cfg(facts, "cfg2_44k1_96k")
    o.push(x[: a["single_push_frames"]])
    assert o.pull_all().shape[0] == a["pullable_after_single_push"]


def test_cfg2_impulse_position(facts):
    a = facts["cfg2_accounting"]
    imp = np.zeros((8000, 2), np.float32)
    imp[a["impulse_in_index"]] = 1
    y = open_cfg(facts, "cfg2_44k1_96k").process(imp)
    assert int(np.argmax(np.abs(y[:, 0]))) == a["impulse_out_peak_index"]


def test_cfg2_poly_prototype(facts):
    p = facts["cfg2_poly_prototype"]
    o = open_cfg(facts, "cfg2_44k1_96k")
    n, L = 24, p["L"]
    tab = o.poly_table().reshape(L, n)
    h = np.zeros(n * L)
    for i in range(n):
        for ph in range(L):
            pos = i * L + ph - 1
            if pos >= 0:
                h[pos] = tab[ph, n - 1 - i]
    h = h[: p["num_taps"]]
    assert np.array_equal(h, h[::-1])
    assert abs((h.sum() - L) - p["sum_minus_L"]) < p["sum_tolerance"]
    assert abs(h.max() - p["peak"]) < p["peak_tolerance"] and int(np.argmax(h)) == p["num_taps"] // 2


# ---- checks that do not depend on the reference at all ----
def test_rdft_conventions():
    rng = np.random.default_rng(1)
    for n in (8, 64, 2048):
        x = rng.standard_normal(n)
        a = x.copy()
        lib().orc_rdft(n, 1, a.ctypes.data)
        X = np.fft.fft(x).conj()  # e^{+i...} convention
        assert np.allclose(a[0], X[0].real) and np.allclose(a[1], X[n // 2].real)
        assert np.allclose(a[2::2], X[1:n // 2].real) and np.allclose(a[3::2], X[1:n // 2].imag)
        lib().orc_rdft(n, -1, a.ctypes.data)
        assert np.allclose(a, x * n / 2)


@pytest.mark.parametrize("rates", [(44100, 96000), (44100, 48000), (96000, 44100), (192000, 44100), (44100, 48001),
                                   (88200, 44100), (32000, 96000), (48000, 32000)])
def test_sine_passes_unchanged(rates):
    """A passband sine must come out as the same sine at the new rate (zero net delay at phase 50),
    to about the 'Best' 28-bit accuracy."""
    fi, fo = rates
    n = 30000
    f0 = 997.0
    t = np.arange(n) / fi
    x = np.stack([np.sin(2 * np.pi * f0 * t), np.cos(2 * np.pi * f0 * t)], 1).astype(np.float32)
    y = Oracle(fi, fo, 2).process(x)
    m = y.shape[0]
    assert m == int(n * fo / fi + .5)
    to = np.arange(m) / fo
    lo, hi = int(0.2 * m), int(0.8 * m)  # away from the start/stop transients
    assert np.max(np.abs(y[lo:hi, 0] - np.sin(2 * np.pi * f0 * to[lo:hi]))) < 2e-7
    assert np.max(np.abs(y[lo:hi, 1] - np.cos(2 * np.pi * f0 * to[lo:hi]))) < 2e-7


def test_stopband_rejected():
    """A tone between the new Nyquist and the old one must vanish when downsampling."""
    fi, fo = 96000, 44100
    n = 40000
    t = np.arange(n) / fi
    x = np.sin(2 * np.pi * 30000.0 * t).astype(np.float32).reshape(-1, 1)
    y = Oracle(fi, fo, 1).process(x)
    m = y.shape[0]
    assert np.max(np.abs(y[int(0.2 * m):int(0.8 * m)])) < 1e-7


def test_flow_equals_push_pull():
    x = lcg_noise(30000, 2, 7)
    a = Oracle(44100, 96000, 2)
    b = Oracle(44100, 96000, 2)
    outs_a, outs_b = [], []
    for s in range(0, 30000, 5000):
        a.push(x[s:s + 5000])
        outs_a.append(a.pull_all())
        iu, o = b.flow(x[s:s + 5000], 1 << 16)
        assert iu == 5000
        outs_b.append(o.copy())
    assert np.array_equal(np.concatenate(outs_a), np.concatenate(outs_b))
