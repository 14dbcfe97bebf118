#!/usr/bin/env python3
"""Seeded random parity sweep: HIP engine vs the CPU oracle over rates, channel counts, qualities,
bandwidths, aliasing, phases, chunk patterns (not part of the pytest suite; run on a GPU box).
`fuzz_parity.py N SEED [PHASE_SHARE]`: PHASE_SHARE (default 0.15) of the cases use phase != 50 -- same bar."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import foo_dsp_resampler_amd as F  # noqa: E402
from oracle_binding import Oracle, lcg_noise  # noqa: E402
from parity import compare_f32  # noqa: E402

RATES = [8000, 11025, 16000, 22050, 24000, 32000, 44100, 48000, 64000, 88200, 96000, 176400, 192000]


def main(n_cases, seed, phase_share=0.15):
    rng = np.random.RandomState(seed)
    worst = {"max_ulp": 0.0, "rel_rms": 0.0}
    bad = 0
    for k in range(n_cases):
        fi, fo = rng.choice(RATES, 2, replace=False)
        if rng.rand() < 0.1:
            fo = int(fo) + int(rng.randint(1, 7))  # irrational-ish ratio -> interpolated polyphase
        nch = int(rng.choice([1, 2, 2, 2, 3, 6]))
        kw = {}
        if rng.rand() < 0.3:
            kw["quality"] = 1
        if rng.rand() < 0.4:
            kw["bandwidth"] = float(rng.choice([90.0, 97.0, 99.0, 99.5]))
        if rng.rand() < 0.2:
            kw["allow_aliasing"] = 1
        if rng.rand() < phase_share:
            kw["phase"] = float(rng.choice([0.0, 10.0, 25.0, 40.0, 60.0, 75.0, 100.0]))
        frames = int(rng.randint(3000, 60000))
        x = lcg_noise(frames, nch, int(rng.randint(1, 1 << 30)))
        # random push pattern, identical for both sides
        cuts = sorted(set(rng.randint(1, frames, size=int(rng.randint(0, 6))).tolist()))
        chunks = np.split(x, cuts)
        try:
            r = F.Resampler(int(fi), int(fo), nch=nch, **kw)
            o = Oracle(int(fi), int(fo), nch, **kw)
        except Exception as e:  # both must refuse the same configurations
            print("case", k, fi, fo, nch, kw, "open failed:", e)
            continue
        got, ref = [], []
        for c in chunks:
            if len(c) == 0:
                continue
            r.push(c); o.push(c)
            a, b = r.pull_all(), o.pull_all()
            assert a.shape == b.shape, (k, fi, fo, nch, kw, a.shape, b.shape)  # same availability after every push
            got.append(a); ref.append(b)
        r.drain(); o.drain()
        got.append(r.pull_all()); ref.append(o.pull_all())
        g, f = np.concatenate(got), np.concatenate(ref)
        assert g.shape == f.shape, (k, fi, fo, g.shape, f.shape)
        rep = compare_f32(g, f)
        worst["max_ulp"] = max(worst["max_ulp"], rep["max_ulp"])
        worst["rel_rms"] = max(worst["rel_rms"], rep["rel_rms"])
        ok = rep["max_ulp"] <= 1.0 and rep["rel_rms"] <= 1e-7  # one bar for every chain and every phase (tests/parity.py)
        if not ok:
            bad += 1
            print("MISMATCH case", k, fi, fo, nch, kw, frames, cuts, rep)
    print("cases", n_cases, "mismatches", bad, "worst", worst)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 2026,
                       float(sys.argv[3]) if len(sys.argv) > 3 else 0.15) else 0)
