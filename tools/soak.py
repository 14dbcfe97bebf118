#!/usr/bin/env python3
"""Long-stream soak: thousands of plugin-sized pushes through the host API against the oracle (ring wrap,
seam-slot reuse, block-table reuse over many launches)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import foo_dsp_resampler_amd as F
from oracle_binding import Oracle, lcg_noise
from parity import compare_f32

CASES = ((44100, 96000, 2, 4410, 3000, {}), (96000, 44100, 6, 9600, 1000, {}), (44100, 192000, 2, 4410, 800, {}),
         # steep passbands: the sub-blocked kernels behind the host mirror (16384-point blocks; 8192-point blocks, two rounds)
         (44100, 96000, 2, 4410, 1500, {"bandwidth": 99.0}), (44100, 48000, 4, 4410, 1000, {"bandwidth": 98.0}),
         (44100, 192000, 2, 4410, 600, {"bandwidth": 99.0}))
for fi, fo, nch, chunk, n_chunks, kw in CASES:
    r, o = F.Resampler(fi, fo, nch=nch, **kw), Oracle(fi, fo, nch, **kw)
    worst, total, t0 = 0.0, 0, time.time()
    rng = np.random.RandomState(5)
    for k in range(n_chunks):
        n = chunk if k % 7 else int(rng.randint(1, 2 * chunk))
        x = lcg_noise(n, nch, 1000 + k)
        r.push(x); o.push(x)
        a, b = r.pull_all(), o.pull_all()
        assert a.shape == b.shape, (k, a.shape, b.shape)
        if a.size:
            rep = compare_f32(a, b)
            worst = max(worst, rep["max_ulp"])
            assert rep["max_ulp"] <= 1.0 and rep["rel_rms"] <= 1e-7, (k, rep)
        total += a.shape[0]
    r.drain(); o.drain()
    a, b = r.pull_all(), o.pull_all()
    assert a.shape == b.shape
    print("%d->%d %dch %r: %d pushes, %d output frames, worst %.3f ulp, %.1f s" % (fi, fo, nch, kw, n_chunks, total + a.shape[0], worst, time.time() - t0))
print("soak ok")
