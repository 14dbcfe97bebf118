#!/usr/bin/env python3
"""PCIe-inclusive rate of the plain ratelib.h calls (RR_push / RR_pull with host buffers)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import foo_dsp_resampler_amd as F
from oracle_binding import lcg_noise

for nch, S in ((2, 1), (2, 64)):
    r = F.Resampler(44100, 96000, nch=nch, nstreams=S)
    P = r.isamp_max
    x = np.stack([lcg_noise(P, nch, 1 + s) for s in range(S)]) if S > 1 else lcg_noise(P, nch, 1)
    r.push(x); r.pull_all()          # warm up (allocations)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        r.push(x)
        n = r.available
        r.pull(n)
    dt = time.perf_counter() - t0
    print("host API: %d stream(s) x %d ch, %d frames/push: %.2f Gsamples/s in (PCIe + pageable host memory included)"
          % (S, nch, P, S * P * nch * reps / dt / 1e9))
