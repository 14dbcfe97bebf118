#!/usr/bin/env python3
"""Bit-invariance to push size through the host API for one configuration (diagnostic)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import foo_dsp_resampler_amd as F
from oracle_binding import lcg_noise
fi, fo, nch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = lcg_noise(35000, nch, 99)
ref = F.Resampler(fi, fo, nch=nch).process(x)
for chunk in (33004, 20829, 10138, 4548, 1000, 333):
    got = F.Resampler(fi, fo, nch=nch).process(x, chunk=chunk)
    same = got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    if same:
        print(fi, fo, nch, "chunk", chunk, "identical")
    else:
        d = np.flatnonzero((got.view(np.uint32) != ref.view(np.uint32)).any(axis=1))
        print(fi, fo, nch, "chunk", chunk, "DIFFERENT frames", len(d), d[:6], "max abs", np.abs(got.astype(np.float64) - ref).max())
