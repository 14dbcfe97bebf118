#!/usr/bin/env python3
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import foo_dsp_resampler_amd as F
from oracle_binding import Oracle, lcg_noise
from parity import compare_f32
fi, fo = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 1800
x = lcg_noise(n, 2, 21)
got = F.Resampler(fi, fo, 2).process(x, chunk=chunk)
ref = Oracle(fi, fo, 2).process(x, chunk=chunk)
print(fi, fo, "fuse" if not os.environ.get("RSMP_NO_FUSE") else "nofuse", got.shape, ref.shape, compare_f32(got, ref))
if got.shape == ref.shape:
    bad = np.nonzero(np.abs(got.astype(np.float64) - ref) > 1e-6)[0]
    if bad.size:
        # contiguous runs
        runs = np.split(bad, np.nonzero(np.diff(bad) > 1)[0] + 1)
        print("bad frames:", bad.size, "runs:", [(int(r[0]), int(r[-1])) for r in runs[:12]])
