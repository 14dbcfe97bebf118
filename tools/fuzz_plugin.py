#!/usr/bin/env python3
"""Seeded random sweep of the plugin layer: the caller-side harness (oracle/plugin_harness.c) over the product's
RR_* entry points against the same harness over the CPU oracle, on identical chunk sequences -- short and long tracks, 1-6 channels, random chunk sizes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import foo_dsp_resampler_amd as F  # noqa: E402
from oracle_binding import OracleDsp, PluginOnGpu  # noqa: E402
from parity import compare_f32  # noqa: E402
from test_plugin_layer import music_like, run_track  # noqa: E402

RATES = [8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 192000]


def main(n_cases, seed):
    rng = np.random.RandomState(seed)
    bad = 0
    for k in range(n_cases):
        fs, fo = [int(v) for v in rng.choice(RATES, 2, replace=False)]
        nch = int(rng.choice([1, 2, 2, 6]))
        n = int(rng.choice([rng.randint(1, 200), rng.randint(200, 6000), rng.randint(6000, 40000)]))
        x = music_like(n, nch, fs, int(rng.randint(1, 1 << 30)))
        sizes = [int(rng.randint(1, 8193)) for _ in range(int(rng.randint(1, 8)))]
        ref, lat_r = run_track(OracleDsp(fo), x, fs, sizes)
        got, lat_g = run_track(PluginOnGpu(fo), x, fs, sizes)
        ok = [c.shape for c, _ in got] == [c.shape for c, _ in ref] and [r for _, r in got] == [r for _, r in ref] and lat_g == lat_r
        if ok and ref:
            yg, yr = np.concatenate([c for c, _ in got]), np.concatenate([c for c, _ in ref])
            if yr.size:
                rep = compare_f32(yg, yr)
                ok = rep["max_ulp"] <= 1.0 and rep["rel_rms"] <= 1e-7
        if not ok:
            bad += 1
            print("MISMATCH case", k, fs, fo, nch, n, sizes, [c.shape for c, _ in got][:4], [c.shape for c, _ in ref][:4])
    print("cases", n_cases, "mismatches", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 3) else 0)
