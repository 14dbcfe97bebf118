#!/usr/bin/env python3
"""PCIe-inclusive rate of the plain ratelib.h calls (RR_push / RR_pull with host buffers) against the CPU oracle on the
same box: one stereo stream at the plugin's chunk sizes (foo_dsp_rate.cpp:99-103,182-202: 1-8 k frames per on_chunk,
push then pull-until-empty), larger pushes, and batch handles.  Prints one line per case; not a bench.py line."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import foo_dsp_resampler_amd as F  # noqa: E402
from oracle_binding import Oracle, lcg_noise  # noqa: E402


def plugin_loop(r, x, chunk, outbuf, seconds=1.5):
    """push one chunk, pull until nothing is left (what dsp_rate::on_chunk does); returns input frames per second"""
    n, pos, frames = x.shape[-2], 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        c = x[..., pos:pos + chunk, :]
        r.push(c)
        while True:
            y = r.pull(outbuf)
            if y.shape[-2] == 0:
                break
        frames += c.shape[-2]
        pos = (pos + chunk) % (n - chunk)
    return frames / (time.perf_counter() - t0)


def main():
    fi, fo = 44100, 96000
    rows = []
    for nch, S, chunks in ((2, 1, (1024, 4096, 8192, 65536, 481689)), (2, 64, (4096, 65536)), (8, 1, (4096,)), (32, 1, (4096,))):
        for chunk in chunks:
            total = max(2 * chunk, 1 << 17)
            x = np.stack([lcg_noise(total, nch, 1 + s) for s in range(S)]) if S > 1 else lcg_noise(total, nch, 1)
            outbuf = int(chunk * fo / fi) + 4096
            g = F.Resampler(fi, fo, nch=nch, nstreams=S)
            plugin_loop(g, x, chunk, outbuf, 0.3)   # warm up: allocations, ring growth
            gpu = plugin_loop(g, x, chunk, outbuf) * nch * S
            cpu = None
            if S == 1:
                o = Oracle(fi, fo, nch)
                cpu = plugin_loop(o, x, chunk, outbuf) * nch
            rows.append((S, nch, chunk, gpu / 1e6, cpu / 1e6 if cpu else None))
            print("host API %d stream(s) x %d ch, %6d-frame push + pull: GPU %8.2f Msamples/s in%s" %
                  (S, nch, chunk, gpu / 1e6, ",  CPU oracle (1 thread) %7.2f,  GPU/CPU %.2f" % (cpu / 1e6, gpu / cpu) if cpu else ""),
                  flush=True)
    return rows


if __name__ == "__main__":
    main()
