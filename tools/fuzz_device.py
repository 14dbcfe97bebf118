#!/usr/bin/env python3
"""Seeded random sweep of the device-pointer / batch API: S lock-stepped streams pushed and pulled through
RRX_flow_device / RRX_push_device / RRX_pull_device with random chunking and random output capacities (so that
outputs land partly in the caller's buffer, partly in the ring) must give, per stream, exactly the bits the
single-stream host API gives for that stream's samples -- odd channel counts included: channel pairs never straddle
two streams (pair_channels in csrc/fifo_device.hpp)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import foo_dsp_resampler_amd as F  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))

RATES = [8000, 11025, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 176400, 192000]


def main(n_cases, seed):
    rng = np.random.RandomState(seed)
    bad = 0
    for k in range(n_cases):
        fi, fo = [int(v) for v in rng.choice(RATES, 2, replace=False)]
        nch = int(rng.choice([1, 2, 2, 3, 4, 5, 6]))
        S = int(rng.choice([1, 2, 3, 8]))
        kw = {}
        if rng.rand() < 0.3:
            kw["bandwidth"] = 99.0
        if rng.rand() < 0.25:
            kw["quality"] = 1
        frames = int(rng.randint(4000, 50000))
        g = torch.Generator(device="cuda").manual_seed(int(rng.randint(1, 1 << 30)))
        x = torch.rand((S, frames, nch), generator=g, device="cuda") - 0.5
        r = F.Resampler(fi, fo, nch=nch, nstreams=S, **kw)
        r.set_stream(torch.cuda.current_stream().cuda_stream)
        outs = []
        pos = 0
        while pos < frames:
            n = min(frames - pos, int(rng.randint(1, min(frames, r.isamp_max) + 1)))
            cap = int(rng.randint(1, int(n * fo / fi) + 4000))
            y = torch.full((S, cap, nch), float("nan"), device="cuda")
            if rng.rand() < 0.5:
                xin = x[:, pos:pos + n].contiguous()
                iu, og = r.flow_device(xin, n, y, cap)
                assert iu == n, (iu, n)
            else:
                xin = x[:, pos:pos + n].contiguous()
                r.push_device(xin, n)
                og = r.pull_device(y, cap)
            outs.append(y[:, :og].clone())
            pos += n
        r.drain()
        while True:
            y = torch.full((S, 8192, nch), float("nan"), device="cuda")
            og = r.pull_device(y, 8192)
            if og == 0:
                break
            outs.append(y[:, :og].clone())
        r.sync()
        got = torch.cat(outs, dim=1).cpu().numpy()
        for s in range(S):
            ref = F.Resampler(fi, fo, nch=nch, **kw).process(x[s].cpu().numpy())
            same = got[s].shape == ref.shape and np.array_equal(got[s].view(np.uint32), ref.view(np.uint32))
            if not same:
                bad += 1
                print("MISMATCH case", k, fi, fo, nch, S, kw, frames, "stream", s, got[s].shape, ref.shape)
                break
    print("cases", n_cases, "mismatches", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 7) else 0)
