#!/usr/bin/env python3
"""Targeted repro for the device API: fixed config, several random chunkings, reports the size of any difference."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import foo_dsp_resampler_amd as F
from parity import compare_f32
fi, fo, nch, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
kw = {"bandwidth": float(sys.argv[5])} if len(sys.argv) > 5 else {}
frames = 35000
for trial in range(12):
    rng = np.random.RandomState(trial)
    g = torch.Generator(device="cuda").manual_seed(trial + 1)
    x = torch.rand((S, frames, nch), generator=g, device="cuda") - 0.5
    r = F.Resampler(fi, fo, nch=nch, nstreams=S, **kw)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    outs, pos, log = [], 0, []
    while pos < frames:
        n = min(frames - pos, int(rng.randint(1, min(frames, r.isamp_max) + 1)))
        cap = int(rng.randint(1, int(n * fo / fi) + 4000))
        y = torch.full((S, cap, nch), float("nan"), device="cuda")
        mode = rng.rand() < 0.5
        xin = x[:, pos:pos + n].contiguous()
        if mode:
            iu, og = r.flow_device(xin, n, y, cap)
        else:
            r.push_device(xin, n); og = r.pull_device(y, cap)
        log.append((n, cap, og, "flow" if mode else "pushpull"))
        outs.append(y[:, :og].clone()); pos += n
    r.drain()
    while True:
        y = torch.full((S, 8192, nch), float("nan"), device="cuda")
        og = r.pull_device(y, 8192)
        if og == 0: break
        outs.append(y[:, :og].clone())
    r.sync()
    got = torch.cat(outs, dim=1).cpu().numpy()
    for s in range(S):
        ref = F.Resampler(fi, fo, nch=nch, **kw).process(x[s].cpu().numpy())
        if not np.array_equal(got[s].view(np.uint32), ref.view(np.uint32)):
            d = np.abs(got[s].astype(np.float64) - ref.astype(np.float64))
            idx = np.flatnonzero(d.max(axis=1) > 0)
            print("trial", trial, "stream", s, "diff frames", len(idx), "first", idx[:5], "last", idx[-5:], "max abs", d.max(), "nan", int(np.isnan(got[s]).sum()), compare_f32(got[s], ref) if not np.isnan(got[s]).any() else "", log)
            break
    else:
        print("trial", trial, "ok")
